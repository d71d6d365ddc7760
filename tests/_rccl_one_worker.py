"""Worker of tests/test_multi_gpu_nccl.py::test_rccl_with_one_rank: ONE rank under torch.distributed with backend "nccl" (= RCCL).
Pushes every collective the multi-GPU forms issue through `geobo_amd.sharding` on device tensors of the shapes an 8-rank 64^3 step
uses, compares with the EmulatedGroup / identity result, then runs a 64 x 48 x 64 inversion in the forced row form with the
collectives forced through RCCL and compares it with the same engine run without a backend call.
    python -m torch.distributed.run --nproc-per-node 1 tests/_rccl_one_worker.py out.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl")
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    from geobo_amd import sharding as S
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *shape: torch.randn(shape, dtype=torch.float64, device="cuda", generator=g)
    # (1) the row form's all-gather: fp64 row blocks of AkA (a rank of 8 at 64^3: 512 rows x 3 Ms_pad), all_gather_into_tensor
    loc = rnd(512, 3 * 4096)
    got = S.gather_rows(loc, 1, None, force=True)
    want = S.gather_rows(loc, 1, S.EmulatedGroup(0, 1))
    res["all_gather_into_tensor_equal"] = bool(torch.equal(got, want)) and tuple(got.shape) == (1, 512, 3 * 4096)
    # (2) its all-reduce: partial sums of squares, P_c N doubles
    v = rnd(2 * 262144)
    w = S.allreduce_sum_(v.clone(), 1, None, force=True)
    res["all_reduce_sum_equal"] = bool(torch.equal(v, w))
    # (3) the agreement of the ranks on the form of a step (one MIN all-reduce of [flag, -flag])
    res["agree"] = [bool(S.agree(f, 1, None, torch.device("cuda"), force=True)) for f in (True, False)]
    # (4) the column form's collectives: all-reduce of a partial AkA block, all-gather of slices, all-to-all (one rank: identity)
    a = rnd(1024, 1024)
    res["all_reduce_matrix_equal"] = bool(torch.equal(S.allreduce_sum_(a.clone(), 1, None, force=True), a))
    send = rnd(1, 4096)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    res["all_to_all_single_equal"] = bool(torch.equal(recv, send))
    ones = torch.ones(1, dtype=torch.float64, device="cuda")
    dist.all_reduce(ones)
    res["ranks_reported_by_backend"] = int(ones.item())
    # (5) a whole step in the row form with its collectives issued through RCCL, against the same step without a backend call
    from conftest import settings_for
    from geobo_amd.inversion import Inversion
    import bench
    os.environ["GEOBO_ROWS"] = "1"
    cubes = {}
    for forced in (False, True):
        inv = Inversion(settings=settings_for(64, 48, 64, kernelfunc="matern32"), props=(0, 1), rank=0, world=1, device="cuda:0")
        inv.engine.force_collectives = forced
        inv.engine.kernel_events = []
        grav, mag, loc_s, drill0 = bench.synthetic_inputs(inv, 20)
        inv.engine.clear_operators()
        inv.gp_length = np.array([200.0, 202.0, 204.0])
        cubes[forced] = np.asarray(inv.cubing(grav, mag, drill0[drill0 != 0], loc_s, drill0))
        assert inv.engine.step_route == "rows"
        res["collectives_timed_%s" % ("forced" if forced else "plain")] = sorted({e[0] for e in inv.engine.kernel_events if e[0].startswith("xgmi")})
    keep = [0, 1, 3, 4]
    res["row_form_step_max_abs_diff"] = float(np.abs(cubes[True][keep] - cubes[False][keep]).max())
    res["row_form_step_bit_identical"] = bool(np.array_equal(cubes[True][keep], cubes[False][keep]))
    torch.cuda.synchronize()
    json.dump(res, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
