"""Worker for the multi-rank tests: one rank of a 32^3 Matern inversion under torch.distributed (backend from argv), rank 0
saves the six cubes.  Started by `python -m torch.distributed.run --nproc-per-node N tests/_rank_worker.py <backend> <out.npz>`."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend, out = sys.argv[1], sys.argv[2]
    dims = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "32").split("x")]
    dims = dims * 3 if len(dims) == 1 else dims
    local = int(os.environ["LOCAL_RANK"]) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    from conftest import settings_for
    from geobo_amd.inversion import Inversion
    import bench
    s = settings_for(*dims, kernelfunc="matern32")
    assembly = sys.argv[4] if len(sys.argv) > 4 else "f64"
    operators = sys.argv[5] if len(sys.argv) > 5 else "resident"
    props = (0, 1) if dims[0] >= 64 and os.environ.get("GEOBO_TEST_PROPS", "") != "3" else (0, 1, 2)
    inv = Inversion(settings=s, props=props, rank=rank, world=world, device="cuda:%d" % local,
                    assembly=assembly, operators=operators)
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv, int(os.environ.get("GEOBO_TEST_DRILL", "20")))
    inv.engine.clear_operators()
    inv.gp_length = np.array([200.0, 202.0, 204.0])
    cubes = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)
    if rank == 0:
        np.savez(out, cubes=np.asarray(cubes), logl=inv.logl, world=int(ones.item()), exchange=bool(inv.engine.exchange),
                 row_gram=bool(inv.engine._row_gram()), rowpath=bool(inv.engine._rowpath))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
