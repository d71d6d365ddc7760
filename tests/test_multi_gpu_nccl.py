"""RCCL path on real hardware: the same 32^3 inversion on 2 (and 4) `nccl` ranks, one process per GPU, against the 1-rank
run.  Skipped on boxes with fewer devices (the driver's single-GPU tier); the gloo tests cover the host logic on CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import normwise

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(n, backend, out, size="32", extra=(), env_extra=None):
    _release_device_memory()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_rank_worker.py"), backend, out, str(size), *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return dict(np.load(out))


def test_rccl_with_one_rank(tmp_path):
    """RCCL itself, on the one device every box has ("RCCL path exercised with world size 1"): backend "nccl" with one
    rank, every collective of the multi-GPU forms through `geobo_amd.sharding` on device tensors -- all_gather_into_tensor of fp64 row
    blocks, all_reduce SUM of P_c N doubles, the ranks' agreement (MIN), all_to_all_single -- and a whole row-form step whose
    collectives go through the backend, against the same step without.  Library load, fp64 support, stream ordering between the
    engine's kernels and the communicator, HSA_ENABLE_IPC_MODE_LEGACY=0: all exercised here instead of first on an 8-GPU node."""
    import json
    _release_device_memory()
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = str(tmp_path / "rccl1.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_one_worker.py"), out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = json.load(open(out))
    print(res)
    assert res["backend"] == "nccl" and res["world"] == 1 and res["ranks_reported_by_backend"] == 1
    assert res["all_gather_into_tensor_equal"] and res["all_reduce_sum_equal"] and res["all_reduce_matrix_equal"]
    assert res["all_to_all_single_equal"] and res["agree"] == [True, False]
    assert res["collectives_timed_forced"] == ["xgmi_all_gather", "xgmi_all_reduce"]
    assert res["row_form_step_bit_identical"], res["row_form_step_max_abs_diff"]


@pytest.mark.parametrize("world", [2, 4])
def test_nccl_ranks_match_single_rank(world, tmp_path):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (have %d)" % (world, torch.cuda.device_count()))
    one = _run_ranks(1, "nccl", str(tmp_path / "r1.npz"))
    many = _run_ranks(world, "nccl", str(tmp_path / "rN.npz"))
    assert int(many["world"]) == world
    for a, b in zip(many["cubes"], one["cubes"]):
        assert normwise(a, b) <= 1e-10
    assert abs(float(many["logl"]) - float(one["logl"])) <= 1e-10 * abs(float(one["logl"]))


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher must start N ranks itself and print a line with n_gpus == N."""
    import json
    n = 2
    if torch.cuda.device_count() < n:
        pytest.skip("needs 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--size", "32", "--steps", "1", "--warmup",
                        "1", "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == n and line["config"]["ranks_reported_by_backend"] == n


def test_single_device_gloo_ranks_match_single_rank(tmp_path):
    """Two ranks sharing the one device of this box (gloo transport, device tensors staged through the host by torch): the
    whole multi-rank host path -- shards, collectives, assembly -- through the HIP kernels, against the 1-rank run."""
    one = _run_ranks(1, "gloo", str(tmp_path / "g1.npz"))
    two = _run_ranks(2, "gloo", str(tmp_path / "g2.npz"))
    assert int(two["world"]) == 2
    for a, b in zip(two["cubes"], one["cubes"]):
        assert normwise(a, b) <= 1e-10


@pytest.mark.parametrize("posterior", ["rows", "columns"])
def test_four_gloo_ranks_row_exchange_and_row_gram(posterior, tmp_path):
    """The multi-rank forms of a lattice survey end to end on the one device of this box (gloo transport), against the 1-rank run.
    rows (default, round 3): row-sharded spectral product, row-sharded lattice Gram (AkA by all-gather of row blocks), row-sharded
    transposed posterior (one all-reduce of partial means and sums of squares) -- no all-to-all, no column shard of A K;
    columns (GEOBO_POSTERIOR=dense: the round-2 form, still what fp32 / streamed modes and non-lattice surveys run): the same with
    the all-to-all of A K block-columns and the column-sharded fused reduction.  64 x 48 x 64 is the smallest grid the lattice Gram is
    instantiated for (the fused (row, z)-plane transform of the rows form needs 64^3: the next test)."""
    env = {} if posterior == "rows" else {"GEOBO_POSTERIOR": "dense"}
    one = _run_ranks(1, "gloo", str(tmp_path / "h1.npz"), "64x48x64", env_extra=env)
    four = _run_ranks(4, "gloo", str(tmp_path / "h4.npz"), "64x48x64", env_extra=env)
    assert int(four["world"]) == 4 and bool(four["exchange"]) and bool(four["row_gram"])
    assert bool(four["rowpath"]) == (posterior == "rows")
    for a, b in zip(four["cubes"], one["cubes"]):
        if np.isnan(b).all():
            assert np.isnan(a).all()
        else:
            assert normwise(a, b) <= 1e-10
    assert abs(float(four["logl"]) - float(one["logl"])) <= 1e-10 * abs(float(one["logl"]))


def test_two_gloo_ranks_row_sharded_posterior_three_property_blocks(tmp_path):
    """The row-sharded form with all three property blocks (the drill-core property has no sensor rows of its own: its cubes come from
    the cross blocks K_02, K_12, K_22 alone) on 2 ranks against the 1-rank run, 64 x 48 x 64."""
    env = {"GEOBO_TEST_PROPS": "3"}
    one = _run_ranks(1, "gloo", str(tmp_path / "p1.npz"), "64x48x64", env_extra=env)
    two = _run_ranks(2, "gloo", str(tmp_path / "p2.npz"), "64x48x64", env_extra=env)
    assert int(two["world"]) == 2 and bool(two["rowpath"])
    assert not any(np.isnan(c).all() for c in one["cubes"])
    for a, b in zip(two["cubes"], one["cubes"]):
        assert normwise(a, b) <= 1e-10
    assert abs(float(two["logl"]) - float(one["logl"])) <= 1e-10 * abs(float(one["logl"]))


def test_two_gloo_ranks_row_sharded_posterior_without_drill_rows(tmp_path):
    """No drill constraints (M = 2 Ms exactly: no rows behind the sensor rows, no drill share, no drill term in the mean)."""
    env = {"GEOBO_TEST_DRILL": "0"}
    one = _run_ranks(1, "gloo", str(tmp_path / "d1.npz"), "64x48x64", env_extra=env)
    two = _run_ranks(2, "gloo", str(tmp_path / "d2.npz"), "64x48x64", env_extra=env)
    assert int(two["world"]) == 2 and bool(two["rowpath"])
    for a, b in zip(two["cubes"], one["cubes"]):
        if np.isnan(b).all():
            assert np.isnan(a).all()
        else:
            assert normwise(a, b) <= 1e-10


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_ranks_row_sharded_posterior_at_64(world, tmp_path):
    """64^3 (the bench workload) on 2 and 8 ranks sharing this box's device: the row-sharded posterior with the fused (row, z)-plane
    inverse transform (rows of L^-1 A in the [iy][iz][ix] layout), the drill tile split over the ranks, one all-reduce -- against
    the 1-rank transposed posterior."""
    one = _run_ranks(1, "gloo", str(tmp_path / "s1.npz"), "64")
    many = _run_ranks(world, "gloo", str(tmp_path / "sN.npz"), "64")
    assert int(many["world"]) == world and bool(many["rowpath"]) and bool(many["row_gram"])
    for a, b in zip(many["cubes"], one["cubes"]):
        if np.isnan(b).all():
            assert np.isnan(a).all()
        else:
            assert normwise(a, b) <= 1e-10
    assert abs(float(many["logl"]) - float(one["logl"])) <= 1e-10 * abs(float(one["logl"]))


@pytest.mark.parametrize("assembly,operators,size", [("f32", "streamed", "32"), ("f64", "streamed", "32"), ("f32", "resident", "32"),
                                                     ("f32", "resident", "64x48x64")])
def test_large_cube_modes_on_the_row_exchange_path(assembly, operators, size, tmp_path):
    """BASELINE config 5's modes (fp32 assembly, streamed operators) with 4 ranks: chunked row exchange (each chunk of sensor rows
    transformed, cropped per destination, exchanged by its own all-to-all and written straight into the A K shard), N/G-deep AkA
    panels, sharded posterior -- against the 1-rank run of the same mode (gloo transport on this box's one device)."""
    # 64 x 48 x 64 with fp32 assembly: since round 4 the ROW form carries it (fp32-rounded covariance tables, no A K shard at all); the
    # 32^3 cases stay on the chunked exchange (the planner keeps small grids without fused kernels in the column form)
    one = _run_ranks(1, "gloo", str(tmp_path / "m1.npz"), size, (assembly, operators))
    four = _run_ranks(4, "gloo", str(tmp_path / "m4.npz"), size, (assembly, operators))
    assert int(four["world"]) == 4 and bool(four["exchange"])
    tol = 1e-10 if assembly == "f64" else 2e-6          # fp32 storage: the shards round different partial sums
    for a, b in zip(four["cubes"], one["cubes"]):
        if np.isnan(b).all():
            assert np.isnan(a).all()
        else:
            assert normwise(a, b) <= tol


@pytest.mark.parametrize("size,world,assembly,operators,env", [("64x48x64", 4, "f32", "streamed", {}), ("32", 2, "f64", "resident", {"GEOBO_ROWS": "1"}),
                                                              ("32", 4, "f32", "streamed", {"GEOBO_ROWS": "1", "GEOBO_ROW_CHUNK": "256"})])
def test_row_form_under_real_collectives(size, world, assembly, operators, env, tmp_path):
    """Round 4: the row form with fp32-rounded tables / streamed operators (config 5's modes) and on a grid without fused kernels
    (32^3, forced: batched-GEMM stand-ins, chunked) -- all-gather of AkA row blocks, all-reduce of the partial sums of squares and the
    rank-agreement check over gloo on this box's one device, against the 1-rank run of the same mode."""
    one = _run_ranks(1, "gloo", str(tmp_path / "q1.npz"), size, (assembly, operators), env_extra=env)
    many = _run_ranks(world, "gloo", str(tmp_path / "qN.npz"), size, (assembly, operators), env_extra=env)
    assert int(many["world"]) == world and bool(many["rowpath"]) and bool(one["rowpath"])
    for a, b in zip(many["cubes"], one["cubes"]):
        if np.isnan(b).all():
            assert np.isnan(a).all()
        else:
            assert normwise(a, b) <= 1e-10
    assert abs(float(many["logl"]) - float(one["logl"])) <= 1e-10 * abs(float(one["logl"]))


def test_emulated_ranks_partial_posteriors_add_up():
    """The row-sharded posterior at the bench's own shape, 64^3 on 8 ranks (512 sensor rows and 16 drill-tile rows per rank), one
    emulated rank after the other on this device: the partial sums of squares of the 8 ranks -- what the all-reduce would
    add -- must add up to the 1-rank posterior variance, and every rank's mean (formed whole on each rank) must be the 1-rank mean.  (A K, the row Gram and the transposed posterior of every rank run
    for real; only AkA, which a lone rank cannot assemble, is the 1-rank step's.)"""
    import bench
    from conftest import settings_for
    from geobo_amd.inversion import Inversion
    from geobo_amd.sharding import EmulatedGroup
    G, n = 8, 64
    s = settings_for(n, n, n, kernelfunc="matern32")
    gp_length = np.array([200.0, 202.0, 204.0])

    def run(inv, data, keep=None):
        cap = {}
        orig = inv.engine.posterior

        def wrapped(*a, **k):
            r = orig(*a, **k)
            cap["mu"], cap["var"] = np.array(r["mu"]), np.array(r["var"])
            return r
        inv.engine.posterior = wrapped
        inv.engine.aka_hook = keep
        grav, mag, loc, drill0 = data
        inv.engine.clear_operators()
        inv.gp_length = gp_length.copy()
        inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
        return cap
    inv1 = Inversion(settings=s, props=(0, 1))
    data = bench.synthetic_inputs(inv1, 50)
    kept = {}
    one = run(inv1, data, keep=lambda AkA: kept.__setitem__("AkA", AkA.clone()))
    true_AkA = kept["AkA"]
    del inv1
    torch.cuda.empty_cache()
    mu, var = None, None
    for r in range(G):
        inv = Inversion(settings=s, props=(0, 1), rank=r, world=G, group=EmulatedGroup(r, G))
        inv.sensor_locations = data[2]
        part = run(inv, data, keep=lambda AkA: AkA.copy_(true_AkA))
        assert inv.engine._rowpath and inv.engine._row_gram()
        ok1 = ~np.isnan(one["mu"])
        assert normwise(part["mu"][ok1], one["mu"][ok1]) <= 1e-11       # the mean is formed whole on every rank
        mu = part["mu"]
        var = part["var"] if var is None else var + part["var"]
        del inv
        torch.cuda.empty_cache()
    ok = ~np.isnan(one["mu"])
    assert ok.sum() == 2 * n ** 3 and np.isnan(mu[~ok]).all()
    assert normwise(mu[ok], one["mu"][ok]) <= 1e-11
    # var_r = amp - ss_r: the sum over ranks is (G - 1) amp + the 1-rank variance
    amp = np.median((var[ok] - one["var"][ok]) / (G - 1))
    assert abs(amp - 1.0) <= 1e-9
    assert np.abs(var[ok] - (G - 1) * amp - one["var"][ok]).max() <= 1e-11
    _release_device_memory()


def _release_device_memory():
    """Engines built inside the pytest process leave their workspaces in torch's caching allocator; the tests below start other
    processes on the same device."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def test_emulated_rank_runs_the_sharded_path_alone():
    """tools/emulate_rank.py (per-rank compute measured on one device, step times predicted): rank 0 of 4 of the
    64^3 step through the real engine path with an EmulatedGroup -- row exchange, row-sharded lattice Gram, the true AkA put in
    place for the replicated factorisation -- must run, report the forms it used, and cost less than the 1-rank step."""
    import json
    _release_device_memory()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emulate_rank.py"), "--of", "4", "--steps", "1", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "PREDICTED, NOT MEASURED" in out["what"]
    four = out["ranks"]["4"]
    assert four["row_posterior"] and four["row_gram"]
    assert 0.0 < four["compute_ms_per_step_measured"] < 0.5 * out["one_rank"]["ms_per_step"]
    post1 = sum(v for k, v in out["one_rank"]["stage_ms"].items() if k.startswith("posterior"))
    post4 = sum(v for k, v in four["stage_ms_measured"].items() if k.startswith("posterior"))
    assert post4 < 0.5 * post1
    for v in four["predicted"].values():
        assert v["step_ms"] >= four["compute_ms_per_step_measured"]
