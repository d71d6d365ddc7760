"""Host-side logic that needs no GPU: settings derivation, create_cov's length mutation, drill selection,
sharding arithmetic, API surface."""
import inspect

import numpy as np
import pytest

from conftest import load_golden, settings_for


def test_settings_derivation_matches_reference_formulas():
    from geobo_amd.config_loader import Settings, load
    s = Settings(dict(xmin=0, xmax=3050, ymin=0, ymax=1952, zmax=0, zoff=1, zLcube=800., xNcube=25, yNcube=16, zNcube=16))
    assert (s.xvoxsize, s.yvoxsize, s.zvoxsize) == (122.0, 122.0, 50.0)
    assert s.Nsensor == 400 and s.zmin == -800.0
    assert np.array_equal(s.magneticField, np.array([0, 0, 1]) * 1e-3)
    assert s.c_MILLIGALS_UNITS == 6.673848e-11 * 10000 * 1000.0
    import geobo_amd.config_loader as cl
    load(s)
    assert cl.xvoxsize == 122.0 and cl.kernelfunc == "sparse" and cl.active() is s


def test_yaml_roundtrip(tmp_path):
    import yaml
    from geobo_amd.config_loader import Settings
    f = load_golden("example1.npz")
    import json
    d = json.loads(str(f["settings_json"]))
    p = tmp_path / "s.yaml"
    p.write_text(yaml.safe_dump(d))
    s = Settings.from_yaml(str(p))
    assert (s.xNcube, s.yNcube, s.zNcube, s.kernelfunc) == (25, 16, 16, "sparse")
    assert s.gp_coeff == [1.0, 0.2, 0.2]


def test_create_cov_length_mutation_is_in_place():
    from geobo_amd.engine import create_cov_lengths
    g = load_golden("kat_kernels.npz")
    for key, start in (("mutated_eq", [200., 200., 200.]), ("mutated_20", [200., 300., 200.]), ("mutated_21", [200., 300., 300.])):
        a = np.array(start)
        r = create_cov_lengths(a)
        assert r is a and np.array_equal(a, g[key])


def test_inversion_surface_and_geometry():
    from geobo_amd.inversion import Inversion
    f = load_golden("tiny_exp.npz")
    inv = Inversion(settings=settings_for(10, 8, 6))
    assert np.array_equal(inv.gp_length, [200., 200., 200.]) and inv.gp_amp == 1.0
    vox = inv.create_cubegeometry()
    assert np.array_equal(vox, f["voxelpos"]) and np.array_equal(inv.Edges, f["Edges"])
    assert inv.xxx.shape == (8, 10, 6)
    for name, params in (("cubing", ["gravfield", "magfield", "drillfield", "sensor_locations", "drilldata0"]),
                         ("predict3", ["calclogl", "full_cov"]), ("calc_logl", ["params"]), ("optimize_gp", []), ("create_cubegeometry", [])):
        assert list(inspect.signature(getattr(Inversion, name)).parameters)[1:] == params


def test_drill_selection_and_A_drill():
    from geobo_amd import sensormodel as sm
    from geobo_amd.inversion import Inversion
    f = load_golden("tiny_exp.npz")
    inv = Inversion(settings=settings_for(10, 8, 6))
    inv.create_cubegeometry()
    d0 = f["drilldata0"]
    inv.drilldata0 = d0
    assert np.array_equal(inv._drill_selection(), f["sel"])
    vd = np.vstack([inv.xxx[d0 != 0], inv.yyy[d0 != 0], inv.zzz[d0 != 0]])
    A = sm.A_drill(vd, inv.voxelpos)
    assert A.shape == (5, 480) and np.array_equal(np.flatnonzero(A.sum(0)), f["sel"]) and (A.sum(1) == 1).all()
    assert np.array_equal(sm.drill_index(vd, inv.voxelpos), f["sel"])


def test_edge_axes_roundtrip_and_validation():
    from geobo_amd import sensormodel as sm
    f = load_golden("tiny_exp.npz")
    xe, ye, ze = sm._edge_axes(f["Edges"], 10, 8, 6)
    assert np.array_equal(xe, np.linspace(0, 10, 11) * 100.0) and ze[-1] == 600.0
    bad = f["Edges"].copy()
    bad[0, 3, 4, 2] += 1.0
    with pytest.raises(ValueError):
        sm._edge_axes(bad, 10, 8, 6)


def test_shard_columns_cover_and_align():
    from geobo_amd.sharding import shard_columns
    for n_pad in (512, 6400 + 0, 262144, 32768 + 128):
        n_pad = (n_pad + 127) // 128 * 128
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                c0, c1 = shard_columns(n_pad, world, r)
                assert c0 == prev and c0 % 128 == 0 and c1 % 128 == 0
                prev = c1
            assert prev == n_pad


def test_diagonal_covariance_object():
    from geobo_amd.inversion import DiagonalCovariance
    d = DiagonalCovariance(np.arange(6.0))
    assert d.shape == (6, 6) and np.array_equal(d.diagonal(), np.arange(6.0))
    # the reference's idiom on predict3's second return value (inversion.py:238) -- without building a (3N)^2 array
    assert np.array_equal(np.diag(d), np.arange(6.0)) and np.array_equal(np.diagonal(d), np.arange(6.0))
    assert np.shape(d) == (6, 6)
    with pytest.raises(TypeError):
        np.asarray(d)
    with pytest.raises(TypeError):
        np.diag(d, 1)
    with pytest.raises(TypeError):
        np.sum(d)


def test_vertical_utility_table_equals_the_per_column_formula():
    """acquisition.column_utility (one vectorised reduction) against the reference's per-call expression, C- and F-ordered cubes."""
    from geobo_amd.acquisition import Acquisition
    s = settings_for(7, 5, 6, kappa=1.3, beta=0.4)
    rng = np.random.default_rng(0)
    for order in ("C", "F"):
        mean, var, cost = (np.asarray(rng.random((5, 7, 6)), order=order) for _ in range(3))
        acq = Acquisition(s, mean, var, cost)
        for i0 in range(1, 4):
            for i1 in range(1, 6):
                want = np.sum(mean[i0, i1, :]) + s.kappa * np.sqrt(np.sum(var[i0, i1, :])) - s.beta * np.sum(cost[i0, i1, :])
                assert acq.futility_vertical([i0 + 0.2, i1 - 0.3]) == -want
        assert acq.futility_vertical([0, 3]) == np.inf and acq.futility_vertical([4, 3]) == np.inf
        assert acq.futility_vertical([2, np.inf]) == np.inf


def test_kernel_id_mapping():
    from geobo_amd import hip
    assert hip.kernel_id("exp", False) == 1 and hip.kernel_id("exp", True) == 2
    assert hip.kernel_id("matern32", True) == 4 and hip.kernel_id("sparse", False) == 5
    with pytest.raises(ValueError):
        hip.kernel_id("rbf", False)
    assert hip.pad_m(8242) == 8448 and hip.pad_n(480) == 512


def test_bench_refuses_a_line_for_another_job_size():
    """bench.py --gpus N: under a launcher WORLD_SIZE must equal N; without one it starts N ranks itself (here, without GPUs,
    those ranks stop at the device check -- which shows the relaunch happened)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "refusing" in r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs 2 visible devices" in r.stderr


def test_lattice_gram_form_decision_table():
    """Which form of the lattice Gram a rank may use (engine.lattice_gram_form): the chunked exchange of the large-cube modes
    keeps no full rows, so it must never get the row form (round-2 advisory: 64^3 on 8 ranks with assembly='f32' died in
    gram_rows' Ly % 16 assertion)."""
    from geobo_amd.engine import lattice_gram_form
    from geobo_amd.sharding import shard_columns
    n = 64
    N, Ms, plane = n ** 3, n * n, n * n
    def form(world, rank, exchange, chunked):
        c0, c1 = shard_columns(N, world, rank)
        return lattice_gram_form(exchange, chunked, world, Ms, c0, c1, plane, N)
    assert form(1, 0, False, False) == "columns"
    assert form(2, 1, False, False) == "columns" and form(4, 3, False, False) == "columns"       # 32 / 16 planes per rank
    assert form(8, 0, False, False) is None                                                       # world > 4: GEMM
    assert form(4, 0, True, False) == "rows" and form(8, 5, True, False) == "rows"               # Ms / G = 1024, 512
    for r in range(8):
        assert form(8, r, True, True) is None                                                     # chunked: 8 planes per rank, no rows kept
    assert form(4, 2, True, True) == "columns"                                                    # 16 planes per rank: column form is fine
    c0, c1 = shard_columns(64 * 48 * 64, 4, 1)
    assert lattice_gram_form(True, True, 4, 64 * 48, c0, c1, plane, 64 * 48 * 64) is None         # 12 planes: not a multiple of 16
    assert lattice_gram_form(True, False, 3, 64 * 48, 0, 64 * 16 * 64, plane, 64 * 48 * 64) == "rows"
    assert lattice_gram_form(True, False, 5, 64 * 48 + 5, 0, 0, plane, 1) is None                 # Ms / G not a multiple of 128


def test_route_table():
    """geobo_amd/plan.py: the one pure function behind every shape / rank-count / precision decision of the engine -- every
    BASELINE.json configuration x world in {1, 2, 4, 8}, the environment overrides, and the notes a user gets when a faster family
    steps aside for a shape reason."""
    from geobo_amd.plan import plan_route
    fam = lambda *a, **k: plan_route(*a, **k).family
    # config 1 (16^3, exp) and config 2 (32^3): no fused kernels, below the size where the batched-GEMM forms of the row algorithm pay
    for n in (16, 32):
        for w in (1, 2, 4, 8):
            r = plan_route(n, n, n, world=w)
            assert r.spectral and r.family == "columns" and not r.rows and r.exchange == (w >= 4) and "structured algorithm pays from" in r.note
    assert dict(plan_route(32, 32, 32).kernels)["xz"] == "quad" and dict(plan_route(32, 32, 32, env={"GEOBO_XZ_QUAD": "0"}).kernels)["xz"] == "pair"
    # configs 3 / 4 (64^3 fp64): one rank on the fused kernels with A K materialised, from two ranks sharded by sensor rows
    r = plan_route(64, 64, 64, operators="auto")
    assert r.family == "single" and r.single and not r.rows and r.note == "" and dict(r.kernels) == dict(xz="fold", y="mfma", gram="fused", ss="fused")
    assert dict(plan_route(64, 64, 64, env={"GEOBO_Y_MFMA": "0"}).kernels)["y"] == "toeplitz" and not plan_route(64, 64, 64, env={"GEOBO_Y_MFMA": "0"}).opt("y_mfma")
    assert dict(plan_route(64, 48, 64).kernels)["y"] == "mfma" and dict(plan_route(32, 32, 32).kernels)["y"] == "mfma"
    assert plan_route(64, 64, 64).opts()["z_mul"] and not plan_route(64, 64, 64, env={"GEOBO_Z_MUL": "0"}).opts()["z_mul"]
    for w in (2, 4, 8):
        r = plan_route(64, 64, 64, world=w, rank=w - 1)
        assert r.family == "rows" and r.rows and not r.single and r.exchange and r.exchange_without_rows == (w >= 4)
    assert fam(64, 48, 64) == "single" and fam(64, 48, 64, world=4) == "rows"
    # fused / four-plane (x, z) kernels with another y extent: the row form from 2^17 voxels (measured; the materialised one-rank form
    # keeps the two shapes with a fused lattice Gram); smaller grids stay where they were
    for g in ((64, 32, 64), (64, 80, 64), (64, 128, 64), (48, 64, 64), (64, 64, 32), (32, 128, 32)):
        assert fam(*g) == "rows" and fam(*g, world=4) == "rows", g
    for g in ((64, 32, 32), (48, 32, 64), (32, 64, 32)):
        assert fam(*g) == "columns", g
    assert fam(64, 16, 64) == "single"            # (transposed posterior on the materialised A K with Z by GEMM: 33 ms, row form 35)
    # batched-GEMM (x, z) passes: from 2^18 voxels with 96 x 96 planes, from 393 216 voxels with 64 x 64 planes, or where the column form's
    # A K cannot fit
    for g in ((80, 80, 80), (80, 64, 80), (96, 32, 96), (80, 128, 80)):
        assert fam(*g) == "rows", g
    for g in ((80, 32, 80), (112, 16, 112), (16, 128, 128), (48, 48, 48), (32, 32, 128)):
        assert fam(*g) == "columns", g
    assert fam(16, 512, 128) == "rows" and fam(16, 512, 128, world=8) == "columns"      # 2^20 voxels, 2048-mode planes: only the memory rule
    # fp32 tables / streamed operators no longer force column shards: the row form carries them (config 5's modes at 64^3)
    assert fam(64, 64, 64, world=8, assembly="f32") == "rows" and fam(64, 64, 64, assembly="f32") == "rows"
    assert fam(64, 64, 64, world=4, operators="streamed") == "rows"
    # config 5 (128^3 x 3, fp32 assembly): row form on the batched-GEMM kernels; operator rows are generated per batch where a rank's
    # share cannot be resident
    for w in (1, 2, 4, 8):
        r = plan_route(128, 128, 128, world=w, assembly="f32", operators="auto")
        assert r.family == "rows" and dict(r.kernels) == dict(xz="gemm+axis4", y="mfma", gram="gemm", ss="stored")
        assert (r.operators == "streamed") == (w < 8)
    assert plan_route(128, 128, 128, world=8, assembly="f32", operators="streamed").operators == "streamed"
    assert fam(96, 96, 96) == "rows" and dict(plan_route(96, 96, 96).kernels)["y"] == "mfma"
    assert dict(plan_route(96, 96, 96, env={"GEOBO_Y_MFMA": "0"}).kernels)["y"] == "toeplitz" and dict(plan_route(64, 16, 64).kernels)["y"] == "toeplitz"
    assert dict(plan_route(144, 144, 144).kernels)["y"] == "spectrum"
    # shapes the spectral route does not take, and padded sensor rows
    r = plan_route(25, 16, 16)
    assert not r.spectral and r.family == "columns" and "not multiples of 16" in r.note
    r = plan_route(64, 16, 16 * 20)          # nx * ny = 1024 (fine), planes 64 x 320 ...
    assert r.spectral
    r = plan_route(48, 16, 64)               # nx * ny = 768 = 3 * 256: unpadded
    assert r.spectral and r.family == "columns"
    r = plan_route(16, 24 * 2, 16)           # nx * ny = 768
    assert r.spectral
    r = plan_route(80, 16, 16)               # nx * ny = 1280 = 5 * 256
    assert r.spectral and r.family == "columns"
    r = plan_route(144, 16, 144)             # nx * ny = 2304 = 9 * 256, N = 331776 >= 2^18, planes 144 x 144: rows
    assert r.family == "rows"
    r = plan_route(112, 48, 112)             # nx * ny = 5376 = 21 * 256
    assert r.family == "rows"
    r = plan_route(112, 16, 112, world=1)    # nx * ny = 1792 = 7 * 256 but N = 200704 < 2^18
    assert r.family == "columns" and "here 112 x 16 x 112: 200704 voxels, 12544 modes, no fused" in r.note
    r = plan_route(16, 16, 16 * 1024)        # padded? nx * ny = 256: fine; planes 16 x 16384
    assert r.spectral
    r = plan_route(96, 88 + 8, 96, world=7)  # 9216 sensor rows do not divide over 7 ranks
    assert r.family == "columns" and ("do not divide" in r.note or not r.spectral)
    # environment overrides go INTO the planner
    assert fam(64, 64, 64, env={"GEOBO_POSTERIOR": "dense"}) == "columns"
    assert fam(64, 64, 64, world=4, env={"GEOBO_POSTERIOR": "dense"}) == "columns"
    assert fam(64, 64, 64, world=4, env={"GEOBO_SPECTRAL_EXCHANGE": "0"}) == "columns"
    assert fam(64, 64, 64, world=2, env={"GEOBO_ROWS": "0"}) == "columns"
    assert fam(64, 64, 64, env={"GEOBO_ROWS": "1"}) == "rows" and fam(16, 16, 16, env={"GEOBO_ROWS": "1"}) == "rows"
    r = plan_route(64, 64, 64, env={"GEOBO_XZ_FOLD": "0"})                   # without the radix-2 kernels: no fused reduction, so not the
    assert r.family == "rows" and dict(r.kernels)["xz"] == "fused" and dict(r.kernels)["ss"] == "stored"   # materialised form; the row form on xz2d
    assert fam(64, 64, 64, env={"GEOBO_AKA_LATTICE": "0"}) == "single"
    assert fam(64, 64, 64, world=2, env={"GEOBO_AKA_LATTICE": "0"}) == "columns"
    assert not plan_route(64, 64, 64, method="dense").spectral
    assert "spectral/rows" in plan_route(64, 64, 64, world=8).describe()


def test_planner_and_kernel_wrappers_share_one_instance_table():
    """Round-4 advisory: plan.py carried its own copies of the kernel-instance tables.  They are now defined once (plan.py) and
    re-exported by hip.py / LatticeGram; this pins that."""
    from geobo_amd import hip, plan
    from geobo_amd.lattice_gram import LatticeGram
    assert hip.XZ2D_SHAPES is plan.XZ2D_SHAPES and hip.XZ2D_FOLD_N is plan.XZ2D_FOLD_N and hip.TOEPLITZ_NY is plan.TOEPLITZ_NY and hip.SPECTRAL_Y_NY is plan.SPECTRAL_Y_NY
    assert (hip.PAD_M, hip.PAD_N) == (plan.PAD_M, plan.PAD_N) == (256, 128)
    for dims in ((64, 64, 64), (64, 48, 64), (64, 32, 64), (32, 32, 32), (48, 64, 64), (128, 128, 128), (20, 16, 16)):
        assert LatticeGram.fast(*dims) == plan.lattice_gram_fast(*dims)
        assert LatticeGram.supported(*dims) == plan.lattice_gram_supported(*dims)


def test_every_rank_plans_the_same_route():
    """Round-4 advisory: slab alignment used to be decided from the calling rank's own shard, so e.g. 16^3 on 3, 5, 6 or 7 ranks gave
    some ranks the spectral route and others the dense one.  The decision is over all shards now: one route per (grid, world)."""
    from geobo_amd.plan import plan_route
    for n in range(16, 145, 16):
        for dims in ((n, n, n), (n, 16, n), (16, n, 32)):
            for world in range(1, 9):
                routes = {plan_route(*dims, world=world, rank=r).describe() + str(plan_route(*dims, world=world, rank=r).spectral)
                          for r in range(world)}
                assert len(routes) == 1, (dims, world, routes)


def test_mandatory_row_form_is_flagged():
    """Round-4 advisory: where the column form's A K cannot fit (80 x 128 x 80: 272 GB) the planner says that only the row form
    fits, so that the engine raises with the reason instead of running into the allocator when that form is denied or switched off."""
    from geobo_amd.plan import plan_route
    r = plan_route(80, 128, 80)
    assert r.family == "rows" and r.rows_mandatory and r.ak_bytes > 160 << 30
    off = plan_route(80, 128, 80, env={"GEOBO_ROWS": "0"})
    assert off.family != "rows" and off.rows_mandatory and "only the row form fits" in off.note
    assert not plan_route(64, 64, 64).rows_mandatory and not plan_route(128, 128, 128, world=8, assembly="f32").rows_mandatory
    # round-5 advisory: the figure follows the element size of the assembly and the number of property blocks
    assert plan_route(96, 96, 96).rows_mandatory and not plan_route(96, 96, 96, assembly="f32").rows_mandatory
    assert plan_route(96, 96, 96, assembly="f32").ak_bytes * 2 == plan_route(96, 96, 96).ak_bytes
    assert not plan_route(80, 96, 80).rows_mandatory and plan_route(80, 96, 80, nprops=3).rows_mandatory
