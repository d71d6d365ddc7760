"""Pin the CPU oracle (oracle/geobo_oracle.py) to golden vectors produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py).  CPU only; this is what makes the oracle a trustworthy checker."""
import json

import numpy as np
import pytest

from conftest import load_golden, normwise
from oracle import geobo_oracle as O


def test_F0_kernel_known_answers():
    g = load_golden("kat_kernels.npz")
    assert O.k_cross("exp", 0., 600., 650.) == float(g["k2_0"]) == 0.9984012779544537
    assert O.k_cross("matern32", 0., 600., 650.) == float(g["m2_0"])
    d2 = g["d2_line"]
    assert np.array_equal(O.k_auto("exp", d2, 200.), g["k_exp"])
    assert np.array_equal(O.k_cross("exp", d2, 200., 204.), g["k_exp2"])
    assert np.array_equal(O.k_auto("sparse", d2, 400.), g["k_sp"])
    assert np.array_equal(O.k_cross("sparse", d2, 400., 408.), g["k_sp2"])
    assert np.array_equal(O.k_cross("sparse", d2, 400., 400.), g["k_sp2_eq"])   # equal-length offset 1e-3
    assert np.array_equal(O.k_auto("matern32", d2, 200.), g["k_m"])
    assert np.array_equal(O.k_cross("matern32", d2, 200., 204.), g["k_m2"])
    assert np.array_equal(O.grid_points((3, 2, 4), (10., 20., 5.)), g["points3D"])
    assert np.array_equal(O.sqdist(np.array([[0., 0., 0.], [100., 0., 0.], [100., 250., 75.]])), g["D2"])


@pytest.mark.parametrize("name", O.KERNELS)
def test_F0_create_cov_blocks_and_mutation(name):
    g = load_golden("kat_kernels.npz")
    for tag, gl, w in (("eq", [200., 200., 200.], [1.0, .2, .2]), ("ne", [200., 230., 270.], [.7, .3, .2])):
        gl = np.array(gl)
        c = O.create_cov(g["D2"], gl, w, name)
        r = g["cov_%s_%s" % (tag, name)]
        assert np.array_equal(np.isnan(c), np.isnan(r))      # matern32 is NaN at equal lengths (blocks 0<->2)
        assert np.array_equal(c[~np.isnan(c)], r[~np.isnan(r)])
    for key, start in (("mutated_eq", [200., 200., 200.]), ("mutated_20", [200., 300., 200.]), ("mutated_21", [200., 300., 300.])):
        gl = np.array(start)
        O.create_cov(g["D2"], gl, [1, 1, 1], "exp")
        assert np.array_equal(gl, g[key])


def _grid_for(f, nx, ny, nz, kern, vox=100.0):
    return O.Grid(nx=nx, ny=ny, nz=nz, xmax=vox * nx, ymax=vox * ny, zLcube=vox * nz, kernelfunc=kern)


@pytest.mark.parametrize("kern", ["exp", "sparse", "matern32"])
def test_F1_tiny_noncubic_grid(kern):
    f = load_golden("tiny_%s.npz" % kern)
    G = _grid_for(f, 10, 8, 6, kern)
    loc = f["sensor_locations"]
    assert np.array_equal(G.sensor_locations(), loc)
    assert np.array_equal(G.edges(), f["Edges"])
    assert np.array_equal(np.vstack([v.flatten() for v in G.voxel_centres()]), f["voxelpos"])
    A_g = O.a_sens(G, G.B * 0, loc, G.edges(), "grav")
    A_m = O.a_sens(G, G.B, loc, G.edges(), "magn")
    assert normwise(A_g, f["A_g"]) < 1e-13 and normwise(A_m, f["A_m"]) < 1e-13
    d0 = f["drilldata0"]
    assert np.array_equal(O.drill_selection(d0), f["sel"])
    for dense, tol in ((True, 1e-12), (False, 1e-11)):
        r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], loc, d0, gp_length=f["gp_length_in"].copy(), dense=dense)
        for a, b in zip(r["cubes"], f["cubes"]):
            assert normwise(a, b) < tol
        assert normwise(r["mu"], f["mu"]) < tol and normwise(r["var"], f["var"]) < tol
        assert abs(r["logl"] - float(f["logl"])) < 1e-9 * abs(float(f["logl"]))
        assert np.array_equal(r["gp_length"], f["gp_length_out"])
        assert normwise(r["AkA"], f["AkA"]) < 1e-13
        assert np.array_equal(r["Fs3"], f["Fs3"])


def test_F1_no_drill_rows_gives_nan_drill_cubes():
    f = load_golden("tiny_exp_nodrill.npz")
    G = _grid_for(f, 10, 8, 6, "exp")
    d0 = f["drilldata0"]
    r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    for i in (0, 1, 3, 4):
        assert normwise(r["cubes"][i], f["cubes"][i]) < 1e-11
    assert np.isnan(r["cubes"][2]).all() and np.isnan(f["cubes"][2]).all()


@pytest.mark.parametrize("name,kern", [("cube16_exp", "exp"), ("cube16_matern32", "matern32"), ("cube16_sparse", "sparse")])
def test_F2_cube16_blocked_form(name, kern):
    f = load_golden(name + ".npz")
    G = _grid_for(f, 16, 16, 16, kern)
    d0 = f["drilldata0"]
    r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0, gp_length=f["gp_length_in"].copy())
    errs = [normwise(a, b) for a, b in zip(r["cubes"], f["cubes"]) if not np.isnan(b).all()]
    assert max(errs) < 1e-10, errs
    assert abs(r["logl"] - float(f["logl"])) < 1e-9 * abs(float(f["logl"]))
    assert normwise(r["A_g"].sum(axis=1), f["A_g_rowsum"]) < 1e-12 and normwise(r["A_m"].sum(axis=0), f["A_m_colsum"]) < 1e-12


@pytest.mark.parametrize("name", ["example1", "example2"])
def test_F3_shipped_examples(name):
    f = load_golden(name + ".npz")
    G = O.Grid.from_settings(json.loads(str(f["settings_json"])))
    r = O.cubing(G, f["gravfield"], f["magfield"], f["drillfield"], f["sensor_locations"], f["drilldata0"])
    for a, b, v in zip(r["cubes"], f["cubes"], f["vtk_cubes"]):
        assert normwise(a, b) < 1e-10          # vs the reference re-run in this container
        assert normwise(a, v) < 5e-8           # vs the committed examples/results/*.vtk (re-run itself: <= 3.7e-8)
    assert np.array_equal(r["gp_length"], f["gp_length_out"])


def test_F5_baseline_config1():
    """BASELINE config 1: settings_example1 extents (anisotropic 190.6 x 122 x 50 m voxels), 16^3, 'exp', gp_coeff = 0."""
    f = load_golden("config1_exp16.npz")
    s = json.loads(str(f["settings_json"]))
    assert (s["xNcube"], s["yNcube"], s["zNcube"], s["kernelfunc"], s["gp_coeff"]) == (16, 16, 16, "exp", [0.0, 0.0, 0.0])
    assert (s["xmax"], s["ymax"], s["zLcube"]) == (3050, 1952, 800.0)
    G = O.Grid.from_settings(s)
    d0 = f["drilldata0"]
    r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    for i in (0, 1, 3, 4):
        assert normwise(r["cubes"][i], f["cubes"][i]) < 1e-10
    assert np.isnan(f["cubes"][2]).all() and np.isnan(r["cubes"][2]).all()
    assert abs(r["logl"] - float(f["logl"])) < 1e-9 * abs(float(f["logl"]))


@pytest.mark.parametrize("name,dims,kern,ls", [("illcond_tiny_exp", (10, 8, 6), "exp", 8), ("illcond_tiny_matern32", (10, 8, 6), "matern32", 10),
                                               ("illcond_cube16_matern32", (16, 16, 16), "matern32", 10)])
def test_F6_ill_conditioned_regime(name, dims, kern, ls):
    """Length scales of 8-10 voxels, noise 0.01, amplitude 2 (where optimize_gp goes): cond(AkA) 2.6e6 .. 1.8e7."""
    f = load_golden(name + ".npz")
    nx, ny, nz = dims
    G = O.Grid(nx=nx, ny=ny, nz=nz, xmax=100.0 * nx, ymax=100.0 * ny, zLcube=100.0 * nz, kernelfunc=kern, gp_lengthscale=ls,
               gp_err=(0.01, 0.01, 0.01))
    assert float(f["cond_AkA"]) > 1e6 and float(f["gp_amp"]) == 2.0
    d0 = f["drilldata0"]
    r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0, gp_length=f["gp_length_in"].copy(),
                 gp_amp=2.0, dense=(nx == 10))
    errs = [normwise(a, b) for a, b in zip(r["cubes"], f["cubes"])]
    assert max(errs) < 1e-8, errs
    assert abs(r["logl"] - float(f["logl"])) < 1e-8 * abs(float(f["logl"]))
    assert normwise(np.diag(r["L"]), f["L_diag"]) < 1e-9


def test_F7_calc_logl_and_optimum():
    f = load_golden("optimize_tiny_exp.npz")
    G = _grid_for(f, 10, 8, 6, "exp")
    P3 = O.grid_points((10, 8, 6), (100., 100., 100.))
    loc = f["sensor_locations"]
    A_g = O.a_sens(G, G.B * 0, loc, G.edges(), "grav")
    A_m = O.a_sens(G, G.B, loc, G.edges(), "magn")
    for p, v in zip(f["probe_params"], f["probe_values"]):
        got = O.neg_logl(G, p, P3, A_g, A_m, f["sel"], f["Fs3"])
        assert abs(got - v) <= 1e-9 * abs(v)
    got = O.neg_logl(G, f["opt_x"], P3, A_g, A_m, f["sel"], f["Fs3"])
    assert abs(got - float(f["opt_fun"])) <= 1e-9 * abs(float(f["opt_fun"]))
    assert (f["opt_fun"] <= f["probe_values"]).all()               # the optimum is below every probe


@pytest.mark.parametrize("kern", ["exp", "matern32", "sparse"])
def test_ak_row_fft_equals_direct_contraction(kern):
    """The FFT form used for the independent 64^3 spot checks equals the direct row-times-block product."""
    G = O.Grid(nx=7, ny=5, nz=6, xmax=700., ymax=450., zLcube=660., kernelfunc=kern)
    P3 = O.grid_points((7, 5, 6), (G.sx, G.sy, G.sz))
    D2 = O.sqdist(P3)
    lengths = np.array([200., 204., 230.])
    W = O.weight_matrix([0.7, 0.3, 0.2])
    a = np.random.default_rng(3).standard_normal(G.N)
    for s_, j in ((0, 0), (0, 1), (1, 2), (2, 0)):
        ref = a @ (1.3 * O.k_block(kern, D2, lengths, W, s_, j))
        got = O.ak_row_fft(G, a, kern, lengths, W, s_, j, gp_amp=1.3)
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def test_ak_rows_fft_equals_ak_row_fft():
    """The batched form behind the whole-cube 64 x 48 x 64 goldens: same numbers as the single-row form (and hence as the direct
    contraction, above), forward transform shared between the blocks."""
    G = O.Grid(nx=7, ny=5, nz=6, xmax=700., ymax=450., zLcube=660., kernelfunc="matern32")
    lengths = np.array([200., 204., 230.])
    W = O.weight_matrix([0.7, 0.3, 0.2])
    A = np.random.default_rng(4).standard_normal((5, G.N))
    got = O.ak_rows_fft(G, A, "matern32", lengths, W, 1, (0, 1, 2), gp_amp=1.3, workers=2, batch=2)
    for j in (0, 1, 2):
        for r in range(5):
            ref = O.ak_row_fft(G, A[r], "matern32", lengths, W, 1, j, gp_amp=1.3)
            assert np.abs(got[j][r] - ref).max() <= 1e-14 * np.abs(ref).max()


@pytest.mark.parametrize("name,dims,kern", [("tiny_matern32", (10, 8, 6), "matern32"), ("tiny_sparse", (10, 8, 6), "sparse"),
                                            ("cube16_exp", (16, 16, 16), "exp")])
def test_posterior_fft_matches_the_reference(name, dims, kern):
    """`cubing(fft=True)` -- the form that produced tests/golden/oracle64x48_matern32.npz -- against the reference's own cubes."""
    f = load_golden(name + ".npz")
    G = _grid_for(f, *dims, kern)
    d0 = f["drilldata0"]
    r = O.cubing(G, f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0, gp_length=f["gp_length_in"].copy(),
                 fft=True, workers=2)
    errs = [normwise(a, b) for a, b in zip(r["cubes"], f["cubes"]) if not np.isnan(b).all()]
    assert max(errs) < 1e-10, errs
    assert abs(r["logl"] - float(f["logl"])) < 1e-9 * abs(float(f["logl"]))


def test_F4_forward_model_known_answer():
    f = load_golden("forward_kat.npz")
    G = O.Grid(nx=25, ny=16, nz=16, xmax=3050, ymax=1952, zLcube=800.)
    loc = f["sensor_locations"]
    A_g = O.a_sens(G, G.B * 0, loc, G.edges(), "grav")
    A_m = O.a_sens(G, G.B, loc, G.edges(), "magn")
    assert normwise(A_g @ f["density"], f["gravity_csv"]) < 1e-13
    assert normwise(A_m @ f["magsus"], f["magnetic_csv"]) < 1e-13


def test_synthetic_survey_matches_golden_inputs():
    f = load_golden("tiny_exp.npz")
    G = _grid_for(f, 10, 8, 6, "exp")
    sv = O.synthetic_survey(G, 5)
    assert np.array_equal(sv["drilldata0"], f["drilldata0"])
    assert np.array_equal(sv["rho"], f["rho"])
    assert normwise(sv["gravfield"], f["gravfield"]) < 1e-6   # float32-rounded data


def test_headline_sample_fixture_is_well_formed():
    """tests/golden/oracle64_sample_matern32.npz (make_oracle64_sample.py: the 64^3 headline BY VALUE at a spread sample, oracle only):
    shape of the sample, admissible variances, symmetry of the kept AkA rows, the create_cov length mutation."""
    f = load_golden("oracle64_sample_matern32.npz")
    nx, ny, nz = (int(v) for v in f["dims"])
    N, q = nx * ny * nz, f["voxels"]
    assert (nx, ny, nz) == (64, 64, 64) and q.size >= 3000 and np.array_equal(q, np.unique(q)) and q[0] == 0 and q[-1] == N - 1
    iy, ix, iz = np.unravel_index(q, (ny, nx, nz))
    for a, n in ((iy, ny), (ix, nx), (iz, nz)):
        assert (a == 0).sum() >= 128 and (a == n - 1).sum() >= 128
    assert np.isin(f["sel"], q).all() and f["sel"].size == 50
    assert f["mu"].shape == (2, q.size) and np.isfinite(f["mu"]).all() and (f["var"] > 0).all() and (f["var"] <= 1.0).all()
    assert np.array_equal(f["gp_length_out"], O.mutate_lengths(f["gp_length_in"].copy()))
    rows, vals = f["AkA_rows"], f["AkA_values"]
    sub = vals[:, rows]
    assert np.abs(sub - sub.T).max() <= 1e-12 * np.abs(sub).max()            # AkA is symmetric (the rows were formed independently)
    assert (f["L_diag"] > 0).all() and f["L_diag"].size == 2 * nx * ny + 50
