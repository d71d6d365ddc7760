"""End-to-end parity of the MI355X path against golden vectors produced by the reference.  GPU only.

Tiers (hard part 1):
  T1 solver parity   -- the reference's own A matrices fed to the device pipeline: <= 1e-10 normwise
  T3 end to end      -- operators built on the device too: <= 1e-8 normwise per cube (north-star tolerance)
"""
import json
import gc
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, normwise, settings_for

pytestmark = pytest.mark.gpu

TINY = dict(nx=10, ny=8, nz=6)
TOL_T1 = 1e-10
TOL_T3 = 1e-8   # north_star: posterior cubes within 1e-8 relative fp64 (normwise per cube)


def _inv(s, **kw):
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s, **kw)
    inv.create_cubegeometry()
    return inv


def elementwise_rel(c, r, floor=1e-3):
    """max |c - r| / |r| over the voxels where |r| > floor * max|r| (element-wise relative error is only meaningful away from
    the zero crossings of the posterior mean; the normwise figure covers the rest)."""
    c, r = np.asarray(c), np.asarray(r)
    m = np.abs(r) > floor * np.abs(r).max()
    return float((np.abs(c - r)[m] / np.abs(r)[m]).max()) if m.any() else 0.0


def _check_cubes(cubes, ref, tol, what, tol_elem=None):
    errs, elem = [], []
    for c, r in zip(cubes, ref):
        if np.isnan(r).all():
            assert np.isnan(c).all(), what
            continue
        errs.append(normwise(c, r))
        elem.append(elementwise_rel(c, r))
    print(what, "normwise", " ".join("%.2e" % e for e in errs), "| element-wise rel (|ref| > 1e-3 max)", " ".join("%.2e" % e for e in elem))
    assert max(errs) <= tol, (what, errs)
    # north_star says "1e-8 relative": besides the normwise bound, the element-wise relative error on every voxel that is not
    # near a zero crossing is bounded as well: 10 x the normwise tolerance = 1e-7 for the T3 tier (observed <= 4e-9 on every fixture;
    # the reference's own A_sens cancellation noise, SURVEY section 7 hard part 1, is what an element-wise figure sees first)
    assert max(elem) <= (10 * tol if tol_elem is None else tol_elem), (what, elem)


@pytest.mark.parametrize("name", ["tiny_exp", "tiny_sparse", "tiny_matern32", "tiny_exp_nodrill"])
def test_tiny_grid_end_to_end(name):
    f = load_golden(name + ".npz")
    s = settings_for(**TINY, kernelfunc=name.split("_")[1])
    inv = _inv(s)
    inv.gp_length = f["gp_length_in"].copy()
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert all(c.shape == (8, 10, 6) for c in cubes)
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " T3")
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
    assert np.array_equal(inv.gp_length, f["gp_length_out"])      # create_cov's in-place mutation is observable
    assert np.array_equal(np.isnan(inv.mu_rec), np.isnan(f["mu"]))


@pytest.mark.parametrize("kern", ["exp", "sparse", "matern32"])
def test_tiny_grid_solver_parity_with_reference_operators(kern):
    """T1: feed the reference's A_g/A_m; everything downstream (fused AK, AkA, Cholesky, posterior) on device."""
    from geobo_amd import hip
    from geobo_amd.engine import PosteriorEngine, create_cov_lengths
    f = load_golden("tiny_%s.npz" % kern)
    s = settings_for(**TINY, kernelfunc=kern)
    eng = PosteriorEngine(s)
    def padA(A):
        out = torch.zeros((eng.Ms_pad, eng.N_pad), dtype=torch.float64, device="cuda")
        out[:A.shape[0], :A.shape[1]] = hip.to_dev(A)
        return out
    lengths = create_cov_lengths(f["gp_length_in"].copy())
    y = f["Fs3"]
    ng = f["gravfield"].size
    r = eng.posterior(padA(f["A_g"]), padA(f["A_m"]), f["sel"], y[:ng], y[ng:2 * ng], y[2 * ng:], [float(v) for v in lengths],
                      s.gp_coeff, kern, s.gp_err)
    e_mu, e_var = normwise(r["mu"], f["mu"]), normwise(r["var"], f["var"])
    print("T1 %s mu %.2e var %.2e logl %.3e cond %.2e" % (kern, e_mu, e_var, r["logl"] - float(f["logl"]), float(f["cond_AkA"])))
    assert e_mu <= TOL_T1 and e_var <= TOL_T1
    assert abs(r["logl"] - float(f["logl"])) <= 1e-10 * abs(float(f["logl"]))
    # AkA (lower triangle is what the engine computes) against the reference matrix via L L^T
    L = eng.last["L"]
    Md = f["sel"].size
    rows = np.r_[0:ng, eng.Ms_pad:eng.Ms_pad + ng, 2 * eng.Ms_pad:2 * eng.Ms_pad + Md]
    Lh = torch.tril(L).cpu().numpy()[np.ix_(rows, rows)]
    assert normwise(Lh @ Lh.T, f["AkA"]) <= 1e-12


@pytest.mark.parametrize("method", ["dense", "spectral"])
@pytest.mark.parametrize("name,kern", [("cube16_exp", "exp"), ("cube16_matern32", "matern32"), ("cube16_sparse", "sparse")])
def test_cube16_end_to_end(name, kern, method):
    f = load_golden(name + ".npz")
    s = settings_for(16, 16, 16, kernelfunc=kern)
    inv = _inv(s, method=method)
    assert inv.engine.use_spectral == (method == "spectral")
    inv.gp_length = f["gp_length_in"].copy()
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " T3")
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))


@pytest.mark.parametrize("name", ["example1", "example2"])
def test_shipped_examples(name):
    """The reference's two shipped examples: cubing() inputs captured from run_geobo.py, outputs = reference re-run
    and the committed examples/results/*.vtk cubes."""
    from geobo_amd.config_loader import Settings
    f = load_golden(name + ".npz")
    s = Settings(json.loads(str(f["settings_json"])))
    inv = _inv(s)
    assert np.array_equal(inv.gp_length, f["gp_length_in"])
    cubes = inv.cubing(f["gravfield"], f["magfield"], f["drillfield"], f["sensor_locations"], f["drilldata0"])
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " vs reference re-run")
    # the committed VTKs come from the reference on ITS NumPy / BLAS of years ago: the reference re-run here differs from them by
    # <= 3.7e-8 normwise and 2e-6 element-wise, so this comparison cannot be tighter than the reference is with itself
    _check_cubes(cubes, f["vtk_cubes"], 5e-8, name + " vs committed VTK", tol_elem=5e-6)


@pytest.mark.parametrize("method", ["dense", "spectral"])
def test_baseline_config1_against_the_reference(method):
    """BASELINE.json configs[0]: examples/settings_example1.yaml extents (3050 x 1952 x 800 m -> anisotropic voxels), 16^3,
    'exp', gp_coeff = [0, 0, 0], no drill rows -- the reference's own CPU-runnable case; fixture = its output
    (tests/golden/make_golden.py F5).  The product has no CPU path by design: the plumbing case runs on the HIP path."""
    from geobo_amd.config_loader import Settings
    f = load_golden("config1_exp16.npz")
    s = Settings(json.loads(str(f["settings_json"])))
    assert abs(s.xvoxsize - 190.625) < 1e-12 and s.yvoxsize == 122.0 and s.zvoxsize == 50.0
    inv = _inv(s, method=method)
    assert np.array_equal(inv.gp_length, f["gp_length_in"])
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    _check_cubes(cubes, f["cubes"], TOL_T3, "config 1 (%s)" % method)
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
    assert np.array_equal(inv.gp_length, f["gp_length_out"])


ILLCOND = [("illcond_tiny_exp", (10, 8, 6), "exp", 8), ("illcond_tiny_matern32", (10, 8, 6), "matern32", 10),
           ("illcond_cube16_matern32", (16, 16, 16), "matern32", 10)]


@pytest.mark.parametrize("name,dims,kern,ls", ILLCOND)
def test_ill_conditioned_regime_end_to_end(name, dims, kern, ls):
    """Length scales of 8-10 voxels, noise 0.01, amplitude 2 -- the corner optimize_gp explores; cond(AkA) = 2.6e6 .. 1.8e7,
    300 x the other fixtures.  The path forms L^-1 explicitly where the reference calls solve_triangular twice
    (inversion.py:105,114): this is where the two could part."""
    f = load_golden(name + ".npz")
    nx, ny, nz = dims
    s = settings_for(nx, ny, nz, kernelfunc=kern, gp_lengthscale=ls, gp_err=[0.01, 0.01, 0.01])
    inv = _inv(s)
    inv.gp_length = f["gp_length_in"].copy()
    inv.gp_amp = 2.0
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    print("cond(AkA) = %.2e" % float(f["cond_AkA"]))
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " T3", tol_elem=1e-6)
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
    eng = inv.engine
    Md = f["sel"].size
    rows = np.r_[0:eng.Ms, eng.Ms_pad:eng.Ms_pad + eng.Ms, 2 * eng.Ms_pad:2 * eng.Ms_pad + Md]
    Ld = torch.diagonal(eng.last["L"]).cpu().numpy()[rows]
    assert normwise(Ld, f["L_diag"]) <= 1e-9


@pytest.mark.parametrize("name,dims,kern,ls", ILLCOND[:2])
def test_ill_conditioned_solver_parity_with_reference_operators(name, dims, kern, ls):
    """T1 in the same regime: the reference's A_g / A_m fed in, so only assembly + factorisation + L^-1 + reductions differ."""
    from geobo_amd import hip
    from geobo_amd.engine import PosteriorEngine, create_cov_lengths
    f = load_golden(name + ".npz")
    s = settings_for(*dims, kernelfunc=kern, gp_lengthscale=ls, gp_err=[0.01, 0.01, 0.01])
    eng = PosteriorEngine(s)

    def padA(A):
        out = torch.zeros((eng.Ms_pad, eng.N_pad), dtype=torch.float64, device="cuda")
        out[:A.shape[0], :A.shape[1]] = hip.to_dev(A)
        return out
    lengths = create_cov_lengths(f["gp_length_in"].copy())
    y = f["Fs3"]
    ng = f["gravfield"].size
    r = eng.posterior(padA(f["A_g"]), padA(f["A_m"]), f["sel"], y[:ng], y[ng:2 * ng], y[2 * ng:], [float(v) for v in lengths],
                      s.gp_coeff, kern, s.gp_err, gp_amp=2.0)
    # the prior variance is amp * k(0) = 2 here: var = 2 - sum V^2
    e_mu, e_var = normwise(r["mu"], f["mu"]), normwise(r["var"], f["var"])
    print("T1 %s mu %.2e var %.2e (cond %.2e)" % (name, e_mu, e_var, float(f["cond_AkA"])))
    assert e_mu <= 1e-9 and e_var <= 1e-9
    Md = f["sel"].size
    rows = np.r_[0:ng, eng.Ms_pad:eng.Ms_pad + ng, 2 * eng.Ms_pad:2 * eng.Ms_pad + Md]
    Lh = torch.tril(eng.last["L"]).cpu().numpy()[np.ix_(rows, rows)]
    assert normwise(Lh @ Lh.T, f["AkA"]) <= 1e-12
    assert normwise(np.diag(Lh), f["L_diag"]) <= 1e-10


def test_optimize_gp_reaches_the_reference_optimum():
    """f1 row: calc_logl at fixed hyper-parameters and the SHGO optimum of Inversion.optimize_gp on the tiny grid against a run
    of the reference (fixture F7); one full sweep timed on the GPU path.  The reference stores the bare scalar lengthscale in
    gp_length afterwards (inversion.py:175), which its own create_cov cannot index; this path keeps the 3-vector
    lengthscale * xvoxsize (pinned below)."""
    import time
    f = load_golden("optimize_tiny_exp.npz")
    s = settings_for(**TINY, kernelfunc="exp")
    inv = _inv(s)
    d0 = f["drilldata0"]
    inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    for p, v in zip(f["probe_params"], f["probe_values"]):
        got = inv.calc_logl(p)
        assert abs(got - v) <= 1e-8 * abs(v), (p, got, v)
    t0 = time.perf_counter()
    inv.optimize_gp()
    dt = time.perf_counter() - t0
    x = np.r_[inv.gp_amp, inv.gp_length[0] / s.xvoxsize, inv.coeffm]
    fun = inv.calc_logl(x)
    ref_x, ref_fun = f["opt_x"], float(f["opt_fun"])
    print("optimize_gp: %.1f s; optimum %s objective %.8f (reference %s %.8f)" % (dt, np.round(x, 5), fun, np.round(ref_x, 5), ref_fun))
    assert inv.gp_length.shape == (3,) and np.all(inv.gp_length == inv.gp_length[0])
    # Same objective (the probes above and the reference's optimum below agree to 1e-8), but SHGO's local SLSQP steps use
    # forward differences with h = 1.5e-8: rounding-level differences of the objective (1e-12 relative) move the gradient by
    # ~1e-2, so two correct implementations stop at slightly different points of the flat valley.  Pinned: the optimum found here
    # is at least as good as the reference's, within 0.5 % of its objective; coordinates as below.
    assert abs(inv.calc_logl(ref_x) - ref_fun) <= 1e-8 * abs(ref_fun)
    assert fun <= ref_fun + 1e-6 * abs(ref_fun)
    assert abs(fun - ref_fun) <= 5e-3 * abs(ref_fun)
    # amplitude and length scale within 10 %; the three correlation coefficients sit in the flat part of the valley (a change of
    # 0.05 in the last one moves the objective in the fourth digit): within 0.1 absolute
    assert (np.abs(x[:2] - ref_x[:2]) <= 0.1 * np.abs(ref_x[:2])).all()
    assert (np.abs(x[2:] - ref_x[2:]) <= 0.1).all()
    # the optimised state must be usable (the reference's is not): one more inversion with it
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert np.isfinite(cubes[0]).all()


def test_predict3_covariance_contract():
    """predict3's second return value: np.diag(cov) (the reference's idiom, inversion.py:238) on the diagonal-backed object, and
    the full (3N, 3N) matrix K - V^T V of inversion.py:117 under full_cov=True, against the oracle's reference-shaped form."""
    from oracle import geobo_oracle as O
    f = load_golden("tiny_matern32.npz")
    s = settings_for(**TINY, kernelfunc="matern32")
    inv = _inv(s)
    inv.gp_length = f["gp_length_in"].copy()
    d0 = f["drilldata0"]
    inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    mu, cov, logl = inv.predict3(calclogl=True)
    assert normwise(np.diag(cov), f["var"]) <= 1e-10 and cov.shape == (3 * 480, 3 * 480)
    mu2, full, _ = inv.predict3(calclogl=True, full_cov=True)
    assert isinstance(full, np.ndarray) and full.shape == (1440, 1440)
    assert np.array_equal(mu, mu2) and normwise(np.diag(full), f["var"]) <= 1e-10
    P3 = O.grid_points((10, 8, 6), (100., 100., 100.))
    r = O.posterior_dense(P3, f["A_g"], f["A_m"], f["sel"], f["Fs3"], O.mutate_lengths(f["gp_length_in"].copy()),
                          O.weight_matrix(s.gp_coeff), "matern32", s.gp_err, return_cov=True)
    assert normwise(r["var"], f["var"]) <= 1e-12                       # the oracle's dense form is pinned to the reference
    assert normwise(full, r["cov"]) <= 1e-10
    assert np.abs(full - full.T).max() <= 1e-12 * np.abs(full).max()


def test_reference_api_odds_and_ends():
    """calcDistanceMatrix with a caller's distFunc / 2-D points (the reference accepts both), A_sens with too few sensors
    (IndexError like the reference, not uninitialised rows), the clearer matern32 message in front of the reference's two lines."""
    from geobo_amd import kernels, sensormodel
    pts = np.random.default_rng(1).random((7, 3)) * 100
    D2 = kernels.calcDistanceMatrix(pts)
    D1 = kernels.calcDistanceMatrix(pts, distFunc=lambda d: sum(abs(v) for v in d))
    want = np.abs(pts[None, :, :] - pts[:, None, :]).sum(axis=2)
    assert np.abs(D1 - want).max() <= 1e-12 and np.abs(D2 - ((pts[None] - pts[:, None]) ** 2).sum(2)).max() <= 1e-10
    p2 = pts[:, :2]
    assert np.abs(kernels.calcDistanceMatrix(p2) - ((p2[None] - p2[:, None]) ** 2).sum(2)).max() <= 1e-10
    f = load_golden("tiny_exp.npz")
    s = settings_for(**TINY)
    with pytest.raises(IndexError):
        sensormodel.A_sens(s.magneticField, f["sensor_locations"][:50], f["Edges"], "grav", settings=s)


# "fp32 K at 16^3 gives mu 3e-5 .. 3e-4, sigma^2 1e-7 .. 8e-7 normwise" -> relaxed tolerances
TOL_F32_MU, TOL_F32_VAR = 3e-4, 1e-6


@pytest.mark.parametrize("method", ["dense", "spectral"])
@pytest.mark.parametrize("name,kern", [("cube16_exp", "exp"), ("cube16_matern32", "matern32"), ("cube16_sparse", "sparse")])
def test_fp32_assembly_against_the_reference(name, kern, method):
    """BASELINE config 5's precision mode (fp32 kernel assembly: covariance tables and A K stored in fp32; fp64 accumulation,
    fp64 Cholesky) against the reference's fp64 cubes at the relaxed tolerance SURVEY 8(d) gives for it."""
    f = load_golden(name + ".npz")
    s = settings_for(16, 16, 16, kernelfunc=kern)
    inv = _inv(s, method=method, assembly="f32")
    inv.gp_length = f["gp_length_in"].copy()
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert inv.engine.last["AK"].dtype == torch.float32
    errs = [normwise(c, r) for c, r in zip(cubes, f["cubes"]) if not np.isnan(r).all()]
    k = len(errs) // 2
    print(name, method, "fp32 assembly: mean", " ".join("%.1e" % e for e in errs[:k]), "| var", " ".join("%.1e" % e for e in errs[k:]))
    assert max(errs[:k]) <= TOL_F32_MU and max(errs[k:]) <= TOL_F32_VAR
    assert max(errs) > 1e-9          # it really is the fp32 path
    assert abs(inv.logl - float(f["logl"])) <= 1e-5 * abs(float(f["logl"]))


@pytest.mark.parametrize("assembly", ["f64", "f32"])
def test_streamed_operators_give_the_same_cubes(assembly):
    """operators="streamed": A_g / A_m are generated in row batches (spectral product) and column slabs (AkA) instead of being
    resident -- the same kernels on the same numbers, so the cubes must be bit-identical to the resident run; 32^3 (batched-GEMM
    passes + N-deep AkA panels) and 64 x 48 x 64 (fused kernels + lattice Gram)."""
    import bench
    for dims in ((32, 32, 32), (64, 48, 64)):
        s = settings_for(*dims, kernelfunc="matern32")
        out, inputs = {}, None
        for mode in ("resident", "streamed"):
            inv = _inv(s, props=(0, 1), assembly=assembly, operators=mode)
            if inputs is None:
                inputs = bench.synthetic_inputs(inv, 20)            # one survey for both modes
            else:
                g2, m2, _, _ = bench.synthetic_inputs(inv, 20)      # the streamed generator of the survey itself: same data
                assert np.abs(g2 - inputs[0]).max() <= 1e-6 * np.abs(inputs[0]).max() and np.abs(m2 - inputs[1]).max() <= 1e-6 * np.abs(inputs[1]).max()
            grav, mag, loc, drill0 = inputs
            inv.engine.clear_operators()
            inv.gp_length = np.array([200.0, 202.0, 204.0])
            out[mode] = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
            if mode == "streamed":
                from geobo_amd.engine import StreamedOperator
                assert all(isinstance(v, StreamedOperator) for v in inv.engine._A.values())
            del inv
            torch.cuda.empty_cache()
        for a, b in zip(out["resident"], out["streamed"]):
            if assembly == "f64" and dims[0] == 64:
                assert np.array_equal(a, b, equal_nan=True)
            else:    # AkA panels accumulate in a different order than the split-K slices / one sweep
                assert np.isnan(b).all() if np.isnan(a).all() else normwise(b, a) <= 1e-11


@pytest.mark.parametrize("dims,kernel,props,md,jitter", [((64, 48, 64), "matern32", (0, 1, 2), 20, False),
                                                         ((64, 64, 64), "exp", (0, 1), 0, False),
                                                         ((64, 64, 64), "matern32", (0, 1, 2), 50, False),
                                                         ((64, 16, 64), "sparse", (0, 1, 2), 7, False),
                                                         ((64, 48, 64), "exp", (0, 1), 150, False),      # two drill-row tiles
                                                         ((64, 48, 64), "matern32", (0, 1), 20, True)])
def test_transposed_posterior_matches_the_fused_reduction(dims, kernel, props, md, jitter, monkeypatch):
    """The transposed posterior (round 3: V = (L^-1 A3) K through the covariance kernels, mean as weighted column sums of A K) against
    the fused MFMA reduction over L^-1 (A K) (round 1-2; inversion.py:114-117 literally) on the same step: every form of
    Z = L^-1 A -- fused (row, z)-plane inverse transform (64^3), the two-GEMM inverse (ny = 48), triangular MFMA GEMMs against the
    materialised operator (GEOBO_Z_LATTICE=0; ny = 16; a survey OFF the lattice: sensor heights jittered) -- with and without drill
    rows, two and three property blocks, all three covariance functions."""
    import bench
    s = settings_for(*dims, kernelfunc=kernel)
    forms = [("dense", {"GEOBO_POSTERIOR": "dense"}), ("default", {}), ("two-GEMM inverse", {"GEOBO_Z_FUSED": "0"}), ("GEMM Z", {"GEOBO_Z_LATTICE": "0"})]
    out, seen, nll, inputs = {}, {}, {}, None
    for name, env in forms:
        for k in ("GEOBO_POSTERIOR", "GEOBO_Z_FUSED", "GEOBO_Z_LATTICE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        inv = _inv(s, props=props)
        if inputs is None:
            inputs = bench.synthetic_inputs(inv, md)
            if jitter:
                loc = inputs[2].copy()
                loc[:, 2] += np.random.default_rng(5).uniform(0.0, 3.0, loc.shape[0])      # no two sensors at one height: no lattice plan
                inputs = (inputs[0], inputs[1], loc, inputs[3])
        grav, mag, loc, drill0 = inputs
        inv.engine.clear_operators()
        inv.gp_length = np.array([200.0, 202.0, 204.0])
        inv.engine.kernel_events = []
        out[name] = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
        seen[name] = {e[0] for e in inv.engine.kernel_events if e[0].startswith("posterior")}
        inv.engine.kernel_events = None
        if kernel != "matern32":          # (equal lengths: the Matern cross term is NaN there, as in the reference)
            # likelihood-only evaluation (optimize_gp's objective): A K / AkA in whatever plan the form uses, no posterior
            nll[name] = inv.calc_logl(np.array([1.0, 2.0, 1.0, 0.2, 0.2]))
        del inv
        gc.collect()                     # (engine <-> closure cycles: the workspaces of a 64^3 engine are ~100 GB)
        torch.cuda.empty_cache()
    lattice = dims[1] in (48, 64) and not jitter
    assert seen["dense"] == {"posterior_reduce"}
    assert ("posterior_zlattice" if lattice else "posterior_zgemm") in seen["default"] and "posterior_reduce" not in seen["default"]
    assert "posterior_zgemm" in seen["GEMM Z"] and "posterior_zlattice" not in seen["GEMM Z"]
    for name, v in nll.items():
        assert np.isfinite(v) and abs(v - nll["dense"]) <= 1e-10 * abs(nll["dense"]), (name, v, nll["dense"])
    ref = out["dense"]
    for name, _ in forms[1:]:
        errs = [normwise(a, b) for a, b in zip(out[name], ref) if not np.isnan(b).all()]
        print(dims, kernel, name, " ".join("%.1e" % e for e in errs))
        assert len(errs) == 2 * len(props) and max(errs) <= 1e-11
        for a, b in zip(out[name], ref):
            assert np.isnan(a).all() == np.isnan(b).all()


def test_transposed_posterior_in_an_ill_conditioned_regime():
    """Length scales of 8 voxels, noise 0.01, amplitude 2 (the corner optimize_gp explores; cond(AkA) ~ 1e7 where the other cases have
    ~1e5): the transposed order against the fused reduction on 64 x 48 x 64.  Both are the reference's arithmetic re-associated, so
    they may differ by cond x eps -- the bound is the north-star tolerance of the cubes."""
    import bench
    s = settings_for(64, 48, 64, kernelfunc="exp", gp_err=[0.01, 0.01, 0.01])
    out, inputs = {}, None
    for name in ("dense", "zpath"):
        os.environ.pop("GEOBO_POSTERIOR", None)
        if name == "dense":
            os.environ["GEOBO_POSTERIOR"] = "dense"
        try:
            inv = _inv(s, props=(0, 1))
            if inputs is None:
                inputs = bench.synthetic_inputs(inv, 20)
            grav, mag, loc, drill0 = inputs
            inv.engine.clear_operators()
            inv.gp_amp = 2.0
            inv.gp_length = np.array([800.0, 800.0, 800.0])
            out[name] = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
            if name == "zpath":
                Ld = torch.diagonal(inv.engine.last["L"])[:2 * inv.engine.Ms]
                print("diag(L) range %.2e .. %.2e" % (Ld.min().item(), Ld.max().item()))
        finally:
            os.environ.pop("GEOBO_POSTERIOR", None)
        del inv
        gc.collect()
        torch.cuda.empty_cache()
    errs = [normwise(a, b) for a, b in zip(out["zpath"], out["dense"]) if not np.isnan(b).all()]
    print("ill-conditioned, transposed vs fused:", " ".join("%.1e" % e for e in errs))
    assert len(errs) == 4 and max(errs) <= 1e-8


@pytest.mark.parametrize("dims,md", [((64, 48, 64), 20), ((64, 64, 64), 0)])
def test_symmetric_gram_plan_matches_the_block_column_form(dims, md):
    """A K / AkA as posterior() assembles them for the transposed order (sym: (grav rows, blocks 0 and 1), (magn rows, block 1); AkA's
    lower-left block transposed from the upper-right one) against the block-column form that computes A_m K_10 and correlates it as
    well: the lower triangle the factorisation reads must agree."""
    import bench
    import geobo_amd.engine as E
    s = settings_for(*dims, kernelfunc="matern32")
    inv = _inv(s, props=(0, 1))
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv, md)
    eng = inv.engine
    eng.clear_operators()
    A_g, A_m = eng.operator("grav", loc, B=s.magneticField * 0.), eng.operator("magn", loc, B=s.magneticField)
    sel = np.nonzero(drill0.reshape(-1) != 0)[0] if md else np.zeros(0, dtype=np.int64)
    sel_t = torch.as_tensor(sel, device="cuda") if md else None
    lengths = [float(v) for v in E.create_cov_lengths(np.array([200.0, 202.0, 204.0]))]
    W = E.weight_matrix(s.gp_coeff)
    eng._W = W
    out = {}
    for sym in (False, True):
        assert (not sym) or eng._sym_ok(A_g, A_m)
        AK, M_pad = eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, (0, 1), sym=sym)
        out[sym] = torch.tril(eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, (0, 1))).clone()
    d = (out[True] - out[False]).abs().max().item() / out[False].abs().max().item()
    print(dims, "symmetric plan vs block columns: %.2e" % d)
    assert d <= 1e-13


def test_fp32_assembly_tracks_fp64_at_32_and_the_headline_shape():
    """fp32 assembly vs the fp64 route on the same inputs: 32^3 (config 2's size) and 64^3 x 2 properties (headline shape)."""
    import bench
    for n in (32, 64):
        s = settings_for(n, n, n, kernelfunc="matern32")
        out = {}
        for assembly in ("f64", "f32"):
            inv = _inv(s, props=(0, 1), assembly=assembly)
            grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
            inv.engine.clear_operators()
            inv.gp_length = np.array([200.0, 202.0, 204.0])
            out[assembly] = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
            del inv
            torch.cuda.empty_cache()
        e_mu = max(normwise(out["f32"][i], out["f64"][i]) for i in (0, 1))
        e_var = max(normwise(out["f32"][i], out["f64"][i]) for i in (3, 4))
        print("%d^3 fp32 assembly vs fp64: mean %.2e var %.2e" % (n, e_mu, e_var))
        assert e_mu <= TOL_F32_MU and e_var <= TOL_F32_VAR


def test_config5_sequential_shards_equal_the_single_rank_run(tmp_path):
    """tests/dryrun_config5.py --sequential (all ranks of a row-sharded fp32-assembly run executed on one device: row blocks of AkA
    placed where the all-gather would put them, partial sums of squares added where the all-reduce would) against the same tool
    with one rank: same posterior checksums; and the by-value oracle checks of the tool at a size where they take seconds."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for world in (1, 4):
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "dryrun_config5.py"), "--size", "32", "--world", str(world),
                            "--sequential"], cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res[world] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("sum_abs_mu", "sum_var"):
        a, b = np.array(res[1]["sequential"]["checksums"][key]), np.array(res[4]["sequential"]["checksums"][key])
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max(), (key, a, b)
    c = res[4]["sequential"]["checks"]
    assert c["finite"] and 0.0 < c["var_min"] and c["var_max"] <= 1.0 + 1e-6 and c["rms_residual_grav"] < 0.1 and c["rms_residual_magn"] < 0.1
    for world in (1, 4):
        o = res[world]["oracle_checks"]
        assert o["a_sens_rows_vs_oracle"] <= 1e-10 and o["ak_rows_vs_oracle"] <= 3e-7 and o["aka_entries_rel"] <= 2e-7
        assert o["posterior_mean_abs_over_max"] <= 1e-7 and o["partial_sumsq_abs"] <= 1e-8 and o["posterior_var_abs"] <= 1e-8


def test_config5_rank0_of_8_at_full_size(tmp_path):
    """BASELINE config 5 at its own size: rank 0 of an 8-rank ROW-sharded run of the 128^3 x 3-property inversion (fp32 covariance
    tables, streamed operators, fp64 Cholesky at M_pad = 33024) on this one device, the REAL AkA factorised (the peers' row blocks are
    computed here as well), with the oracle contacts of tests/dryrun_config5.py asserted: forward-operator rows, rows of A K (oracle
    FFT form), entries of AkA, and -- BY VALUE -- the posterior mean and the rank's partial sums of squares at a sample of voxel
    columns (scipy's triangular solve on oracle covariance columns, given the device's L and u); the rank's footprint stays far
    inside one MI355X (288 GB)."""
    import subprocess
    import sys
    import gc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gc.collect()
    torch.cuda.empty_cache()          # the child needs a large share of the device this process has been caching allocations on
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dryrun_config5.py"), "--size", "128", "--world", "8", "--rank", "0"],
                       cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = out["oracle_checks"]
    print("config 5 rank 0 of 8: step %.1f s, peak %.0f GB, stages %s, checks %s" % (
        out["rank_step_seconds"], out["max_memory_allocated_GB"], out["wall_seconds"], c))
    assert out["N_voxels"] == 128 ** 3 and out["M_pad"] == 33024 and out["covariance_tables"] == "fp32-rounded"
    assert c["a_sens_rows_vs_oracle"] <= 1e-10
    assert c["ak_rows_vs_oracle"] <= 3e-7                                    # fp32-rounded covariance tables against the exact covariance
    assert c["aka_entries_rel"] <= 2e-7
    # posterior by value (fp32-table accuracy, SURVEY 8(d): mean 3e-4, variance 1e-6 normwise; the check itself rounds the oracle's
    # covariance columns like the device's tables, so what is left is summation order and the 1-ulp_fp32 flips of the rounding)
    assert c["posterior_mean_abs_over_max"] <= 1e-6, c
    assert c["partial_sumsq_abs"] <= 1e-7, c
    assert out["posterior_finite"]
    assert out["max_memory_allocated_GB"] < 200.0
    assert out["rank_step_seconds"] <= 12.0, out["wall_seconds"]
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "config5_rank0_of_8_from_test.json"), "w") as f:
        json.dump(out, f, indent=1)


@pytest.mark.parametrize("case", ["off_lattice", "inclined_field", "off_lattice_dense"])
def test_irregular_surveys_against_the_oracle(case):
    """Inputs the fast lattice forms must step aside for, end to end against the (pinned) oracle at 16^3: sensors that are NOT on the
    cube's x-y lattice (A_sens by the direct kernel, AkA by the N-deep GEMM) and an inclined magnetic field (odd stencil table:
    no lattice Gram for the magnetic block)."""
    from oracle import geobo_oracle as O
    from conftest import oracle_grid
    kw = dict(kernelfunc="matern32")
    if case == "inclined_field":
        kw.update(XMAG=0.4, YMAG=-0.3, ZMAG=0.85)
    s = settings_for(16, 16, 16, **kw)
    G = oracle_grid(s)
    loc = G.sensor_locations()
    if case.startswith("off_lattice"):
        rng = np.random.default_rng(4)
        loc = loc + np.c_[rng.uniform(-30, 30, loc.shape[0]), rng.uniform(-30, 30, loc.shape[0]), rng.uniform(0, 40, loc.shape[0])]
    e = G.edges()
    A = (O.a_sens(G, G.B * 0., loc, e, "grav"), O.a_sens(G, G.B, loc, e, "magn"))
    rho, chi = O.synthetic_truth(G)
    grav, mag = A[0] @ rho.flatten(), A[1] @ chi.flatten()
    d0 = np.zeros_like(rho)
    sel = np.random.default_rng(2020).choice(rho.size, 12, replace=False)
    d0.reshape(-1)[sel] = rho.reshape(-1)[sel]
    gl = np.array([200.0, 202.0, 204.0])
    ref = O.cubing(G, grav, mag, d0[d0 != 0], loc, d0, gp_length=gl.copy(), A=A)
    inv = _inv(s, method="dense" if case.endswith("dense") else "auto")
    inv.gp_length = gl.copy()
    cubes = inv.cubing(grav, mag, d0[d0 != 0], loc, d0)
    eng = inv.engine
    if case.startswith("off_lattice"):
        assert eng._lattice_plan[1] is None and not any(v is not None for v in eng._lam.values())
    else:
        assert eng._lattice_plan[1] is not None          # lattice form of A_sens also for an inclined field (it is exact for any B)
    _check_cubes(cubes, ref["cubes"], TOL_T3, case)
    assert abs(inv.logl - ref["logl"]) <= 1e-8 * abs(ref["logl"])


def test_props_subset_and_errors():
    f = load_golden("tiny_exp.npz")
    s = settings_for(**TINY, kernelfunc="exp")
    inv = _inv(s, props=(0, 1))
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert np.isnan(cubes[2]).all() and np.isnan(cubes[5]).all()
    for i in (0, 1, 3, 4):
        assert normwise(cubes[i], f["cubes"][i]) <= TOL_T3
    # matern32 with the default (equal) lengths is singular in the reference -> Cholesky failure -> sys.exit(1)
    s2 = settings_for(**TINY, kernelfunc="matern32")
    inv2 = _inv(s2)
    import contextlib
    import io
    buf = io.StringIO()
    with pytest.raises(SystemExit), contextlib.redirect_stdout(buf):
        inv2.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    said = buf.getvalue()
    assert "DISTINCT length scales" in said and said.rstrip().endswith("Change GP parameter settings")
    inv3 = _inv(settings_for(**TINY, kernelfunc="exp"))
    inv3.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    v = inv3.calc_logl([1.0, 2.0, 1.0, 0.2, 0.2])
    assert np.isfinite(v)


# ---- BASELINE config 2 and the 32^3 Matern case: oracle-generated vectors (the reference cannot run 32^3) -----------
@pytest.mark.parametrize("method", ["dense", "spectral"])
@pytest.mark.parametrize("name,kern", [("oracle32_exp", "exp"), ("oracle32_matern32", "matern32")])
def test_cube32_against_oracle_vectors(name, kern, method):
    f = load_golden(name + ".npz")
    s = settings_for(32, 32, 32, kernelfunc=kern)
    inv = _inv(s, method=method)
    inv.gp_length = f["gp_length_in"].copy()
    d0 = np.zeros(32 ** 3)
    d0[f["sel"]] = f["drillvalues"]
    d0 = d0.reshape(32, 32, 32)
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " T3")
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
    assert np.array_equal(inv.gp_length, f["gp_length_out"])


# ---- 64 x 48 x 64: WHOLE-CUBE oracle vectors on the headline's kernel family (tests/golden/make_oracle64.py) -----------------------
@pytest.mark.parametrize("family", ["planner", "rows"])
def test_cube64x48_against_whole_cube_oracle(family, monkeypatch):
    """The smallest grid on which the radix-2 (x, z) transforms, the fused lattice Gram, the fused sum of squares and the lattice form of
    Z = L^-1 A all run -- the kernels of the 64^3 bench -- against six whole cubes from the pinned CPU oracle (`cubing(fft=True)`:
    FFT rows of A K, scipy Cholesky, column-blocked triangular solves; Matern-3/2, 20 drill rows, three property blocks): one hop from
    the device path to the oracle, on the planner's route (`single`) and on the forced row form (what a rank of N > 1 runs)."""
    if family == "rows":
        monkeypatch.setenv("GEOBO_ROWS", "1")
    f = load_golden("oracle64x48_matern32.npz")
    nx, ny, nz = (int(v) for v in f["dims"])
    assert (nx, ny, nz) == (64, 48, 64)
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    inv = _inv(s)
    inv.gp_length = f["gp_length_in"].copy()
    d0 = np.zeros(nx * ny * nz)
    d0[f["sel"]] = f["drillvalues"]
    d0 = d0.reshape(ny, nx, nz)
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    route = inv.engine.route.describe()
    assert inv.engine.step_route == ("rows" if family == "rows" else "single"), (inv.engine.step_route, route)
    assert "xz=fold" in route and "gram=fused" in route and "ss=fused" in route, route
    _check_cubes(cubes, f["cubes"], TOL_T3, "oracle64x48 whole cubes, %s" % route)
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
    assert np.array_equal(inv.gp_length, f["gp_length_out"])
    L = inv.engine._ws.get("AkA")      # holds the factor after the step; rows [grav | magn | drill | padding], 3072 = Ms_pad
    if L is not None:                  # its diagonal against scipy's (M = 6164 rows)
        M = f["L_diag"].size
        assert normwise(torch.diagonal(L)[:M].cpu().numpy(), f["L_diag"]) <= 1e-10


@pytest.mark.parametrize("dims", [(64, 48, 64), (64, 32, 64)])
def test_zero_cross_weight_between_density_and_susceptibility(dims, monkeypatch):
    """Round-5 advisory: with gp_coeff[2] = 0 (no density-susceptibility correlation: a legal prior, the reference's own gravity-only
    case) the cross generators K_01 = K_10 are exactly zero, and the symmetry residual of the three-product y stage was 0 / 0 = NaN,
    which the engine read as "this prior is not symmetric" after the whole step had run.  The step must run, the three-product form
    must be in use, and the cubes must agree with the four-product path; with the drill block decoupled from density as well."""
    import bench
    nx, ny, nz = dims
    for coeff in ([0.3, 0.2, 0.0], [0.0, 0.2, 0.0]):      # (gp_coeff[1] also scales the synthetic susceptibility: keep it non-zero)
        s = settings_for(nx, ny, nz, kernelfunc="matern32", gp_coeff=coeff)
        inv = _inv(s, props=(0, 1))
        grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 20)
        del inv
        outs = []
        for y2s in ("1", "0"):
            monkeypatch.setenv("GEOBO_Y2S", y2s)
            inv = _inv(s, props=(0, 1))
            inv.gp_length = np.array([200.0, 202.0, 204.0])
            inv.engine.kernel_events = []
            outs.append(inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0))
            names = {e[0] for e in inv.engine.kernel_events}
            inv.engine.kernel_events = None
            assert ("kernel:toeplitz_y2s" in names) == (y2s == "1"), names
            del inv
            gc.collect()
            torch.cuda.empty_cache()
        monkeypatch.delenv("GEOBO_Y2S")
        for i in (0, 1, 3, 4):
            assert np.isfinite(outs[0][i]).all()
            assert np.abs(outs[0][i] - outs[1][i]).max() <= 1e-11 * np.abs(outs[1][i]).max()


def test_three_product_y_stage_against_the_four_product_one_and_its_symmetry_check(monkeypatch):
    """The two-term rows of the transposed posterior take K_10 = K_01 for granted (three y-stage products per mode instead of four,
    geobo_toeplitz_y2s): (i) the cubes agree with the four-product path (GEOBO_Y2S=0) far inside the parity tolerance, (ii) the route
    really runs the three-product kernel, (iii) the symmetry is VERIFIED on the device -- a prior whose blocks (0, 1) and (1, 0)
    differ makes the step raise instead of returning cubes of a different model."""
    f = load_golden("oracle64x48_matern32.npz")
    nx, ny, nz = (int(v) for v in f["dims"])
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    d0 = np.zeros(nx * ny * nz)
    d0[f["sel"]] = f["drillvalues"]
    d0 = d0.reshape(ny, nx, nz)

    def run(inv):
        inv.gp_length = f["gp_length_in"].copy()
        return inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)

    inv = _inv(s)
    inv.engine.kernel_events = []
    three = run(inv)
    names = {e[0] for e in inv.engine.kernel_events}
    inv.engine.kernel_events = None
    assert "kernel:toeplitz_y2s" in names and "kernel:toeplitz_y2t" not in names, names
    monkeypatch.setenv("GEOBO_Y2S", "0")
    four = run(_inv(s))
    monkeypatch.delenv("GEOBO_Y2S")
    for a, b in zip(three, four):
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()
    # an asymmetric "prior": the generator of block (1, 0) scaled by (1 + 1e-6) behind the planner's back
    inv2 = _inv(s)
    run(inv2)                                       # builds the spectral product
    sp = inv2.engine._spectral
    orig = sp.y2s_tables
    sp.y2s_tables = lambda tg, tm, pair=(0, 1): orig(tg, [tm[0] * (1.0 + 1e-6)] + list(tm[1:]), pair)
    with pytest.raises(RuntimeError, match="not symmetric"):
        run(inv2)


def _engine_posterior(eng, f, kern, props=(0, 1, 2)):
    from geobo_amd.engine import create_cov_lengths
    A_g = eng.operator("grav", f["sensor_locations"])
    A_m = eng.operator("magn", f["sensor_locations"])
    ng = f["gravfield"].size
    z = lambda v: (v - v.mean()) / v.std()
    dv = f["drillvalues"]
    y_d = z(dv) if dv.size else dv
    lengths = create_cov_lengths(f["gp_length_in"].copy())
    return eng.posterior(A_g, A_m, f["sel"], z(f["gravfield"]), z(f["magfield"]), y_d, [float(v) for v in lengths],
                         eng.s.gp_coeff, kern, eng.s.gp_err, props=props)


def test_column_shards_reproduce_the_unsharded_posterior():
    """Multi-GPU partition math on one device: run the rank-0/1/2 shards of a 3-way split one after the other, sum their
    partial AkA by hand (what the all-reduce does) and check AkA and the assembled mu/var against the unsharded run."""
    import geobo_amd.engine as E
    from geobo_amd import hip
    from geobo_amd.sharding import assemble_columns, shard_columns
    f = load_golden("oracle32_matern32.npz")
    s = settings_for(32, 32, 32, kernelfunc="matern32")
    full = E.PosteriorEngine(s, method="dense")
    ref = _engine_posterior(full, f, "matern32")
    L_ref = torch.tril(full.last["L"]).clone()
    world = 3
    engs = [E.PosteriorEngine(s, rank=r, world=1, method="dense") for r in range(world)]
    parts, AKs = [], []
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(f["gp_length_in"].copy())]
    sel_t = torch.as_tensor(f["sel"], device="cuda")
    total = None
    for r, eng in enumerate(engs):
        eng.c0, eng.c1 = shard_columns(eng.N_pad, world, r)      # this engine plays rank r of a 3-way split
        eng.nc = eng.c1 - eng.c0
        A_g, A_m = eng.operator("grav", f["sensor_locations"]), eng.operator("magn", f["sensor_locations"])
        AK, M_pad = eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, (0, 1, 2))
        AkA = torch.zeros((M_pad, M_pad), dtype=torch.float64, device="cuda")
        for s_, A in ((0, A_g), (1, A_m)):
            hip.gemm_nt(AK[:, s_ * eng.nc:(s_ + 1) * eng.nc], A[:, eng.c0:eng.c1], AkA[:, s_ * eng.Ms_pad:(s_ + 1) * eng.Ms_pad])
        total = AkA if total is None else total + AkA
        AKs.append(AK)
    # finish exactly like PosteriorEngine._assemble_AkA after the all-reduce
    eng = engs[0]
    off_d, Md = 2 * eng.Ms_pad, f["sel"].size
    dvec = torch.ones(total.shape[0], dtype=torch.float64, device="cuda")
    dvec[0:eng.Ms] = 0.01
    dvec[eng.Ms_pad:eng.Ms_pad + eng.Ms] = 0.01
    total[:off_d, off_d:off_d + Md] = total[off_d:off_d + Md, :off_d].t()
    rows = tuple(c[sel_t] for c in eng.grid_points())
    hip.k_block(hip.kernel_id("matern32", False), rows, rows, lengths[2], lengths[2], 1.0, 1.0, total[off_d:off_d + Md, off_d:off_d + Md])
    dvec[off_d:off_d + Md] = 0.01
    total.diagonal().add_(dvec)
    Linv, info = hip.potrf_inv(total)
    assert int(info.item()) == 0
    assert (torch.tril(total) - L_ref).abs().max().item() <= 1e-11 * L_ref.abs().max().item()
    y = full._pad_y(*[(v - v.mean()) / v.std() for v in (f["gravfield"], f["magfield"], f["drillvalues"])], total.shape[0])
    u, _ = hip.trmv_stats(Linv, y, total)
    mus, vars_ = [], []
    for AK in AKs:
        m, v = hip.posterior_reduce(Linv, AK, u, 1.0)
        mus.append(m)
        vars_.append(v)
    mu = assemble_columns(mus, (0, 1, 2), eng.N, eng.N_pad, world)
    var = assemble_columns(vars_, (0, 1, 2), eng.N, eng.N_pad, world)
    assert normwise(mu, ref["mu"]) <= 1e-10 and normwise(var, ref["var"]) <= 1e-10


def test_calc_logl_matches_oracle_and_handles_failure():
    from oracle import geobo_oracle as O
    f = load_golden("tiny_exp.npz")
    s = settings_for(**TINY, kernelfunc="exp")
    inv = _inv(s)
    d0 = f["drilldata0"]
    inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    params = [1.3, 1.7, 0.8, 0.25, 0.3]
    got = inv.calc_logl(params)
    G = O.Grid(nx=10, ny=8, nz=6, xmax=1000, ymax=800, zLcube=600., kernelfunc="exp")
    P3 = O.grid_points((10, 8, 6), (100., 100., 100.))
    lengths = O.mutate_lengths(params[1] * np.array([100., 100., 100.]))
    r = O.posterior_dense(P3, f["A_g"], f["A_m"], f["sel"], f["Fs3"], lengths, O.weight_matrix(params[2:]), "exp", G.gp_err,
                          gp_amp=params[0])
    want = 0.5 * (r["u"] @ r["u"] + np.log(np.diag(r["L"]) ** 2).sum())     # inversion.py:147-149 (no N log 2pi)
    assert abs(got - want) <= 1e-9 * abs(want)
    inv2 = _inv(settings_for(**TINY, kernelfunc="matern32"))
    inv2.cubing.__func__  # noqa: B018  (surface exists)
    inv2.gravfield, inv2.magfield, inv2.drillfield = inv.gravfield, inv.magfield, inv.drillfield
    inv2.sensor_locations, inv2.drilldata0, inv2._sel, inv2.Fs3 = inv.sensor_locations, inv.drilldata0, inv._sel, inv.Fs3
    assert inv2.calc_logl([1.0, 2.0, 1.0, 0.2, 0.2]) == np.inf               # singular Matern -> inf, not an exception


def test_full_size_64cube_properties():
    """BASELINE's full size (64^3, Matern-3/2, 50 drill rows, 2 property blocks): size-independent properties.
       * 0 <= var <= prior variance (1) everywhere;
       * the posterior mean reproduces the data to the noise level: A3 mu ~ y (residual consistent with sigma = 0.1);
       * drill rows: mu at drilled voxels of block 2 is not computed here (P_c = 2) -> NaN pattern as documented;
       * a CPU-oracle spot check on 64 voxel columns (same check bench.py runs)."""
    import bench
    from geobo_amd.config_loader import Settings
    from geobo_amd.inversion import Inversion
    n = 64
    s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc="matern32"))
    inv = Inversion(settings=s, props=(0, 1), operators="resident")   # method="auto" -> spectral route at 64^3; the operator tensors are read below
    assert inv.engine.use_spectral
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
    inv.gp_length = np.array([200.0, 202.0, 204.0])
    cubes = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    N = n ** 3
    var = inv.cov_rec.diagonal()
    assert np.isnan(var[2 * N:]).all() and np.isnan(inv.mu_rec[2 * N:]).all()
    v = var[:2 * N]
    assert v.min() > 0.0 and v.max() <= 1.0 + 1e-12
    eng = inv.engine
    A_g, A_m = inv._operators()
    pad = lambda x: torch.cat([torch.as_tensor(x, device="cuda"), torch.zeros(eng.N_pad - N, dtype=torch.float64, device="cuda")])
    rg = (A_g @ pad(inv.mu_rec[:N]))[:eng.Ms].cpu().numpy() - inv.Fs3[:eng.Ms]
    rm = (A_m @ pad(inv.mu_rec[N:2 * N]))[:eng.Ms].cpu().numpy() - inv.Fs3[eng.Ms:2 * eng.Ms]
    # with sigma = 0.1 on unit-variance data the residual of a well-posed GP fit is small compared with the data
    assert np.sqrt(np.mean(rg ** 2)) < 0.1 and np.sqrt(np.mean(rm ** 2)) < 0.1
    cb, (cols, smp) = bench.cpu_baseline(inv, [float(x) for x in inv.gp_length], target_seconds=3.0)
    iy_s = cols // (n * n)
    assert iy_s.min() == 0 and iy_s.max() == n - 1 and np.isin(inv._sel[:5], cols).all()      # spread: padded slabs, drilled voxels
    assert np.abs(smp[0][0] - inv.mu_rec[cols]).max() <= 1e-10 * np.abs(inv.mu_rec[:N]).max()
    assert np.abs(smp[0][1] - var[cols]).max() <= 1e-10
    assert np.abs(smp[1][0] - inv.mu_rec[N + cols]).max() <= 1e-10 * np.abs(inv.mu_rec[N:2 * N]).max()
    assert np.abs(smp[1][1] - var[N + cols]).max() <= 1e-10
    assert all(c.shape == (n, n, n) for c in cubes)

    # ---- independent spot checks: nothing below borrows A, A K, AkA or L from the device -------------------------------------
    from oracle import geobo_oracle as O
    from conftest import oracle_grid
    G = oracle_grid(s)
    lengths = np.array([float(x) for x in inv.gp_length])
    W = O.weight_matrix(s.gp_coeff)
    edges = G.edges()
    # (a) forward operators: 8 sensors incl. the first / last sensor row (voxel slabs iy = 0 and ny-1 carry the 1e6 padding)
    sens = [0, 37, n * 31 + 5, n * 32 + 40, n * (n - 1) + 9, n * n - 1, n * 17 + 63, n * 63]
    Ag_o = O.a_sens(G, G.B * 0., loc, edges, "grav", rows=sens)
    Am_o = O.a_sens(G, G.B, loc, edges, "magn", rows=sens)
    Ag_d, Am_d = A_g[sens, :N].cpu().numpy(), A_m[sens, :N].cpu().numpy()
    e_g = np.abs(Ag_d - Ag_o).max() / np.abs(Ag_o).max()
    e_m = np.abs(Am_d - Am_o).max() / np.abs(Am_o).max()
    print("64^3 A_sens rows vs oracle: grav %.2e magn %.2e" % (e_g, e_m))
    assert e_g <= 1e-10 and e_m <= 1e-12                                     # the T2 tier
    # (b), (c): rows of A K, entries of AkA and the factor -- on the resident-operator engine and then, again, on the engine the
    # benchmark times: the default operators="auto" (no operator materialised; the forward transform reads windows of the stencil
    # table, AkA from the lattice Gram).  Its cubes must equal the resident run's bit for bit.
    _independent_checks_64(inv, s, G, Ag_o, Am_o, sens, lengths, W, "resident")
    res_mu, res_var = inv.mu_rec.copy(), var.copy()
    del A_g, A_m
    inv_auto = Inversion(settings=s, props=(0, 1))
    assert inv_auto.engine.auto_ops and inv_auto.engine.use_spectral
    inv_auto.gp_length = np.array([200.0, 202.0, 204.0])
    cubes_auto = inv_auto.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    from geobo_amd.engine import StreamedOperator
    ops = inv_auto._operators()
    assert all(isinstance(o, StreamedOperator) and o.lattice is not None for o in ops), "auto did not choose implicit operators at 64^3"
    # (A K and AkA ran on the implicit operators; the transposed posterior path materialises copies for its L^-1 A products)
    assert inv_auto.engine._edgeV and all(k in inv_auto.engine._lam for k in ("grav", "magn"))
    for i in (0, 1, 3, 4):
        assert np.array_equal(cubes_auto[i], cubes[i]), "cube %d: operators='auto' differs from operators='resident'" % i
    assert np.array_equal(inv_auto.mu_rec[:2 * N], res_mu[:2 * N]) and np.array_equal(inv_auto.cov_rec.diagonal()[:2 * N], res_var[:2 * N])
    assert inv_auto.logl == inv.logl
    _independent_checks_64(inv_auto, s, G, Ag_o, Am_o, sens, lengths, W, "auto")


def test_headline_64cube_against_the_independent_oracle_sample():
    """BASELINE config 3 itself (64^3, Matern-3/2, 50 drill rows, two property blocks) BY VALUE at 3078 voxels spread over the cube -- every
    131st voxel, all drilled voxels, 256 voxels of each 1e6-padded slab iy = 0 / ny-1, the x and z faces, the eight corners -- against
    golden values in which nothing comes from the device (tests/golden/make_oracle64_sample.py: the oracle's operators, FFT rows of
    A K, scipy Cholesky, V on the sampled columns).  Round-5 review, item 5c: the full-size value check was a contiguous block at the
    cube's centre computed with the device's own A and L."""
    from geobo_amd.config_loader import Settings
    from geobo_amd.inversion import Inversion
    f = load_golden("oracle64_sample_matern32.npz")
    nx, ny, nz = (int(v) for v in f["dims"])
    n = nx
    s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc="matern32"))
    d0 = np.zeros(nx * ny * nz)
    d0[f["sel"]] = f["drillvalues"]
    d0 = d0.reshape(ny, nx, nz)
    N = nx * ny * nz
    q = f["voxels"]
    iy = q // (nx * nz)
    assert (iy == 0).sum() >= 256 and (iy == ny - 1).sum() >= 256 and np.isin(f["sel"], q).all() and 0 in q and N - 1 in q
    for kw, what in ((dict(), "auto"), (dict(operators="resident"), "resident")):
        inv = Inversion(settings=s, props=(0, 1), **kw)
        inv.gp_length = f["gp_length_in"].copy()
        cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
        var = inv.cov_rec.diagonal()
        for j in (0, 1):
            e_mu = normwise(inv.mu_rec[j * N + q], f["mu"][j])
            e_var = float(np.abs(var[j * N + q] - f["var"][j]).max())
            print("64^3 [%s] block %d vs the independent oracle sample: mean %.2e (normwise), variance %.2e (abs)" % (what, j, e_mu, e_var))
            assert e_mu <= 1e-8 and e_var <= 1e-8
            assert elementwise_rel(inv.mu_rec[j * N + q], f["mu"][j]) <= 1e-7
            # the cubes carry the data scaling of inversion.py:240-245
            assert normwise(cubes[j].reshape(-1)[q], f["mu"][j] * f["data_std"][j]) <= 1e-8
            assert normwise(cubes[3 + j].reshape(-1)[q], f["var"][j] * f["data_std"][j] ** 2) <= 1e-8
        assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))
        assert np.array_equal(inv.gp_length, f["gp_length_out"])
        L = inv.engine.last["L"]
        M = f["L_diag"].size
        rows = np.r_[0:inv.engine.Ms, inv.engine.Ms_pad:inv.engine.Ms_pad + inv.engine.Ms, 2 * inv.engine.Ms_pad:2 * inv.engine.Ms_pad + f["sel"].size]
        assert rows.size == M and normwise(torch.diagonal(L).cpu().numpy()[rows], f["L_diag"]) <= 1e-10
        del inv
        gc.collect()
        torch.cuda.empty_cache()


def _independent_checks_64(inv, s, G, Ag_o, Am_o, sens, lengths, W, what):
    """Oracle contact that needs no operator tensor: rows of A K through the oracle's FFT form, entries of AkA as (L L^T) of the device
    against oracle operator rows x oracle A K rows, and || L L^T - AkA || with AkA re-assembled by the engine's own route."""
    from oracle import geobo_oracle as O
    eng = inv.engine
    N = eng.N
    L = eng.last["L"]
    Lt = torch.tril(L)
    sel = inv._sel
    off = {0: 0, 1: eng.Ms_pad, 2: 2 * eng.Ms_pad}
    AK = eng.last["AK"] if eng.last["AK"] is not None else eng.last["AK_partial"]
    for s_, A_o, gs in ((0, Ag_o, 0.1), (1, Am_o, 0.1)):
        for k in (1, 4):
            r = sens[k]
            w = {j: O.ak_row_fft(G, A_o[k], "matern32", lengths, W, s_, j) for j in (0, 1, 2)}
            for jj, j in enumerate((0, 1)):                                  # rows of A K (the spectral product, P_c = 2)
                if eng._ak_sym and (s_, j) == (1, 0):
                    continue     # not assembled: AkA's lower-left block is the transpose of (grav rows, magn columns) -- checked below
                got = AK[off[s_] + r, jj * N:(jj + 1) * N].cpu().numpy()
                e = np.abs(got - w[j]).max() / np.abs(w[j]).max()
                print("64^3 [%s] A K row %d block (%d,%d) vs oracle: %.2e" % (what, r, s_, j, e))
                assert e <= 1e-10, (s_, r, j)      # observed 3e-12 (gravity rows: the operator's own 1.5e-11) / 5e-15 (magnetic)
            want = np.r_[Ag_o @ w[0], Am_o @ w[1], w[2][sel]]
            cols = np.r_[np.array(sens), eng.Ms_pad + np.array(sens), 2 * eng.Ms_pad + np.arange(sel.size)]
            got = (Lt[off[s_] + r] @ Lt[cols].t()).cpu().numpy()            # (L L^T)[row, cols]
            want[(cols == off[s_] + r)] += gs ** 2                           # sigma^2 on the diagonal
            e = np.abs(got - want).max() / np.abs(want).max()
            print("64^3 [%s] AkA row %d block %d vs oracle: %.2e" % (what, r, s_, e))
            assert e <= 1e-11                                                # observed <= 8e-14
    # the factor itself: || tril(L) tril(L)^T - AkA || / || AkA || on the device (AkA re-assembled from the resident A K)
    Lc = Lt.clone()
    M_pad = L.shape[0]
    sel_t = torch.as_tensor(sel, device="cuda")
    A_g, A_m = inv._operators()
    AkA2 = eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, [float(x) for x in lengths], "matern32", 1.0, s.gp_err, (0, 1))
    AkA2 = torch.tril(AkA2)
    R = torch.tril(Lc @ Lc.t()) - AkA2
    e = (torch.linalg.matrix_norm(R) / torch.linalg.matrix_norm(AkA2)).item()
    print("64^3 [%s] ||L L^T - AkA||_F / ||AkA||_F = %.2e" % (what, e))
    assert e <= 1e-13


def test_spectral_y_slab_shards_match_dense(monkeypatch):
    """Column shards of the spectral route without the row exchange (replicated forward passes, y-slab cropping in the
    backward pass: GEOBO_SPECTRAL_EXCHANGE=0, also the fallback for uneven shards) against the dense AK, rank by rank."""
    import geobo_amd.engine as E
    monkeypatch.setenv("GEOBO_SPECTRAL_EXCHANGE", "0")
    f = load_golden("oracle32_matern32.npz")
    s = settings_for(32, 32, 32, kernelfunc="matern32")
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(f["gp_length_in"].copy())]
    sel_t = torch.as_tensor(f["sel"], device="cuda")
    world = 4
    for r in (0, 3):
        out = {}
        for method in ("dense", "spectral"):
            eng = E.PosteriorEngine(s, rank=r, world=world, method=method)
            assert eng.use_spectral == (method == "spectral")
            A_g, A_m = eng.operator("grav", f["sensor_locations"]), eng.operator("magn", f["sensor_locations"])
            out[method], _ = eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, (0, 1, 2))
        d = (out["dense"] - out["spectral"]).abs().max().item()
        assert d <= 1e-12 * out["dense"].abs().max().item(), (r, d)


def test_spectral_transform_matrices_diagonalise_toeplitz():
    """CPU-side identity behind the spectral route: crop[G^T diag(E k / P) G] equals the symmetric Toeplitz matrix of k."""
    from geobo_amd.spectral import eigen_matrix, forward_matrix
    n = 16
    k = np.exp(-0.3 * np.arange(n)) * (1 + 0.1 * np.arange(n))
    G, Em = forward_matrix(n), eigen_matrix(n)
    T = G.T @ np.diag(Em @ k / (2 * n)) @ G
    ref = k[np.abs(np.arange(n)[:, None] - np.arange(n)[None, :])]
    assert np.abs(T - ref).max() < 1e-13


@pytest.mark.parametrize("name,sub", [("example1", "synthetic"), ("example2", "sample")])
def test_yaml_workflow_from_raw_files(name, sub, tmp_path):
    """f3 row end to end: YAML + the reference's raw GeoTIFF / CSV inputs -> ingestion -> GPU inversion -> VTK cubes,
    compared with the cubes the reference produced from the same files."""
    import os
    import yaml
    from conftest import GOLDEN
    from geobo_amd import dataio, run_geobo
    f = load_golden(name + ".npz")
    d = json.loads(str(f["settings_json"]))
    d.update(inpath=os.path.join(GOLDEN, "data", sub) + "/", outpath=str(tmp_path) + "/", bayesopt_vertical=True,
             bayesopt_nonvertical=False)
    y = tmp_path / "settings.yaml"
    y.write_text(yaml.safe_dump(d))
    out = run_geobo.run(str(y))
    names = ["cube_density", "cube_magsus", "cube_drill", "cube_density_variance", "cube_magsus_variance", "cube_drill_variance"]
    _check_cubes([out[n] for n in names], f["cubes"], TOL_T3, name + " workflow")
    for n, ref in zip(names, f["cubes"]):
        cube, _, _ = dataio.read_vtkcube(os.path.join(str(tmp_path), n + ".vtk"))
        assert normwise(cube, ref) <= TOL_T3
    assert os.path.exists(os.path.join(str(tmp_path), "newdrill_proposals_vertical.csv"))
    assert len(out["proposals_vertical"]) >= 1 and np.isfinite(out["proposals_vertical"]["BO_GAIN"]).all()
    # f4 by value (run_geobo.py:175-235): the reference's committed proposals (its SHGO local optima, NORTHING / EASTING / BO_GAIN to four
    # decimals) against the utility of the cubes THIS run produced on the device, at the tolerance of the CPU tier
    # (tests/test_dataio_cpu.py: 6e-5); and the best proposal of this run is the reference's best (same gain, one of its locations)
    import pandas as pd
    from geobo_amd.acquisition import Acquisition
    from geobo_amd.config_loader import Settings
    st = Settings(d)
    acq = Acquisition(st, out["cube_drill"], out["cube_drill_variance"])
    ref = pd.read_csv(os.path.join(GOLDEN, "data", "results_cylinders" if name == "example1" else "results_sample",
                                   "newdrill_proposals_vertical.csv"))
    for _, r in ref.iterrows():
        i0 = (r.NORTHING - st.ymin - 0.5 * st.yvoxsize) / st.yvoxsize
        i1 = (r.EASTING - st.xmin - 0.5 * st.xvoxsize) / st.xvoxsize
        assert abs(-acq.futility_vertical([i0, i1]) - r.BO_GAIN) <= 6e-5, (r.NORTHING, r.EASTING)
    got = pd.DataFrame(out["proposals_vertical"])
    best, top = ref.BO_GAIN.max(), got.iloc[int(np.argmax(got.BO_GAIN.values))]
    assert abs(top.BO_GAIN - best) <= 6e-5
    at_best = ref[np.abs(ref.BO_GAIN - best) <= 6e-5]
    assert ((np.abs(at_best.NORTHING - top.NORTHING) < 1e-6) & (np.abs(at_best.EASTING - top.EASTING) < 1e-6)).any()


def test_row_sharded_exchange_matches_dense_columns():
    """Multi-GPU row-sharded spectral product, simulated on one device: each of 4 'ranks' transforms its sensor rows and fills
    its send buffer; the all-to-all is done by hand (recv_r[src] = send_src[r]); the assembled AK of every rank must equal the
    dense route's AK for that rank's columns.  Also checks the slab-restricted forward operator against the full one."""
    import geobo_amd.engine as E
    f = load_golden("oracle32_matern32.npz")
    s = settings_for(32, 32, 32, kernelfunc="matern32")
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(f["gp_length_in"].copy())]
    sel_t = torch.as_tensor(f["sel"], device="cuda")
    world, props = 4, (0, 1, 2)
    full = E.PosteriorEngine(s, method="dense")
    Af = {k: full.operator(k, f["sensor_locations"]).clone() for k in ("grav", "magn")}
    engs = [E.PosteriorEngine(s, rank=r, world=world) for r in range(world)]
    assert all(e.exchange for e in engs)
    sends = []
    for e in engs:
        A_g, A_m = e.operator("grav", f["sensor_locations"]), e.operator("magn", f["sensor_locations"])
        # this rank's slab of every sensor row, and all voxels of its own sensor rows, are bit-identical to the full operator
        rows_r = e.Ms // world
        for k, A in (("grav", A_g), ("magn", A_m)):
            assert A.shape == (e.Ms_pad, e.nc) and torch.equal(A[:e.Ms], Af[k][:e.Ms, e.c0:e.c1])     # compact slab buffer
            assert torch.equal(e._Arows[k][:, :e.N], Af[k][e.rank * rows_r:(e.rank + 1) * rows_r, :e.N])
        from geobo_amd.spectral import SpectralProduct
        e._spectral = SpectralProduct(e.nx, e.ny, e.nz, e.device)
        sends.append([e._exchange_send(s_, func, lengths, W, "matern32", 1.0, props).clone() for s_, func in ((0, "grav"), (1, "magn"))])
    for r, e in enumerate(engs):
        M_pad = E.hip.pad_m(2 * e.Ms_pad + f["sel"].size)
        AK = torch.zeros((M_pad, len(props) * e.nc), dtype=torch.float64, device="cuda")
        for s_ in (0, 1):               # what the all-to-all of operator s_ delivers to rank r: block r of every source's buffer
            e._exchange_place(AK, torch.stack([sends[src][s_][r] for src in range(world)]), props, s_)
        d = E.PosteriorEngine(s, rank=r, world=world, method="dense")
        ref, _ = d._assemble_AK(Af["grav"], Af["magn"], sel_t, lengths, W, "matern32", 1.0, props)
        rows = np.r_[0:e.Ms, e.Ms_pad:e.Ms_pad + e.Ms]
        diff = (AK[rows] - ref[rows]).abs().max().item()
        assert diff <= 1e-12 * ref.abs().max().item(), (r, diff)


def test_row_sharded_lattice_gram_matches_the_single_rank_AkA():
    """Row-sharded lattice Gram: every rank correlates its OWN sensor rows of A K (all voxels; the three blocks AkA's lower triangle
    needs) with the stencil tables and AkA arrives as row blocks by an all-gather, the (magn, grav) block transposed from
    (grav, magn).  Four ranks simulated on one device; the assembled matrix against the single-rank block-column AkA (lower triangle
    incl. drill rows)."""
    import geobo_amd.engine as E
    from geobo_amd.spectral import SpectralProduct
    nx, ny, nz = 64, 48, 64
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s, props=(0, 1))
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    del inv
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(np.array([200.0, 202.0, 204.0]))]
    sel_t = torch.as_tensor(np.array([5, 777, 12345, 100000]), device="cuda")
    props = (0, 1)
    ref_eng = E.PosteriorEngine(s)
    A_g, A_m = ref_eng.operator("grav", loc), ref_eng.operator("magn", loc)
    AK, M_pad = ref_eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, props)
    ref = torch.tril(ref_eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, props)).clone()
    del ref_eng, A_g, A_m, AK
    torch.cuda.empty_cache()
    world = 4
    blocks, drill = [], None
    for r in range(world):
        e = E.PosteriorEngine(s, rank=r, world=world)
        assert e.route.family == "rows"
        Ag_r, Am_r = e.operator("grav", loc), e.operator("magn", loc)
        e._spectral_product()
        assert e._rows_ok(Ag_r, Am_r)
        # this rank's rows of A K -- (grav, 0), (grav, 1), (magn, 1) -- a chunk at a time, straight through the lattice Gram
        lo, dr = e._rows_aka_local(props, sel_t, lengths, W, "matern32", 1.0)
        blocks.append(lo.clone())
        drill = dr.clone()
        del e, lo, dr
        gc.collect()
        torch.cuda.empty_cache()
    eng = E.PosteriorEngine(s, rank=0, world=world)
    rows_r, Msp, off_d, Md = eng.Ms // world, eng.Ms_pad, 2 * eng.Ms_pad, sel_t.numel()
    AkA = torch.zeros((M_pad, M_pad), dtype=torch.float64, device="cuda")
    for src in range(world):                                       # what _assemble_AkA_rows does with the gathered blocks
        r0 = src * rows_r
        AkA[r0:r0 + rows_r, :off_d] = blocks[src][:, :off_d]
        AkA[Msp + r0:Msp + r0 + rows_r, Msp:off_d] = blocks[src][:, off_d:]
    AkA[Msp:off_d, :Msp] = AkA[:Msp, Msp:off_d].t()
    AkA[off_d:off_d + Md, :off_d] = drill[:Md]
    got = torch.tril(eng._finish_AkA(AkA, M_pad, sel_t, lengths, "matern32", 1.0, s.gp_err))
    d = (got - ref).abs().max().item()
    assert d <= 1e-12 * ref.abs().max().item(), d


@pytest.mark.parametrize("dims", [(48, 32, 64), (64, 48, 64), (32, 16, 64), (16, 80, 16), (64, 64, 64), (16, 128, 16), (64, 128, 64), (16, 96, 32),
                                  (32, 112, 16), (16, 144, 16), (32, 32, 32), (32, 16, 32)])
@pytest.mark.parametrize("kern,cross", [("matern32", True), ("sparse", False)])
def test_spectral_product_matches_lattice_contraction_on_non_cubic_grids(dims, kern, cross):
    """Every kernel combination of the spectral route against the dense lattice-table contraction (geobo_ak_fused_grid) on
    random operator rows: fused (x,z) transform for nx = 48 / 64 with nz = 64, batched-GEMM passes otherwise; Toeplitz y
    stage for ny <= 64 and for ny = 80 / 96 / 112 / 128 (windowed kernel), y through the spectrum for ny = 144; one to three property
    blocks per sweep."""
    from geobo_amd import hip
    from geobo_amd.spectral import SpectralProduct
    nx, ny, nz = dims
    N = nx * ny * nz
    rows = 256
    g = torch.Generator().manual_seed(nx * 7 + ny)
    A = torch.zeros((rows, N + 16), dtype=torch.float64, device="cuda")[:, :N]
    A[:37] = (torch.rand((37, N), generator=g, dtype=torch.float64) * 2 - 1).cuda()
    sp = SpectralProduct(nx, ny, nz, "cuda")
    assert sp.fused_xz == ((nx, nz) in ((48, 64), (64, 64))) and sp.dense_y == (ny in (16, 32, 48, 64, 80, 96, 112, 128))
    kid = hip.kernel_id(kern, cross)
    tabs = [hip.cov_table(kid, nx, ny, nz, 100.0, 90.0, 110.0, l1, l2, w, 1.3, "cuda")
            for l1, l2, w in ((210.0, 170.0, 0.7), (260.0, 240.0, 1.0), (150.0, 300.0, 0.4))]
    for nblk in (1, 2, 3):
        outs = [torch.full((rows, N), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nblk)]
        sp.product(A, 37, [sp.eigenvalues(t) for t in tabs[:nblk]], outs)
        for j in range(nblk):
            ref = torch.empty((rows, N), dtype=torch.float64, device="cuda")
            hip.ak_fused_grid(A, nx, ny, nz, tabs[j], 0, N, ref)
            d = (outs[j][:37] - ref[:37]).abs().max().item()
            assert d <= 2e-13 * ref[:37].abs().max().item(), (nblk, j, d)
            assert torch.isnan(outs[j][37:]).all()


def test_graft_entry_build_then_smoke_in_one_process():
    """build() loads the C-ABI library before anything touched the GPU; smoke() must still run in the same process (the
    library has to come up on the HIP runtime bundled with torch -- geobo_amd/_lib.py imports torch first)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('OK')"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("ny", [48, 64])
def test_lattice_gram_matches_the_gemm(ny, monkeypatch):
    """AkA by the (y, x) correlation of the lattice survey (lattice_gram.py) against the N-deep GEMM, all blocks incl. drill rows."""
    import geobo_amd.engine as E
    nx, nz = 64, 64
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s, props=(0, 1))
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(np.array([200.0, 202.0, 204.0]))]
    sel_t = torch.as_tensor(np.array([5, 777, 12345, 100000]), device="cuda")
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GEOBO_AKA_LATTICE", flag)
        eng = E.PosteriorEngine(s)
        A_g, A_m = eng.operator("grav", loc), eng.operator("magn", loc)
        assert (eng._lam.get("grav") is not None) == (flag == "1")
        AK, M_pad = eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, (0, 1))
        AkA = eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, (0, 1))
        out[flag] = torch.tril(AkA).clone()
        del eng, A_g, A_m, AK, AkA
        torch.cuda.empty_cache()
    d = (out["0"] - out["1"]).abs().max().item()
    assert d <= 1e-12 * out["0"].abs().max().item(), d


def test_lattice_gram_needs_an_even_stencil():
    """An inclined magnetic field makes the operator's stencil table odd in x / y: the lattice Gram must step aside (GEMM)."""
    import geobo_amd.engine as E
    nx, ny, nz = 64, 48, 64
    s = settings_for(nx, ny, nz, kernelfunc="exp")
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s, props=(0, 1))
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    eng = E.PosteriorEngine(s)
    eng.operator("grav", loc)
    eng.operator("magn", loc, B=(0.4, -0.3, 0.85))
    assert eng._lam["grav"] is not None and eng._lam["magn"] is None
    eng.operator("magn", loc, B=(0.0, 0.0, 1.0))
    assert eng._lam["magn"] is not None
    # a survey off the lattice: no plan at all
    loc2 = loc.copy(); loc2[:, 0] += 3.0
    eng2 = E.PosteriorEngine(s)
    eng2.operator("grav", loc2)
    assert eng2._lam.get("grav") is None
    # operators="auto": no operator is materialised where the stencil is even; the inclined field falls back to a resident one
    eng3 = E.PosteriorEngine(s, operators="auto")
    assert isinstance(eng3.operator("grav", loc), E.StreamedOperator)
    Am = eng3.operator("magn", loc, B=(0.4, -0.3, 0.85))
    assert isinstance(Am, torch.Tensor) and eng3._lam["magn"] is None
    assert torch.equal(Am, eng.operator("magn", loc, B=(0.4, -0.3, 0.85)))
    assert isinstance(eng3.operator("magn", loc, B=(0.0, 0.0, 1.0)), E.StreamedOperator)
    assert isinstance(eng3.operator("grav", loc, full=True), torch.Tensor)          # an explicit request for the matrix


@pytest.mark.parametrize("world", [2, 4])
def test_lattice_gram_y_slab_shards_add_up(world, monkeypatch):
    """Column-sharded ranks correlate their own y-slab of the A K rows; the partial block columns must add up to the same AkA as
    the column-sharded N-deep GEMM (the all-reduce is replaced by an explicit sum over the simulated ranks)."""
    import geobo_amd.engine as E
    nx, ny, nz = 64, 64, 64
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s, props=(0, 1))
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    del inv
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    W = E.weight_matrix(s.gp_coeff)
    lengths = [float(v) for v in E.create_cov_lengths(np.array([200.0, 202.0, 204.0]))]
    sel_t = torch.as_tensor(np.array([5, 777, 12345, 100000]), device="cuda")
    monkeypatch.setattr(E, "allreduce_sum_", lambda t, world, group=None: t)
    monkeypatch.setenv("GEOBO_SPECTRAL_EXCHANGE", "0")
    total = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GEOBO_AKA_LATTICE", flag)
        acc = None
        for r in range(world):
            eng = E.PosteriorEngine(s, rank=r, world=world)
            A_g, A_m = eng.operator("grav", loc), eng.operator("magn", loc)
            assert (eng._lam.get("grav") is not None) == (flag == "1")
            AK, M_pad = eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, (0, 1))
            AkA = torch.tril(eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, (0, 1)))
            acc = AkA.clone() if acc is None else acc + AkA
            del eng, A_g, A_m, AK, AkA
            torch.cuda.empty_cache()
        total[flag] = acc
    d = (total["0"] - total["1"]).abs().max().item()
    assert d <= 1e-12 * total["0"].abs().max().item(), d


def test_full_size_inversion_is_bitwise_reproducible():
    """Two 64^3 inversions of the same survey (operators rebuilt in between) give bit-identical cubes: no atomics, fixed
    summation orders, and the hand-synchronised LDS pipelines never read a tile before it has landed."""
    import bench
    from geobo_amd.config_loader import Settings
    from geobo_amd.inversion import Inversion
    n = 64
    s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc="matern32"))
    inv = Inversion(settings=s, props=(0, 1))
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
    runs = []
    for _ in range(3):
        inv.engine.clear_operators()
        inv.gp_length = np.array([200.0, 202.0, 204.0])
        cubes = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
        runs.append([np.array(c, copy=True) for c in (cubes[0], cubes[1], cubes[3], cubes[4])])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a, b)


def test_streamed_operator_rows_are_windows_of_the_stencil_table():
    """Lattice survey, streamed operators: the forward transform reads the operator rows as windows of the stencil table Q plus
    the two boundary slabs (geobo_xz2d_fold_lattice) -- bit-identical to transforming the materialised rows, for batches that
    start anywhere, and the inversion gives the resident-operator cubes."""
    import geobo_amd.engine as E
    from geobo_amd.inversion import Inversion
    nx, ny, nz = 64, 16, 64
    s = settings_for(nx, ny, nz, kernelfunc="matern32")
    inv = Inversion(settings=s, props=(0, 1), operators="streamed")
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    eng = inv.engine
    for func, B in (("grav", s.magneticField * 0.0), ("magn", s.magneticField)):
        A = eng.operator(func, loc, B=B)
        assert isinstance(A, E.StreamedOperator) and A.lattice is not None
        sp = eng._spectral
        buf = eng._op_rows_buffer()
        for r0, R in ((0, 7), (300, 64), (eng.Ms - 5, 5)):
            rows = A.rows_into(buf, r0, R)
            planes = lambda t: t[:R * ny * sp.Cp].view(R * ny, sp.Cp)[:, :4 * nx * nz]     # (the planes sit sp.Cp doubles apart: padded)
            ref = planes(sp.forward_zx(rows, R, sp.G, src_row_stride=rows.stride(0), out_name="feed_ref")).clone()
            got = planes(sp.forward_zx(A.lattice.rows(r0), R, sp.G, out_name="feed_got"))
            assert torch.equal(got, ref), (func, r0)
    rng = np.random.default_rng(3)
    grav, mag = rng.standard_normal(nx * ny), rng.standard_normal(nx * ny)
    d0 = np.zeros((ny, nx, nz)); d0[3, 5, 7] = 1.0; d0[10, 40, 20] = 2.0
    inv.gp_length = np.array([200.0, 202.0, 204.0])
    got = inv.cubing(grav, mag, d0[d0 != 0], loc, d0)
    ref_inv = Inversion(settings=s, props=(0, 1))
    ref_inv.gp_length = np.array([200.0, 202.0, 204.0])
    ref = ref_inv.cubing(grav, mag, d0[d0 != 0], loc, d0)
    for g, r in zip(got, ref):      # (ny = 16 has no lattice Gram: AkA panels accumulate in another order than the resident split-K)
        if np.isfinite(r).any():
            assert normwise(g, r) < 1e-10


def test_auto_operators_with_an_inclined_field_match_resident_operators():
    """operators="auto" with an inclined magnetic field: the gravity operator is never materialised (even stencil), the magnetic
    one is resident (odd stencil: AkA by the GEMM); the cubes equal the all-resident run's."""
    from geobo_amd.inversion import Inversion
    nx, ny, nz = 64, 48, 64
    s = settings_for(nx, ny, nz, kernelfunc="matern32", XMAG=0.4, YMAG=-0.3, ZMAG=0.85)
    out = {}
    rng = np.random.default_rng(11)
    grav, mag = rng.standard_normal(nx * ny), rng.standard_normal(nx * ny)
    d0 = np.zeros((ny, nx, nz)); d0[3, 5, 7] = 1.0; d0[40, 40, 20] = 2.0
    for mode in ("auto", "resident"):
        inv = Inversion(settings=s, props=(0, 1), operators=mode)
        inv.create_cubegeometry()
        xe, ye, ze = inv.engine.node_axes()
        X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
        loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
        inv.gp_length = np.array([200.0, 202.0, 204.0])
        out[mode] = inv.cubing(grav, mag, d0[d0 != 0], loc, d0)
        kinds = sorted(type(v).__name__ for v in inv.engine._A.values())
        assert kinds == (["StreamedOperator", "Tensor"] if mode == "auto" else ["Tensor", "Tensor"]), kinds
        del inv
        torch.cuda.empty_cache()
    for a, b in zip(out["auto"], out["resident"]):
        assert np.isnan(b).all() if np.isnan(a).all() else normwise(a, b) <= 1e-11
