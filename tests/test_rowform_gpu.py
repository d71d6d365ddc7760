"""The row form of the structured step (geobo_amd/rowform.py, plan.Route.family "rows") on ONE rank and on grids WITHOUT fused
kernels: the batched-GEMM stand-ins of every n = 64 kernel (lattice Gram x step + geobo_lamdot_z, boundary-slab spectra for any
extent, lattice convolution L^-1 A through two GEMM passes, stored-batch sum of squares geobo_sumsq_accum) and the chunking of the
rank's rows, against the reference's golden cubes (16^3), the pinned oracle's vectors (32^3) and the column form (non-cubic grids).
GPU only; every call goes through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden, normwise, settings_for
from test_inversion_gpu import TOL_T3, _check_cubes, _inv

pytestmark = pytest.mark.gpu


def test_streaming_reductions_match_torch():
    from geobo_amd import hip
    g = torch.Generator().manual_seed(5)
    for rows, n, slots in ((37, 4096, 8), (5, 130, 3), (256, 32768, 8)):
        a = torch.rand((rows + 3, n + 6), generator=g, dtype=torch.float64).cuda()[:, :n]
        b = torch.rand((rows, n), generator=g, dtype=torch.float64).cuda()
        for second in (None, b):
            ss = torch.rand((slots, n), generator=g, dtype=torch.float64).cuda()
            want = ss.sum(0) + ((a[:rows] + (second if second is not None else 0.0)) ** 2).sum(0)
            hip.sumsq_accum(a, second, rows, ss)
            assert (ss.sum(0) - want).abs().max().item() <= 1e-12 * want.abs().max().item()
    for batch, planes, px, nz in ((12, 4, 64, 32), (7, 7, 96, 48), (3, 2, 256, 128)):
        D = torch.rand((batch, px, nz), generator=g, dtype=torch.float64).cuda()
        lam = torch.rand((planes, px, nz), generator=g, dtype=torch.float64).cuda()
        out = torch.empty((batch, px), dtype=torch.float64, device="cuda")
        hip.lamdot_z(batch, planes, px, nz, D, lam, out)
        want = (D * lam[torch.arange(batch) % planes]).sum(2)
        assert (out - want).abs().max().item() <= 1e-13 * want.abs().max().item()


@pytest.mark.parametrize("name,kern", [("cube16_exp", "exp"), ("cube16_matern32", "matern32"), ("cube16_sparse", "sparse")])
def test_row_form_on_one_rank_against_the_reference(name, kern, monkeypatch):
    """16^3 reference cubes through the row form forced onto a grid it would not pick (GEOBO_ROWS=1): every stage runs on its
    shape-independent kernels, 64-row chunks."""
    monkeypatch.setenv("GEOBO_ROWS", "1")
    monkeypatch.setenv("GEOBO_ROW_CHUNK", "64")
    f = load_golden(name + ".npz")
    s = settings_for(16, 16, 16, kernelfunc=kern)
    inv = _inv(s)
    assert inv.engine.route.family == "rows"
    inv.gp_length = f["gp_length_in"].copy()
    d0 = f["drilldata0"]
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert inv.engine.step_route == "rows"
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " row form T3")
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))


@pytest.mark.parametrize("name,kern", [("oracle32_exp", "exp"), ("oracle32_matern32", "matern32")])
@pytest.mark.parametrize("operators", ["resident", "streamed"])
def test_row_form_32_against_oracle_vectors(name, kern, operators, monkeypatch):
    monkeypatch.setenv("GEOBO_ROWS", "1")
    monkeypatch.setenv("GEOBO_ROW_CHUNK", "256")
    f = load_golden(name + ".npz")
    s = settings_for(32, 32, 32, kernelfunc=kern)
    inv = _inv(s, operators=operators)
    inv.gp_length = f["gp_length_in"].copy()
    d0 = np.zeros(32 ** 3)
    d0[f["sel"]] = f["drillvalues"]
    d0 = d0.reshape(32, 32, 32)
    cubes = inv.cubing(f["gravfield"], f["magfield"], d0[d0 != 0], f["sensor_locations"], d0)
    assert inv.engine.step_route == "rows"
    _check_cubes(cubes, f["cubes"], TOL_T3, name + " row form T3")
    assert abs(inv.logl - float(f["logl"])) <= 1e-8 * abs(float(f["logl"]))


@pytest.mark.parametrize("dims,kern,props,md", [((48, 32, 64), "matern32", (0, 1, 2), 20), ((32, 48, 16), "exp", (0, 1), 0),
                                               ((16, 80, 32), "sparse", (0, 1, 2), 7), ((64, 48, 64), "matern32", (0, 1), 50),
                                               ((64, 80, 64), "matern32", (0, 1), 20)])
def test_row_form_matches_the_column_form(dims, kern, props, md, monkeypatch):
    """Non-cubic grids (fused (x, z) kernels with batched-GEMM Gram; the windowed y stage at ny = 80, with the batched-GEMM transforms
    and -- 64 x 80 x 64 -- with the radix-2 ones, where the two-term rows add in a padded-stride spectrum; the all-fused 64 x 48 x 64
    shape whose default is the one-rank materialised form): the row form against whatever the planner picks without it."""
    import bench
    from geobo_amd.inversion import Inversion
    nx, ny, nz = dims
    s = settings_for(nx, ny, nz, kernelfunc=kern)
    lengths = np.array([200.0, 202.0, 204.0])
    monkeypatch.setenv("GEOBO_ROWS", "0")          # (the planner itself picks the row form for the larger of these grids)
    ref = Inversion(settings=s, props=props)
    grav, mag, loc, drill0 = bench.synthetic_inputs(ref, md)
    ref.gp_length = lengths.copy()
    want = ref.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    assert ref.engine.step_route != "rows"
    monkeypatch.setenv("GEOBO_ROWS", "1")
    monkeypatch.setenv("GEOBO_ROW_CHUNK", "512")
    inv = Inversion(settings=s, props=props)
    inv.gp_length = lengths.copy()
    got = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    assert inv.engine.step_route == "rows"
    for c, r in zip(got, want):
        if np.isnan(r).all():
            assert np.isnan(c).all()
        else:
            assert normwise(c, r) <= 1e-10, (dims, normwise(c, r))
    assert abs(inv.logl - ref.logl) <= 1e-10 * abs(ref.logl)


def test_off_lattice_survey_demotes_the_row_form(monkeypatch):
    """A survey that is not the cube's own lattice cannot use the row form: the engine says why, runs the column form and returns
    to the row form when the next survey allows it."""
    import bench
    from geobo_amd.inversion import Inversion
    monkeypatch.setenv("GEOBO_ROWS", "1")
    s = settings_for(16, 16, 16, kernelfunc="exp")
    inv = Inversion(settings=s, props=(0, 1))
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 5)
    inv.gp_length = np.array([200.0, 202.0, 204.0])
    on = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    assert inv.engine.step_route == "rows"
    loc2 = loc.copy()
    loc2[:, 0] += 3.0 * np.sin(np.arange(loc.shape[0]))
    inv.cubing(grav, mag, drill0[drill0 != 0], loc2, drill0)
    assert inv.engine.step_route == "columns" and inv.engine._rows_denied
    again = inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
    assert inv.engine.step_route == "rows" and not inv.engine._rows_denied
    for a, b in zip(on, again):
        assert np.array_equal(a, b, equal_nan=True)
