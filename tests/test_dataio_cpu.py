"""f3/f4 rows: ingestion, export and acquisition against what the reference produced from the same files (CPU)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

DATA = os.path.join(GOLDEN, "data")


def _settings(name):
    from geobo_amd.config_loader import Settings
    f = load_golden(name + ".npz")
    d = json.loads(str(f["settings_json"]))
    d["inpath"] = os.path.join(DATA, "synthetic" if name == "example1" else "sample") + "/"
    return Settings(d), f


@pytest.mark.parametrize("name", ["example1", "example2"])
def test_ingestion_reproduces_the_reference_cubing_inputs(name):
    """read_surveydata / read_drilldata / align_drill -> exactly the arrays run_geobo.py handed to Inversion.cubing."""
    from geobo_amd import dataio
    from geobo_amd.inversion import Inversion
    s, f = _settings(name)
    inv = Inversion(settings=s)
    voxelpos = inv.create_cubegeometry()
    grav, mag, loc = dataio.read_surveydata(s)
    assert grav.dtype == f["gravfield"].dtype and mag.dtype == f["magfield"].dtype     # float32 survives (z-scoring dtype)
    assert np.array_equal(grav, f["gravfield"]) and np.array_equal(mag, f["magfield"])
    assert np.array_equal(loc, f["sensor_locations"])
    drilldata, coord, _ = dataio.read_drilldata(s, s.drill_features, voxelpos)
    d0 = drilldata[s.ifeature]
    assert d0.shape == f["drilldata0"].shape == (s.xNcube, s.yNcube, s.zNcube)
    assert np.array_equal(d0 != 0, f["drilldata0"] != 0)
    assert np.abs(d0 - f["drilldata0"]).max() <= 1e-13 * np.abs(f["drilldata0"]).max()
    assert np.allclose(d0[d0 != 0], f["drillfield"], rtol=1e-13, atol=0)


def test_vtk_writer_is_byte_identical_to_the_reference_file(tmp_path):
    from geobo_amd import dataio
    ref = os.path.join(DATA, "results_cylinders", "cube_density.vtk")
    cube, origin, spacing = dataio.read_vtkcube(ref)
    assert cube.shape == (16, 25, 16)
    f = load_golden("example1.npz")
    assert np.array_equal(cube, f["vtk_cubes"][0])
    out = tmp_path / "cube.vtk"
    dataio.create_vtkcube(cube, origin, spacing, str(out))
    assert out.read_bytes() == open(ref, "rb").read()


@pytest.mark.parametrize("name,res", [("example1", "results_cylinders"), ("example2", "results_sample")])
def test_vertical_acquisition_reproduces_committed_gains(name, res):
    """The committed newdrill_proposals_vertical.csv lists (NORTHING, EASTING, BO_GAIN): the utility evaluated on the
    committed posterior cubes at those voxels must give those gains (4 decimals)."""
    import pandas as pd
    from geobo_amd.acquisition import Acquisition
    s, f = _settings(name)
    acq = Acquisition(s, f["vtk_cubes"][2], f["vtk_cubes"][5])
    df = pd.read_csv(os.path.join(DATA, res, "newdrill_proposals_vertical.csv"))
    for _, r in df.iterrows():
        i0 = (r.NORTHING - s.ymin - 0.5 * s.yvoxsize) / s.yvoxsize
        i1 = (r.EASTING - s.xmin - 0.5 * s.xvoxsize) / s.xvoxsize
        assert abs(-acq.futility_vertical([i0, i1]) - r.BO_GAIN) <= 6e-5
    assert acq.futility_vertical([0, 5]) == np.inf and acq.futility_vertical([np.nan, 5]) == np.inf


def test_nonvertical_acquisition_reproduces_committed_gains():
    import pandas as pd
    from geobo_amd.acquisition import Acquisition
    s, f = _settings("example1")
    acq = Acquisition(s, f["vtk_cubes"][2], f["vtk_cubes"][5])
    df = pd.read_csv(os.path.join(DATA, "results_cylinders", "newdrill_proposals_non-vertical.csv"))
    ok = 0
    for _, r in df.head(10).iterrows():
        g = -acq.futility_drill([r.NORTHING - s.ymin, r.EASTING - s.xmin, r.AZIMUTH, r.DIP])
        ok += abs(g - r.BO_GAIN) <= 0.02 * abs(r.BO_GAIN)      # proposals are rounded to 0.1 m / 0.01 deg in the csv
    assert ok >= 8
    assert acq.futility_drill([1e9, 1e9, 0., 45.]) == 0.0     # out-of-cube path -> the reference's except branch
