import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def settings_for(nx, ny, nz, vox=100.0, **kw):
    """Settings of the synthetic survey used across the tests."""
    from geobo_amd.config_loader import Settings
    base = dict(xmin=0, xmax=vox * nx, ymin=0, ymax=vox * ny, zmax=0, zoff=1, zLcube=vox * nz,
                xNcube=nx, yNcube=ny, zNcube=nz, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2],
                kernelfunc="exp", XMAG=0, YMAG=0, ZMAG=1)
    base.update(kw)
    return Settings(base)


def oracle_grid(s):
    from oracle import geobo_oracle as O
    return O.Grid(nx=s.xNcube, ny=s.yNcube, nz=s.zNcube, xmin=s.xmin, xmax=s.xmax, ymin=s.ymin, ymax=s.ymax,
                  zmax=s.zmax, zLcube=s.zLcube, zoff=s.zoff, gp_lengthscale=s.gp_lengthscale, gp_err=tuple(s.gp_err),
                  gp_coeff=tuple(s.gp_coeff), kernelfunc=s.kernelfunc, mag=(s.XMAG, s.YMAG, s.ZMAG), c_G=s.c_G,
                  c_SI_TO_MILLIGALS=s.c_SI_TO_MILLIGALS, c_GCM3_TO_SI=s.c_GCM3_TO_SI, fcor_grav=s.fcor_grav,
                  fcor_mag=s.fcor_mag)


def normwise(a, b):
    """max|a-b| / max|b| -- the per-cube parity measure (elementwise relative error is meaningless for
    near-zero posterior means)."""
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / np.abs(b).max())
