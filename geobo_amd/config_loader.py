"""Settings of the inversion (mirror of the reference's geobo/config_loader.py:20-59).

The reference reads `sys.argv[1]` at import time and injects every YAML key into module globals that
the other modules star-import.  Here the same keys and the same derived quantities live in an explicit
`Settings` object; `load()` additionally publishes them as module globals of this module so that code
written against the reference (`from geobo.config_loader import *` style) keeps working.

YAML keys honoured (examples/settings_example1.yaml in the reference): inpath, outpath, FNAME_*,
drill_features, ifeature, xmin..ymax, zmax, zoff, zLcube, xNcube, yNcube, zNcube, gen_simulation,
modelname, gp_lengthscale, gp_err, gp_coeff, kernelfunc, optimize_gp, XMAG, YMAG, ZMAG, plotting and
bayesopt_* switches, kappa, beta, c_G, c_SI_TO_MILLIGALS, c_GCM3_TO_SI, fcor_grav, fcor_mag.
"""
import os

import numpy as np

DEFAULTS = dict(
    inpath="./", outpath="./results/", FNAME_drilldata=None, FNAME_gravsurvey=None, FNAME_magsurvey=None,
    drill_features=["DENSITY", "MAGSUS"], ifeature=0,
    xmin=0, xmax=1000, ymin=0, ymax=1000, zmax=0, zoff=1, zLcube=1000.0, xNcube=10, yNcube=10, zNcube=10,
    gen_simulation=False, modelname="cylinders",
    gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="sparse", optimize_gp=False,
    XMAG=0, YMAG=0, ZMAG=1,
    plot_vertical=False, plot3d=False, regrid_sparse=False, font_scale=1.5,
    bayesopt_vertical=False, bayesopt_nonvertical=False, kappa=1, beta=0.0,
    c_G=6.673848e-11, c_SI_TO_MILLIGALS=10000, c_GCM3_TO_SI=1000.0, fcor_grav=1.0, fcor_mag=0.001,
)


class Settings:
    """YAML keys as attributes + the derived constants of config_loader.py:41-59."""

    def __init__(self, cfg=None, **overrides):
        d = dict(DEFAULTS)
        d.update(cfg or {})
        d.update(overrides)
        self._keys = list(d.keys())
        for k, v in d.items():
            setattr(self, k, v)
        self.derive()

    def derive(self):
        # config_loader.py:41-59, same expressions / same rounding
        self.xLcube = self.xmax - self.xmin
        self.yLcube = self.ymax - self.ymin
        self.zmin = self.zmax - self.zLcube
        self.magneticField = np.asarray([self.XMAG, self.YMAG, self.ZMAG]) * 1e-3
        self.c_MILLIGALS_UNITS = self.c_G * self.c_SI_TO_MILLIGALS * self.c_GCM3_TO_SI
        self.xvoxsize = self.xLcube / self.xNcube * 1.
        self.yvoxsize = self.yLcube / self.yNcube * 1.
        self.zvoxsize = self.zLcube / self.zNcube * 1.
        self.Nsensor = self.xNcube * self.yNcube
        return self

    @classmethod
    def from_yaml(cls, fname, **overrides):
        import yaml
        with open(fname) as f:
            return cls(yaml.safe_load(f), **overrides)

    def as_dict(self):
        return {k: getattr(self, k) for k in self._keys}

    def public_names(self):
        return self._keys + ["xLcube", "yLcube", "zmin", "magneticField", "c_MILLIGALS_UNITS", "xvoxsize", "yvoxsize",
                             "zvoxsize", "Nsensor"]

    def __repr__(self):
        return "Settings(%dx%dx%d, kernelfunc=%r)" % (self.xNcube, self.yNcube, self.zNcube, self.kernelfunc)


settings = None  # the active settings (what the reference keeps as module globals)


def load(source=None, create_outpath=False, **overrides):
    """Activate settings from a YAML path, a dict or a Settings object; returns the Settings.

    Like the reference, the keys are also published as globals of this module."""
    global settings
    if isinstance(source, Settings):
        s = source
        for k, v in overrides.items():
            setattr(s, k, v)
        s.derive()
    elif isinstance(source, str):
        s = Settings.from_yaml(source, **overrides)
    else:
        s = Settings(source, **overrides)
    if create_outpath and s.outpath:
        os.makedirs(s.outpath, exist_ok=True)  # config_loader.py:39
    settings = s
    g = globals()
    for k in s.public_names():
        g[k] = getattr(s, k)
    return s


def active():
    if settings is None:
        raise RuntimeError("no settings loaded: call geobo_amd.config_loader.load(<yaml path | dict | Settings>) first")
    return settings
