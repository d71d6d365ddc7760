"""Joint GP inversion -- the reference's `geobo/inversion.py` `Inversion` class on the MI355X path.

Same constructor-less usage, attributes and method signatures as the reference:

    inv = Inversion()                       # settings: geobo_amd.config_loader.load(...) (or pass settings=)
    voxelpos = inv.create_cubegeometry()
    density_rec, magsus_rec, drill_rec, density_var, magsus_var, drill_var = \
        inv.cubing(gravfield, magfield, drillfield, sensor_locations, drilldata0)

`method` selects how A K is formed: "dense" = fused fp64-MFMA contraction with the covariance tile generated in the
kernel (any grid), "spectral" = real-DFT route on batched MFMA GEMMs (regular grids with extents % 16 == 0, ~40x
faster at 64^3, same results to ~1e-13), "auto" (default) = spectral when applicable.

`cubing`/`predict3`/`calc_logl` run matrix-free on the GPU (engine.PosteriorEngine): D2, the 3N x 3N prior
and the 3N x 3N posterior covariance of the reference (inversion.py:92,117) are never formed -- only the
posterior diagonal that `cubing` consumes (inversion.py:238).  No CPU fallback exists.
"""
import sys

import numpy as np

from . import config_loader
from . import kernels as kernel
from . import sensormodel as sm
from .engine import CholeskyError, PosteriorEngine, create_cov_lengths


class DiagonalCovariance:
    """What `predict3` returns in place of the (3N,3N) posterior covariance: only its diagonal exists.
    `np.diag(obj)` / `obj.diagonal()` work; anything needing off-diagonal entries raises."""

    def __init__(self, diag):
        self._d = np.asarray(diag)
        self.shape = (self._d.size, self._d.size)

    def diagonal(self):
        return self._d

    def __array__(self, dtype=None, copy=None):
        raise TypeError("the MI355X path keeps only the diagonal of the posterior covariance; use .diagonal()")


class Inversion:
    """Class for inversion and reconstruction of 3D cubes from 2D sensor data (inversion.py:23-248)."""

    def __init__(self, settings=None, props=(0, 1, 2), rank=0, world=1, group=None, device=None, profile=False,
                 method="auto"):
        self.settings = s = settings or config_loader.active()
        # inversion.py:46-51 -- NB x voxel size for all three length scales
        self.gp_length = s.gp_lengthscale * np.asarray([s.xvoxsize, s.xvoxsize, s.xvoxsize])
        self.gp_sigma = np.asarray(s.gp_err)
        self.coeffm = np.asarray(s.gp_coeff)
        self.gp_amp = 1.
        self.props = tuple(props)
        self._engine_args = dict(rank=rank, world=world, group=group, device=device, profile=profile, method=method)
        self._engine = None

    # ---- geometry (host, inversion.py:54-74) -------------------------------------------------------------------
    def create_cubegeometry(self):
        s = self.settings
        xedge = np.linspace(0, s.xNcube, s.xNcube + 1) * s.xvoxsize
        yedge = np.linspace(0, s.yNcube, s.yNcube + 1) * s.yvoxsize
        zedge = np.linspace(0, -s.zNcube, s.zNcube + 1) * s.zvoxsize + s.zmax
        xEdges, yEdges, zEdges = np.meshgrid(xedge, yedge, zedge)
        self.Edges = np.asarray([xEdges, yEdges, -zEdges])
        xnew = np.arange(s.xvoxsize / 2., s.xLcube + s.xvoxsize / 2., s.xvoxsize)
        ynew = np.arange(s.yvoxsize / 2., s.yLcube + s.yvoxsize / 2., s.yvoxsize)
        znew = s.zmax - np.arange(s.zvoxsize / 2., s.zLcube + s.zvoxsize / 2., s.zvoxsize)
        self.xxx, self.yyy, self.zzz = np.meshgrid(xnew, ynew, znew)
        self.voxelpos = np.vstack([self.xxx.flatten(), self.yyy.flatten(), self.zzz.flatten()])
        return self.voxelpos

    # ---- engine --------------------------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = PosteriorEngine(self.settings, **self._engine_args)
        return self._engine

    def _operators(self):
        eng = self.engine
        axes = None
        if hasattr(self, "Edges"):
            axes = sm._edge_axes(self.Edges, eng.nx, eng.ny, eng.nz)
        A_g = eng.operator("grav", self.sensor_locations, B=self.settings.magneticField * 0., axes=axes)
        A_m = eng.operator("magn", self.sensor_locations, B=self.settings.magneticField, axes=axes)
        return A_g, A_m

    def _drill_selection(self):
        # inversion.py:219 + sensormodel.A_drill: rows = voxels with non-zero drill data, ascending flat index
        return np.flatnonzero(np.asarray(self.drilldata0).reshape(-1) != 0)

    def _run(self, gp_amp, gp_length, coeffm, calclogl, want_mean_var):
        A_g, A_m = self._operators()
        lengths = create_cov_lengths(gp_length)  # in-place edit of the caller's array, like create_cov
        ng, nm = self.gravfield.size, self.magfield.size
        return self.engine.posterior(A_g, A_m, self._sel, self.Fs3[:ng], self.Fs3[ng:ng + nm], self.Fs3[ng + nm:],
                                     [float(v) for v in lengths], self.coeffm if coeffm is None else coeffm,
                                     self.settings.kernelfunc, self.gp_sigma, gp_amp=gp_amp, props=self.props,
                                     calclogl=calclogl, want_mean_var=want_mean_var)

    # ---- inversion.py:77-122 ----------------------------------------------------------------------------------------
    def predict3(self, calclogl=False):
        """Mean, covariance (diagonal only) and log-likelihood of the GP with the 3x3 block kernel."""
        self.datastd = np.mean([np.nanstd(self.gravfield), np.nanstd(self.magfield), np.nanstd(self.drillfield)])
        try:
            r = self._run(self.gp_amp, self.gp_length, None, calclogl, True)
        except CholeskyError:
            print("Cholesky decompostion failed, AkA matrix i likely not positive semitive.")
            print("Change GP parameter settings")
            sys.exit(1)
        return r["mu"], DiagonalCovariance(r["var"]), r["logl"]

    # ---- inversion.py:125-152 ---------------------------------------------------------------------------------------
    def calc_logl(self, params):
        """Negative marginal log-likelihood for hyper-parameters (amplitude, lengthscale, 3 correlation coefficients)."""
        s = self.settings
        gp_amp = params[0]
        gp_length = params[1] * np.asarray([s.xvoxsize, s.xvoxsize, s.xvoxsize])
        coeffm = params[2:]
        try:
            r = self._run(gp_amp, gp_length, coeffm, True, False)
            logl = -0.5 * (r["uu"] + r["logdet"])  # no N log 2pi term here (inversion.py:147-149)
            if not np.isfinite(logl):
                logl = -np.inf
        except Exception:
            logl = -np.inf
        return -logl

    # ---- inversion.py:155-178 ---------------------------------------------------------------------------------------
    def optimize_gp(self):
        from scipy.optimize import shgo
        s = self.settings
        print("Optimizing GP hyperparameters and correlation coefficients, this may take a while...")
        self.datastd = np.mean([np.nanstd(self.gravfield), np.nanstd(self.magfield), np.nanstd(self.drillfield)])
        bopt_res = shgo(self.calc_logl, bounds=((0.5, 2), (0.5 * s.gp_lengthscale, 10 * s.gp_lengthscale),
                                                (0.5 * s.gp_coeff[0], 1), (0.5 * s.gp_coeff[1], 1), (0.5 * s.gp_coeff[2], 1)),
                        n=10, iters=10, sampling_method='sobol')
        if not bopt_res.success:
            print('WARNING: ' + bopt_res.message)
        else:
            print("Initial parameter [amplitude, lengthscale, corr1, corr2, corr3]:")
            print(self.gp_amp, self.gp_length, self.coeffm)
            self.gp_amp = bopt_res.x[0]
            # the reference stores the bare scalar here (inversion.py:175), which breaks create_cov's indexing
            # afterwards; the lengthscale is a multiple of the voxel size everywhere else (:48,:137), so keep that
            self.gp_length = bopt_res.x[1] * np.asarray([s.xvoxsize, s.xvoxsize, s.xvoxsize])
            self.coeffm = np.asarray([bopt_res.x[2:]]).flatten()
            print("Optimized parameter [amplitude, lengthscale, corr1, corr2, corr3]:")
            print(self.gp_amp, self.gp_length, self.coeffm)

    # ---- inversion.py:182-248 ---------------------------------------------------------------------------------------
    def cubing(self, gravfield, magfield, drillfield, sensor_locations, drilldata0):
        """Joint inversion and cubing of sensor data; returns the six cubes of the reference, each (yN, xN, zN)."""
        s = self.settings
        # no dtype cast: the reference z-scores the survey in the dtype it arrives in (float32 from a GeoTIFF
        # through scipy zoom, run_geobo.py:56-60) -- inversion.py:209-214
        self.gravfield = np.asarray(gravfield)
        self.magfield = np.asarray(magfield)
        self.drillfield = np.asarray(drillfield)
        self.sensor_locations = sensor_locations
        self.drilldata0 = drilldata0
        if not hasattr(self, "voxelpos"):
            self.create_cubegeometry()
        with np.errstate(all="ignore"):
            grav_mean, grav_std = self.gravfield.mean(), self.gravfield.std()
            gravfield_norm = (self.gravfield - grav_mean) / grav_std
            magn_mean, magn_std = self.magfield.mean(), self.magfield.std()
            magfield_norm = (self.magfield - magn_mean) / magn_std
            if self.drillfield.size:
                drill_mean, drill_std = self.drillfield.mean(), self.drillfield.std()
            else:
                drill_mean, drill_std = np.nan, np.nan
            drillfield_norm = (self.drillfield - drill_mean) / drill_std
        self.points3D = kernel.calcGridPoints3D((s.xNcube, s.yNcube, s.zNcube), (s.xvoxsize, s.yvoxsize, s.zvoxsize))
        self._sel = self._drill_selection()
        if self._sel.size != self.drillfield.size:
            raise ValueError("drillfield must hold one value per non-zero voxel of drilldata0")
        self.Fs3 = np.hstack((gravfield_norm, magfield_norm, drillfield_norm))
        if s.optimize_gp:
            self.optimize_gp()
        self.mu_rec, self.cov_rec, self.logl = self.predict3(calclogl=True)
        shp = (3, s.yNcube, s.xNcube, s.zNcube)
        results_rec = self.mu_rec.reshape(shp)
        results_var = self.cov_rec.diagonal().reshape(shp)
        with np.errstate(all="ignore"):
            density_rec = results_rec[0] * grav_std  # model represents deviation from the mean (means are not added back)
            density_var = results_var[0] * grav_std ** 2
            magsus_rec = results_rec[1] * magn_std
            magsus_var = results_var[1] * magn_std ** 2
            drill_rec = results_rec[2] * drill_std
            drill_var = results_var[2] * drill_std ** 2
        return density_rec, magsus_rec, drill_rec, density_var, magsus_var, drill_var
