"""Joint GP inversion -- the reference's `geobo/inversion.py` `Inversion` class on the MI355X path.

Same constructor-less usage, attributes and method signatures as the reference:

    inv = Inversion()                       # settings: geobo_amd.config_loader.load(...) (or pass settings=)
    voxelpos = inv.create_cubegeometry()
    density_rec, magsus_rec, drill_rec, density_var, magsus_var, drill_var = \
        inv.cubing(gravfield, magfield, drillfield, sensor_locations, drilldata0)

`method` selects how A K is formed: "dense" = fused fp64-MFMA contraction with the covariance tile generated in the
kernel (any grid), "spectral" = real-DFT route on batched MFMA GEMMs (regular grids with extents % 16 == 0, ~40x
faster at 64^3, same results to ~1e-13), "auto" (default) = spectral when applicable.  `assembly="f32"` keeps the covariance
tables and A K in fp32 (BASELINE config 5: fp32 kernel assembly + fp64 Cholesky; results at fp32-storage accuracy, ~1e-5),
`operators="streamed"` generates the forward operators in batches instead of keeping them resident; "auto" (default) does so where it costs nothing (lattice survey on one device: the transforms read the stencil table, AkA is the lattice Gram), "resident" never.

`cubing`/`predict3`/`calc_logl` run matrix-free on the GPU (engine.PosteriorEngine): D2, the 3N x 3N prior
and the 3N x 3N posterior covariance of the reference (inversion.py:92,117) are never formed -- only the
posterior diagonal that `cubing` consumes (inversion.py:238).  No CPU fallback exists.
"""
import sys

import numpy as np

from . import config_loader, geometry
from . import kernels as kernel
from . import sensormodel as sm
from .engine import CholeskyError, FactorisationTimeout, PosteriorEngine, create_cov_lengths


class DiagonalCovariance:
    """What `predict3` returns in place of the (3N,3N) posterior covariance unless `full_cov=True`: only its diagonal exists.

    The reference consumes nothing but `np.diag(self.cov_rec)` (inversion.py:238).  That idiom works on this object:
    `np.diag(obj)` / `np.diagonal(obj)` are answered through NumPy's `__array_function__` protocol with the stored diagonal
    (no (3N)^2 array is ever built); `.diagonal()` and `.shape` behave like the matrix's.  Anything that needs off-diagonal
    entries (`np.asarray(obj)`, arithmetic) raises TypeError -- ask `predict3(full_cov=True)` for the matrix on small cubes."""

    def __init__(self, diag):
        self._d = np.asarray(diag)
        self.shape = (self._d.size, self._d.size)
        self.ndim = 2
        self.dtype = self._d.dtype

    def diagonal(self, offset=0):
        if offset != 0:
            raise TypeError("only the main diagonal of the posterior covariance is kept")
        return self._d

    def __array_function__(self, func, types, args, kwargs):
        if func in (np.diag, np.diagonal) and args and args[0] is self:
            k = kwargs.get("k", kwargs.get("offset", args[1] if len(args) > 1 else 0))
            return self.diagonal(k)
        if func is np.shape:
            return self.shape
        return NotImplemented

    def __array__(self, dtype=None, copy=None):
        raise TypeError("the MI355X path keeps only the diagonal of the posterior covariance; use np.diag(obj) / "
                        ".diagonal(), or predict3(full_cov=True) on a cube small enough to hold (3N)^2 doubles")


def _zscore(v):
    """(v - mean) / std with the population std, and that std (inversion.py:209-214); NaN for an empty vector."""
    with np.errstate(all="ignore"):
        if not v.size:
            return v - np.nan, np.nan
        std = v.std()
        return (v - v.mean()) / std, std


class Inversion:
    """Class for inversion and reconstruction of 3D cubes from 2D sensor data (inversion.py:23-248)."""

    def __init__(self, settings=None, props=(0, 1, 2), rank=0, world=1, group=None, device=None, profile=False,
                 method="auto", assembly="f64", operators="auto"):
        self.settings = s = settings or config_loader.active()
        # inversion.py:46-51 -- NB x voxel size for all three length scales
        self.gp_length = s.gp_lengthscale * np.asarray([s.xvoxsize, s.xvoxsize, s.xvoxsize])
        self.gp_sigma = np.asarray(s.gp_err)
        self.coeffm = np.asarray(s.gp_coeff)
        self.gp_amp = 1.
        self.props = tuple(props)
        self._engine_args = dict(rank=rank, world=world, group=group, device=device, profile=profile, method=method,
                                 assembly=assembly, operators=operators)
        self._engine = None

    # ---- geometry (host, inversion.py:54-74) -------------------------------------------------------------------
    def create_cubegeometry(self):
        """Node grid `Edges` (3, yN+1, xN+1, zN+1; z negated), voxel centres `xxx, yyy, zzz` (yN, xN, zN) and the (3, N)
        centre list `voxelpos` it returns, all expanded from the 1-D axes of geobo_amd.geometry."""
        s = self.settings
        self.Edges = geometry.expand(*geometry.node_axes(s))
        centres = geometry.expand(*geometry.centre_axes(s))
        self.xxx, self.yyy, self.zzz = centres[0], centres[1], centres[2]
        self.voxelpos = centres.reshape(3, -1).copy()
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
            # the meshgrid check of the node tensor costs a millisecond at 64^3: once per Edges object
            if getattr(self, "_axes_of", (None, None))[0] is not self.Edges:
                self._axes_of = (self.Edges, sm._edge_axes(self.Edges, eng.nx, eng.ny, eng.nz))
            axes = self._axes_of[1]
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
        step = lambda: self.engine.posterior(A_g, A_m, self._sel, self.Fs3[:ng], self.Fs3[ng:ng + nm], self.Fs3[ng + nm:],
                                             [float(v) for v in lengths], self.coeffm if coeffm is None else coeffm,
                                             self.settings.kernelfunc, self.gp_sigma, gp_amp=gp_amp, props=self.props,
                                             calclogl=calclogl, want_mean_var=want_mean_var)
        try:
            return step()
        except FactorisationTimeout:
            # (round-5 advisory) the tile DAG's bounded spins can trip on a device shared with other processes: once more, on the
            # stream schedule of rounds 2-4 (no inter-workgroup hand-offs), instead of returning an undefined factor
            import os
            import warnings
            warnings.warn("geobo_potrf_inv: tile-DAG hand-off timed out (info = -7); repeating the step on the stream schedule", RuntimeWarning)
            before = os.environ.get("GEOBO_POTRF")
            os.environ["GEOBO_POTRF"] = "streams"
            try:
                return step()
            finally:
                if before is None:
                    os.environ.pop("GEOBO_POTRF", None)
                else:
                    os.environ["GEOBO_POTRF"] = before

    # ---- inversion.py:77-122 ----------------------------------------------------------------------------------------
    def predict3(self, calclogl=False, full_cov=False):
        """Mean, covariance and log-likelihood of the GP with the 3x3 block kernel.
        The covariance is a diagonal-backed object (`np.diag(cov)` works, inversion.py:238) unless `full_cov=True`, which
        returns the (3N, 3N) matrix K - V^T V of inversion.py:117 -- only for cubes where 9 N^2 doubles fit (tests, small N)."""
        self.datastd = np.mean([np.nanstd(self.gravfield), np.nanstd(self.magfield), np.nanstd(self.drillfield)])
        try:
            r = self._run(self.gp_amp, self.gp_length, None, calclogl, True)
        except CholeskyError:
            if self.settings.kernelfunc == "matern32" and len(set(np.asarray(self.gp_length, dtype=float).tolist())) < 3:
                # the reference exits here too (its Matern cross term is 0/0 at equal lengths, kernels.py:148-156, and
                # create_cov turns the default [l,l,l] into [l,1.02l,l]); say why before the two reference lines
                print("matern32 needs three DISTINCT length scales: gp_length = %s has equal entries, which makes the "
                      "cross-covariance NaN. Set e.g. inv.gp_length = l * np.array([1.00, 1.02, 1.04])." % (self.gp_length,))
            print("Cholesky decompostion failed, AkA matrix i likely not positive semitive.")
            print("Change GP parameter settings")
            sys.exit(1)
        cov = DiagonalCovariance(r["var"])
        if full_cov:
            cov = self.engine.posterior_covariance(self.settings.kernelfunc, r["lengths"], self.coeffm, self.gp_amp)
        return r["mu"], cov, r["logl"]

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
    def hyper_bounds(self):
        """Search box of optimize_gp: amplitude, lengthscale (in x-voxels) and the three cross-correlation weights."""
        s = self.settings
        box = [(0.5, 2), (0.5 * s.gp_lengthscale, 10 * s.gp_lengthscale)]
        return tuple(box + [(0.5 * w, 1) for w in s.gp_coeff])

    def set_hyperparameters(self, x):
        """Adopt a hyper-parameter vector (amplitude, lengthscale in x-voxels, w1, w2, w3).
        The reference keeps the bare lengthscale scalar in gp_length at this point (inversion.py:175), which its own create_cov
        can no longer index; everywhere else gp_length is lengthscale * x-voxel size for all three blocks (:48, :137), so
        that is what is stored here."""
        x = np.asarray(x, dtype=float)
        self.gp_amp = x[0]
        self.gp_length = x[1] * np.full(3, self.settings.xvoxsize)
        self.coeffm = x[2:5].copy()

    def optimize_gp(self):
        """Maximise the marginal likelihood over the box of hyper_bounds() with SciPy's SHGO (10 Sobol points x 10
        iterations, as the reference); every objective evaluation is one AkA + Cholesky + log-det on the device."""
        from scipy.optimize import shgo
        print("Optimizing GP hyperparameters and correlation coefficients, this may take a while...")
        self.datastd = np.mean([np.nanstd(v) for v in (self.gravfield, self.magfield, self.drillfield)])
        found = shgo(self.calc_logl, bounds=self.hyper_bounds(), n=10, iters=10, sampling_method="sobol")
        if not found.success:
            print("WARNING: " + found.message)     # parameters stay as they were
            return found
        report = lambda title: print(title + "\n" + " ".join(str(v) for v in (self.gp_amp, self.gp_length, self.coeffm)))
        report("Initial parameter [amplitude, lengthscale, corr1, corr2, corr3]:")
        self.set_hyperparameters(found.x)
        report("Optimized parameter [amplitude, lengthscale, corr1, corr2, corr3]:")
        return found

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
        # population z-score of each data vector in the dtype it arrives in; an empty drill vector gives NaN statistics
        # (and NaN drill cubes), like the reference
        gravfield_norm, grav_std = _zscore(self.gravfield)
        magfield_norm, magn_std = _zscore(self.magfield)
        drillfield_norm, drill_std = _zscore(self.drillfield)
        gkey = (s.xNcube, s.yNcube, s.zNcube, s.xvoxsize, s.yvoxsize, s.zvoxsize)
        if getattr(self, "_points_key", None) != gkey:
            self._points_key, self._points3D = gkey, kernel.calcGridPoints3D(gkey[:3], gkey[3:])
        self.points3D = self._points3D
        self._sel = self._drill_selection()
        if self._sel.size != self.drillfield.size:
            raise ValueError("drillfield must hold one value per non-zero voxel of drilldata0")
        self.Fs3 = np.hstack((gravfield_norm, magfield_norm, drillfield_norm))
        if s.optimize_gp:
            self.optimize_gp()
        self.mu_rec, self.cov_rec, self.logl = self.predict3(calclogl=True)
        shape = (3, s.yNcube, s.xNcube, s.zNcube)
        mean_cubes = self.mu_rec.reshape(shape)
        var_cubes = np.diag(self.cov_rec).reshape(shape)           # the reference's own idiom (inversion.py:238)
        # deviations from the data means, back in data units (the means themselves are not restored, inversion.py:242-247)
        # (the std scalars keep the survey's dtype: a float32 survey squares its std in float32, as the reference does)
        scale = (grav_std, magn_std, drill_std)
        with np.errstate(all="ignore"):
            rec = [mean_cubes[i] * scale[i] for i in range(3)]
            var = [var_cubes[i] * scale[i] ** 2 for i in range(3)]
        return rec[0], rec[1], rec[2], var[0], var[1], var[2]
