"""CPU oracle for the GeoBO joint-inversion hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference algorithm (sebhaan/geobo,
`geobo/kernels.py`, `geobo/sensormodel.py`, `geobo/inversion.py`; citations below are
`file:line` inside /root/reference).  It exists to *check* the HIP path:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg import it;
  * the product package `geobo_amd` never imports, links or executes anything from here.

Parity pinning: PINNED.  `tests/test_oracle_golden.py` checks every function in this file
against golden vectors produced by running the reference itself (tests/golden/make_golden.py:
kernel known answers, a non-cubic 10x8x6 grid, 16^3 cubes for all three kernels, the two
shipped examples incl. the committed examples/results/*/*.vtk cubes, and the forward-model
known answer simcube_cylinders.csv -> simsurveydata_cylinders.csv).

Two forms of the posterior are provided:
  * `posterior_dense`   -- reference-shaped (materialises D2, the 3N x 3N prior and the full
                           posterior covariance exactly like inversion.py:77-122); usable to ~20^3;
  * `posterior_blocked` -- the same mathematics, matrix-free and column-blocked (never holds K);
                           this is the bridge to 32^3 and the `cpu_baseline` "port" that bench.py times;
  * `posterior_fft`     -- `posterior_blocked` with the sensor rows of A K by FFT convolution on the regular grid
                           (`ak_rows_fft`): whole-cube goldens at 64 x 48 x 64 (tests/golden/make_oracle64.py).

Conventions: grid nx,ny,nz; flat voxel index p = (iy*nx + ix)*nz + iz;
property blocks 0 = density (gravity rows), 1 = magnetic susceptibility (magnetic rows),
2 = drill property (drill rows); cross weights (w1,w2,w3) = (0<->2, 1<->2, 0<->1).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.linalg import cholesky, solve_triangular

KERNELS = ("exp", "sparse", "matern32")


# ------------------------------------------------------------------------------------------------
# settings / geometry  (config_loader.py:41-59, inversion.py:46-74)
# ------------------------------------------------------------------------------------------------
@dataclass
class Grid:
    nx: int
    ny: int
    nz: int
    xmin: float = 0.0
    xmax: float = 1.0
    ymin: float = 0.0
    ymax: float = 1.0
    zmax: float = 0.0
    zLcube: float = 1.0
    zoff: float = 1.0
    gp_lengthscale: float = 2.0
    gp_err: tuple = (0.1, 0.1, 0.1)
    gp_coeff: tuple = (1.0, 0.2, 0.2)
    kernelfunc: str = "sparse"
    mag: tuple = (0.0, 0.0, 1.0)                 # XMAG, YMAG, ZMAG
    c_G: float = 6.673848e-11
    c_SI_TO_MILLIGALS: float = 10000
    c_GCM3_TO_SI: float = 1000.0
    fcor_grav: float = 1.0
    fcor_mag: float = 0.001
    derived: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        # config_loader.py:41-59
        self.xL = self.xmax - self.xmin
        self.yL = self.ymax - self.ymin
        self.sx = self.xL / self.nx * 1.
        self.sy = self.yL / self.ny * 1.
        self.sz = self.zLcube / self.nz * 1.
        self.B = np.asarray(self.mag, dtype=float) * 1e-3
        self.c_mgal = self.c_G * self.c_SI_TO_MILLIGALS * self.c_GCM3_TO_SI
        self.N = self.nx * self.ny * self.nz

    @classmethod
    def from_settings(cls, s):
        """`s` = dict with the reference's YAML keys (examples/settings_example1.yaml)."""
        return cls(nx=s["xNcube"], ny=s["yNcube"], nz=s["zNcube"], xmin=s["xmin"], xmax=s["xmax"],
                   ymin=s["ymin"], ymax=s["ymax"], zmax=s["zmax"], zLcube=s["zLcube"], zoff=s["zoff"],
                   gp_lengthscale=s["gp_lengthscale"], gp_err=tuple(s["gp_err"]), gp_coeff=tuple(s["gp_coeff"]),
                   kernelfunc=s["kernelfunc"], mag=(s["XMAG"], s["YMAG"], s["ZMAG"]), c_G=s["c_G"],
                   c_SI_TO_MILLIGALS=s["c_SI_TO_MILLIGALS"], c_GCM3_TO_SI=s["c_GCM3_TO_SI"],
                   fcor_grav=s["fcor_grav"], fcor_mag=s["fcor_mag"])

    # inversion.py:46-51 -- NB: the x voxel size is used for all three properties
    def default_gp_length(self):
        return self.gp_lengthscale * np.asarray([self.sx, self.sx, self.sx])

    # inversion.py:58-66 -- node ("edge") coordinates, z axis negated
    def edges(self):
        xe = np.linspace(0, self.nx, self.nx + 1) * self.sx
        ye = np.linspace(0, self.ny, self.ny + 1) * self.sy
        ze = np.linspace(0, -self.nz, self.nz + 1) * self.sz + self.zmax
        X, Y, Z = np.meshgrid(xe, ye, ze)          # shape (ny+1, nx+1, nz+1)
        return np.asarray([X, Y, -Z])

    # inversion.py:67-74 -- voxel centres, arrays of shape (ny, nx, nz)
    def voxel_centres(self):
        xc = np.arange(self.sx / 2., self.xL + self.sx / 2., self.sx)
        yc = np.arange(self.sy / 2., self.yL + self.sy / 2., self.sy)
        zc = self.zmax - np.arange(self.sz / 2., self.zLcube + self.sz / 2., self.sz)
        return np.meshgrid(xc, yc, zc)

    # run_geobo.py:61-65 -- sensors above the voxel centres
    def sensor_locations(self):
        xs = np.linspace(0.5, self.nx - 0.5, self.nx) * self.sx
        ys = np.linspace(0.5, self.ny - 0.5, self.ny) * self.sy
        X, Y, Z = np.meshgrid(xs, ys, self.zmax + self.zoff)
        return np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T


# ------------------------------------------------------------------------------------------------
# kernels.py
# ------------------------------------------------------------------------------------------------
def grid_points(n, vox):
    """kernels.py:27-42 -- (N,3) coordinates, x=(ix+1)*sx ..., row p=(iy*nx+ix)*nz+iz."""
    nx, ny, nz = n
    sx, sy, sz = vox
    X, Y, Z = np.meshgrid(np.arange(1, nx + 1) * sx, np.arange(1, ny + 1) * sy, np.arange(1, nz + 1) * sz)
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)


def sqdist(P, Q=None):
    """kernels.py:45-61 -- D2[p,q] = 0 + dx^2 + dy^2 + dz^2 with d = Q[q]-P[p] (that summation order)."""
    Q = P if Q is None else Q
    acc = 0
    for d in range(P.shape[1]):
        acc = acc + (Q[None, :, d] - P[:, None, d]) ** 2
    return acc


def k_auto(name, d2, l):
    """Auto-covariance blocks: kernels.py:81-88 (exp), :101-114 (sparse), :140-146 (matern32)."""
    d2 = np.asarray(d2, dtype=float)
    if name == "exp":
        return np.exp(-0.5 * d2 / l ** 2)
    if name == "matern32":
        nu = np.sqrt(3) * np.sqrt(d2) / l
        return (1 + nu) * np.exp(-nu)
    if name == "sparse":
        d = np.atleast_1d(np.sqrt(d2))
        inside = d < l
        t = d[inside]
        out = np.zeros_like(d)
        out[inside] = (2 + np.cos(2 * np.pi * t / l)) / 3. * (1 - t / l) + 1 / (2. * np.pi) * np.sin(2 * np.pi * t / l)
        out[out < 0.] = 0.
        return out.reshape(np.shape(d2)) if np.ndim(d2) else out[0]
    raise ValueError(name)


def k_cross(name, d2, l1, l2):
    """Cross-covariance blocks: kernels.py:90-99 (exp), :116-138 (sparse), :148-156 (matern32).

    Quirks kept: the sparse equal-length offset `l2 += 1e-3*l2` (:125-126); branch A of the sparse
    cross kernel has the cos *inside* the sin argument (:133); branch B is assigned after A and
    wins at equality (:135); matern32 is singular (NaN/inf) at l1 == l2."""
    d2 = np.asarray(d2, dtype=float)
    if name == "exp":
        return np.sqrt(2. * l1 * l2 / (l1 ** 2 + l2 ** 2)) * np.exp(-d2 / (l1 ** 2 + l2 ** 2))
    if name == "matern32":
        with np.errstate(divide="ignore", invalid="ignore"):
            norm = 2 * np.sqrt(l1 * l2) / (l1 ** 2 - l2 ** 2)
            return norm * (l1 * np.exp(-np.sqrt(3 * d2) / l1) - l2 * np.exp(-np.sqrt(3 * d2) / l2))
    if name == "sparse":
        d = np.atleast_1d(np.sqrt(d2))
        if l1 == l2:
            l2 = l2 + 1e-3 * l2
        lmean = np.mean([l1, l2])
        lmin = np.min([l1, l2])
        lmax = np.max([l1, l2])
        out = np.zeros_like(d)
        a = d <= abs(l2 - l1) / 2.
        da = d[a]
        out[a] = 2. / (3 * np.sqrt(l1 * l2)) * (
            lmin + 1 / np.pi * lmax ** 3 / (lmax ** 2 - lmin ** 2) * np.sin(np.pi * lmin / lmax * np.cos(2 * np.pi * da / lmax)))
        b = (d >= abs(l2 - l1) / 2.) & (d <= (l1 + l2) / 2.)
        db = d[b]
        out[b] = 2. / (3 * np.sqrt(l1 * l2)) * (
            lmean - db + l1 ** 3 * np.sin(np.pi * (l2 - 2. * db) / l1) / (2 * np.pi * (l1 ** 2 - l2 ** 2))
            - l2 ** 3 * np.sin(np.pi * (l1 - 2. * db) / l2) / (2 * np.pi * (l1 ** 2 - l2 ** 2)))
        out[out < 0.] = 0.
        return out.reshape(np.shape(d2)) if np.ndim(d2) else out[0]
    raise ValueError(name)


def mutate_lengths(gl):
    """kernels.py:174-180 -- create_cov aliases the caller's array and edits it IN PLACE.
    Default [l,l,l] becomes [l, 1.02*l, l] (the second rule writes index 1, sic)."""
    p = np.asarray(gl)
    if p[1] == p[0]:
        p[1] = 1.01 * p[0]
    if p[2] == p[0]:
        p[1] = 1.02 * p[0]
    if p[2] == p[1]:
        p[2] = 1.01 * p[1]
    return p


def weight_matrix(w):
    """kernels.py:166-169,181 -- (w1,w2,w3) -> 3x3 block weights."""
    w1, w2, w3 = np.asarray(w, dtype=float)
    return np.array([[1., w3, w1], [w3, 1., w2], [w1, w2, 1.]])


def k_block(name, d2, lengths, W, i, j):
    """Block (i,j) of the 3x3 prior for squared distances d2 (lengths already mutated)."""
    if i == j:
        return k_auto(name, d2, lengths[i])
    return W[i, j] * k_cross(name, d2, lengths[i], lengths[j])


def create_cov(D2, gplength, crossweights=(1, 1, 1), fkernel="sparse"):
    """kernels.py:158-195 -- (3N,3N) prior; mutates `gplength` like the reference."""
    p = mutate_lengths(gplength)
    W = weight_matrix(crossweights)
    rows = [np.hstack([k_block(fkernel, D2, p, W, i, j) for j in range(3)]) for i in range(3)]
    return np.vstack(rows)


# ------------------------------------------------------------------------------------------------
# sensormodel.py
# ------------------------------------------------------------------------------------------------
def grav_potential(x, y, z):
    """sensormodel.py:96-110."""
    eps = 1e-9
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    return x * np.log(y + r) + y * np.log(x + r) - z * np.arctan((x * y) / (z * r + eps))


def magn_potential(x, y, z, bx, by, bz):
    """sensormodel.py:113-133."""
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    normB = np.sqrt(bx * bx + by * by + bz * bz)
    f = 1. / normB * ((2. * by * bz * np.log(x + r)) + (2. * bz * bx * np.log(y + r)) + (2. * by * bx * np.log(z + r))
                      + (bz * bz - by * by) * np.arctan((x * z) / (y * r)) + (bz * bz - bx * bx) * np.arctan((y * z) / (x * r)))
    return -f


def a_sens(grid: Grid, B, loc, edges, func, rows=None):
    """sensormodel.py:29-93 -- (M_s, N) prism sensitivities.

    Per sensor: node offsets, +-1e6 m padding applied on axis 0 (= iy) to BOTH the x and y
    offsets (:63-68), node potential, 8-corner alternating sum evaluated left to right (:81-86),
    unit scaling (:88-91).  Vectorised over nodes; one Python iteration per sensor."""
    ny, nx, nz = grid.ny, grid.nx, grid.nz
    xe, ye, ze = edges[0], edges[1], edges[2]
    rows = range(nx * ny) if rows is None else rows
    out = np.empty((len(rows), grid.N))
    far = 1e6
    for o, n in enumerate(rows):
        x0 = xe - loc[n, 0]
        y0 = ye - loc[n, 1]
        z0 = ze - loc[n, 2]
        x0[0] -= far
        y0[0] -= far
        x0[-1] += far
        y0[-1] += far
        with np.errstate(all="ignore"):
            e = grav_potential(x0, y0, z0) if func == "grav" else magn_potential(x0, y0, z0, B[0], B[1], B[2])
        hi = (e[1:, 1:, 1:] - e[1:, 1:, :-1] - e[1:, :-1, 1:] + e[1:, :-1, :-1])
        lo = (e[:-1, 1:, 1:] - e[:-1, 1:, :-1] - e[:-1, :-1, 1:] + e[:-1, :-1, :-1])
        out[o] = (-(hi - lo)).reshape(-1)
    if func == "grav":
        return grid.c_mgal * out / grid.fcor_grav
    return out / grid.fcor_mag


def drill_selection(drilldata0):
    """inversion.py:219 + sensormodel.py:136-153 -- A_drill is a 0/1 selection of the voxels with
    non-zero drill data, rows in ascending flat index; returned as the index list."""
    return np.flatnonzero(np.asarray(drilldata0).reshape(-1) != 0)


# ------------------------------------------------------------------------------------------------
# inversion.py
# ------------------------------------------------------------------------------------------------
def _noise(gp_sigma, mg, mm, md):
    return np.hstack([np.full(mg, gp_sigma[0]), np.full(mm, gp_sigma[1]), np.full(md, gp_sigma[2])])


def posterior_dense(P3, A_g, A_m, sel, y, lengths, W, name, gp_sigma, gp_amp=1.0, n_voxels_for_logl=None, return_cov=False):
    """inversion.py:77-122, literally: full K, A3 (K A3^T), Cholesky, V, mu, full covariance."""
    N = P3.shape[0]
    mg, mm, md = A_g.shape[0], A_m.shape[0], len(sel)
    D2 = sqdist(P3)
    K = gp_amp * np.vstack([np.hstack([k_block(name, D2, lengths, W, i, j) for j in range(3)]) for i in range(3)])
    A3 = np.zeros((mg + mm + md, 3 * N))
    A3[:mg, :N] = A_g
    A3[mg:mg + mm, N:2 * N] = A_m
    A3[mg + mm + np.arange(md), 2 * N + sel] = 1.
    AkA = A3 @ (K @ A3.T) + np.diag(_noise(gp_sigma, mg, mm, md) ** 2)
    L = cholesky(AkA, lower=True)
    u = solve_triangular(L, y, lower=True)
    nlog = (N if n_voxels_for_logl is None else n_voxels_for_logl) * np.log(2 * np.pi)
    logl = -0.5 * (u @ u + np.log(np.diag(L) ** 2).sum() + nlog)
    V = solve_triangular(L, A3 @ K, lower=True)
    mu = V.T @ u
    cov = K - V.T @ V
    res = dict(mu=mu, var=np.diag(cov).copy(), logl=logl, AkA=AkA, L=L, u=u)
    if return_cov:
        res["cov"] = cov
    return res


def posterior_blocked(P3, A_g, A_m, sel, y, lengths, W, name, gp_sigma, gp_amp=1.0, props=(0, 1, 2),
                      block=2048, return_AK=False):
    """Same result as `posterior_dense`, matrix-free (SURVEY.md section 8(a) "algorithm"):

      AK[s-rows, j-cols] = A_s K_sj,  AK[d-rows, j-cols] = K_2j[sel, :]        (K generated per column block)
      AkA = AK A3^T + diag(sigma^2);  L = chol(AkA);  u = L^-1 y
      V = L^-1 AK;  mu_q = sum_m V_mq u_m;  var_q = amp*k_jj(0) - sum_m V_mq^2   (k_jj(0) = 1)

    Only the property column blocks listed in `props` are produced (others are NaN)."""
    N = P3.shape[0]
    mg, mm, md = A_g.shape[0], A_m.shape[0], len(sel)
    M = mg + mm + md
    ops = ((0, A_g, slice(0, mg)), (1, A_m, slice(mg, mg + mm)))
    AK = {j: np.empty((M, N)) for j in props}
    for c0 in range(0, N, block):
        c1 = min(N, c0 + block)
        D2 = sqdist(P3, P3[c0:c1])                       # (N, b): rows p (contracted), cols q
        for j in props:
            for s, A, rs in ops:
                AK[j][rs, c0:c1] = A @ (gp_amp * k_block(name, D2, lengths, W, s, j))
            if md:
                AK[j][mg + mm:, c0:c1] = gp_amp * k_block(name, D2[sel], lengths, W, 2, j)
    # AkA: sensor columns by GEMM; drill columns by symmetry / direct evaluation
    AkA = np.zeros((M, M))
    if 0 in props:
        AkA[:, :mg] = AK[0] @ A_g.T
    else:
        raise ValueError("property block 0 required")
    AkA[:, mg:mg + mm] = AK[1] @ A_m.T
    if md:
        AkA[:mg + mm, mg + mm:] = AkA[mg + mm:, :mg + mm].T
        AkA[mg + mm:, mg + mm:] = gp_amp * k_block(name, sqdist(P3[sel]), lengths, W, 2, 2)
    AkA = AkA + np.diag(_noise(gp_sigma, mg, mm, md) ** 2)
    L = cholesky(AkA, lower=True)
    u = solve_triangular(L, y, lower=True)
    logl = -0.5 * (u @ u + np.log(np.diag(L) ** 2).sum() + N * np.log(2 * np.pi))
    mu = np.full(3 * N, np.nan)
    var = np.full(3 * N, np.nan)
    for j in props:
        for c0 in range(0, N, 4 * block):
            c1 = min(N, c0 + 4 * block)
            V = solve_triangular(L, AK[j][:, c0:c1], lower=True)
            mu[j * N + c0:j * N + c1] = V.T @ u
            var[j * N + c0:j * N + c1] = gp_amp * 1.0 - np.einsum("mq,mq->q", V, V)
    res = dict(mu=mu, var=var, logl=logl, AkA=AkA, L=L, u=u)
    if return_AK:
        res["AK"] = AK
    return res


def cubing(grid: Grid, gravfield, magfield, drillfield, sensor_locations, drilldata0, gp_length=None,
           dense=False, props=(0, 1, 2), A=None, block=2048, gp_amp=1.0, fft=False, workers=1):
    """inversion.py:182-248 -- z-score the data (population std), build operators, posterior,
    reshape to (3, ny, nx, nz), scale by the data std / std^2 (means are NOT added back).

    Returns dict(cubes=(6, ny, nx, nz) in the reference's return order, mu, var, logl, gp_length
    (after the create_cov mutation), A_g, A_m, sel, Fs3)."""
    # NB no dtype cast: run_geobo.py hands float32 survey arrays (scipy zoom of a float32 GeoTIFF) and the
    # reference z-scores them in that dtype (inversion.py:209-214) -- observable at the 1e-7 level
    gravfield, magfield, drillfield = np.asarray(gravfield), np.asarray(magfield), np.asarray(drillfield)
    with np.errstate(all="ignore"):
        gs, ms = gravfield.std(), magfield.std()
        ds = drillfield.std() if drillfield.size else np.nan
        y = np.hstack([(gravfield - gravfield.mean()) / gs, (magfield - magfield.mean()) / ms,
                       (drillfield - drillfield.mean()) / ds if drillfield.size else drillfield])
    P3 = grid_points((grid.nx, grid.ny, grid.nz), (grid.sx, grid.sy, grid.sz))
    sel = drill_selection(drilldata0)
    edges = grid.edges()
    if A is None:
        A_g = a_sens(grid, grid.B * 0., sensor_locations, edges, "grav")
        A_m = a_sens(grid, grid.B, sensor_locations, edges, "magn")
    else:
        A_g, A_m = A
    gl = grid.default_gp_length() if gp_length is None else gp_length
    lengths = mutate_lengths(gl)
    W = weight_matrix(grid.gp_coeff)
    if dense:
        r = posterior_dense(P3, A_g, A_m, sel, y, lengths, W, grid.kernelfunc, grid.gp_err, gp_amp=gp_amp)
    elif fft:
        r = posterior_fft(grid, P3, A_g, A_m, sel, y, lengths, W, grid.kernelfunc, grid.gp_err, gp_amp=gp_amp, props=props,
                          workers=workers)
    else:
        r = posterior_blocked(P3, A_g, A_m, sel, y, lengths, W, grid.kernelfunc, grid.gp_err, gp_amp=gp_amp, props=props,
                              block=block)
    shp = (3, grid.ny, grid.nx, grid.nz)
    rec = r["mu"].reshape(shp)
    var = r["var"].reshape(shp)
    with np.errstate(all="ignore"):
        cubes = np.asarray([rec[0] * gs, rec[1] * ms, rec[2] * ds, var[0] * gs ** 2, var[1] * ms ** 2, var[2] * ds ** 2])
    r.update(cubes=cubes, gp_length=lengths, A_g=A_g, A_m=A_m, sel=sel, Fs3=y)
    return r


def neg_logl(grid: Grid, params, P3, A_g, A_m, sel, y):
    """inversion.py:125-152 (calc_logl) -- negative marginal log-likelihood for (amplitude, lengthscale in x-voxels, w1, w2, w3);
    no N log 2pi term (:147-149); any failure -> +inf (:150-152)."""
    try:
        lengths = mutate_lengths(params[1] * np.asarray([grid.sx, grid.sx, grid.sx]))
        r = posterior_dense(P3, A_g, A_m, sel, y, lengths, weight_matrix(params[2:]), grid.kernelfunc, grid.gp_err,
                            gp_amp=params[0])
        v = 0.5 * (r["u"] @ r["u"] + np.log(np.diag(r["L"]) ** 2).sum())
        return v if np.isfinite(v) else np.inf
    except Exception:
        return np.inf


def ak_row_fft(grid: Grid, a_row, name, lengths, W, s, j, gp_amp=1.0):
    """One row of A K for block (s, j):  w[q] = sum_p a[p] K_sj[p, q], by a zero-padded FFT convolution -- on the grid of
    calcGridPoints3D (kernels.py:27-42) K_sj[p, q] = k(dy*sy, dx*sx, dz*sz) depends on the index difference only.  Same numbers
    as `a_row @ k_block(name, sqdist(P3), ...)` (checked at small sizes in tests/test_oracle_golden.py); this is the form that
    reaches 64^3 on a CPU, for independent spot checks of rows of A K and entries of AkA."""
    ny, nx, nz = grid.ny, grid.nx, grid.nz
    dy = np.arange(-(ny - 1), ny) * grid.sy
    dx = np.arange(-(nx - 1), nx) * grid.sx
    dz = np.arange(-(nz - 1), nz) * grid.sz
    d2 = 0 + (dx[None, :, None]) ** 2 + (dy[:, None, None]) ** 2 + (dz[None, None, :]) ** 2   # same summation order as sqdist
    kk = gp_amp * k_block(name, d2, lengths, W, s, j)                                        # (2ny-1, 2nx-1, 2nz-1), offset n-1
    shape = (2 * ny, 2 * nx, 2 * nz)
    kpad = np.zeros(shape)
    kpad[:2 * ny - 1, :2 * nx - 1, :2 * nz - 1] = kk
    kpad = np.roll(kpad, (-(ny - 1), -(nx - 1), -(nz - 1)), axis=(0, 1, 2))                  # offset 0 at index 0, negatives wrapped
    apad = np.zeros(shape)
    apad[:ny, :nx, :nz] = np.asarray(a_row).reshape(ny, nx, nz)
    w = np.fft.irfftn(np.fft.rfftn(apad) * np.fft.rfftn(kpad), s=shape, axes=(0, 1, 2))
    return w[:ny, :nx, :nz].reshape(-1)


def _lattice_spectrum(grid: Grid, name, lengths, W, s, j, gp_amp):
    """rfftn of block (s, j) of the prior on the index-difference lattice, zero-padded to (2ny, 2nx, 2nz) with offset 0 at index 0
    (the kernel half of `ak_row_fft`, same arithmetic)."""
    from scipy import fft as sfft
    ny, nx, nz = grid.ny, grid.nx, grid.nz
    dy = np.arange(-(ny - 1), ny) * grid.sy
    dx = np.arange(-(nx - 1), nx) * grid.sx
    dz = np.arange(-(nz - 1), nz) * grid.sz
    d2 = 0 + (dx[None, :, None]) ** 2 + (dy[:, None, None]) ** 2 + (dz[None, None, :]) ** 2
    kk = gp_amp * k_block(name, d2, lengths, W, s, j)
    shape = (2 * ny, 2 * nx, 2 * nz)
    kpad = np.zeros(shape)
    kpad[:2 * ny - 1, :2 * nx - 1, :2 * nz - 1] = kk
    kpad = np.roll(kpad, (-(ny - 1), -(nx - 1), -(nz - 1)), axis=(0, 1, 2))
    return sfft.rfftn(kpad)


def ak_rows_fft(grid: Grid, A_rows, name, lengths, W, s, js, gp_amp=1.0, workers=1, batch=16, out=None):
    """Rows of A K_sj for a stack of operator rows (R, N) and every block j in `js`: `ak_row_fft` with the row's forward
    transform shared between the blocks and the kernel spectra computed once (scipy's pocketfft, double precision, `workers`
    threads over the batch).  Returns {j: (R, N)} (or fills `out[j]`).  Pinned to `ak_row_fft` / the direct contraction in
    tests/test_oracle_golden.py."""
    from scipy import fft as sfft
    ny, nx, nz = grid.ny, grid.nx, grid.nz
    shape = (2 * ny, 2 * nx, 2 * nz)
    spec = {j: _lattice_spectrum(grid, name, lengths, W, s, j, gp_amp) for j in js}
    A_rows = np.asarray(A_rows)
    R = A_rows.shape[0]
    res = out if out is not None else {j: np.empty((R, grid.N)) for j in js}
    for r0 in range(0, R, batch):
        r1 = min(R, r0 + batch)
        apad = np.zeros((r1 - r0,) + shape)
        apad[:, :ny, :nx, :nz] = A_rows[r0:r1].reshape(r1 - r0, ny, nx, nz)
        fa = sfft.rfftn(apad, axes=(1, 2, 3), workers=workers)
        for j in js:
            w = sfft.irfftn(fa * spec[j][None], s=shape, axes=(1, 2, 3), workers=workers)
            res[j][r0:r1] = w[:, :ny, :nx, :nz].reshape(r1 - r0, -1)
    return res


def posterior_fft(grid: Grid, P3, A_g, A_m, sel, y, lengths, W, name, gp_sigma, gp_amp=1.0, props=(0, 1, 2), workers=1,
                  block=8192, log=None):
    """`posterior_blocked` with the sensor rows of A K formed by FFT convolution on the grid of calcGridPoints3D (`ak_rows_fft`)
    instead of the column-blocked contraction: the form of inversion.py:77-122 that reaches 64 x 48 x 64 on a CPU (whole-cube
    goldens for the grids on which the radix-2 / lattice kernels of the HIP path run).  Same AkA, Cholesky (scipy), V = L^-1 (A K)
    column-blocked, mu and var as `posterior_blocked`; checked against it in tests/test_oracle_golden.py."""
    N = P3.shape[0]
    mg, mm, md = A_g.shape[0], A_m.shape[0], len(sel)
    M = mg + mm + md
    say = log if log is not None else (lambda *a: None)
    AK = {j: np.empty((M, N)) for j in props}
    for s, A, r0 in ((0, A_g, 0), (1, A_m, mg)):
        ak_rows_fft(grid, A, name, lengths, W, s, props, gp_amp=gp_amp, workers=workers,
                    out={j: AK[j][r0:r0 + A.shape[0]] for j in props})
        say("A K rows of operator %d" % s)
    if md:
        D2s = sqdist(P3[sel], P3)                       # (md, N): rows p = drilled voxels
        for j in props:
            AK[j][mg + mm:] = gp_amp * k_block(name, D2s, lengths, W, 2, j)
    if 0 not in props or 1 not in props:
        raise ValueError("property blocks 0 and 1 required")
    AkA = np.zeros((M, M))
    AkA[:, :mg] = AK[0] @ A_g.T
    AkA[:, mg:mg + mm] = AK[1] @ A_m.T
    if md:
        AkA[:mg + mm, mg + mm:] = AkA[mg + mm:, :mg + mm].T
        AkA[mg + mm:, mg + mm:] = gp_amp * k_block(name, sqdist(P3[sel]), lengths, W, 2, 2)
    AkA = AkA + np.diag(_noise(gp_sigma, mg, mm, md) ** 2)
    say("AkA")
    L = cholesky(AkA, lower=True)
    u = solve_triangular(L, y, lower=True)
    logl = -0.5 * (u @ u + np.log(np.diag(L) ** 2).sum() + N * np.log(2 * np.pi))
    mu = np.full(3 * N, np.nan)
    var = np.full(3 * N, np.nan)
    for j in props:
        for c0 in range(0, N, block):
            c1 = min(N, c0 + block)
            V = solve_triangular(L, AK[j][:, c0:c1], lower=True)
            mu[j * N + c0:j * N + c1] = V.T @ u
            var[j * N + c0:j * N + c1] = gp_amp * 1.0 - np.einsum("mq,mq->q", V, V)
        say("posterior of block %d" % j)
    return dict(mu=mu, var=var, logl=logl, AkA=AkA, L=L, u=u)


# ------------------------------------------------------------------------------------------------
# synthetic survey used by the tests and the benchmark (simcube.py:83-92,147-150)
# ------------------------------------------------------------------------------------------------
def synthetic_truth(grid: Grid):
    """Reference 'cylinders' model (simcube.py:83-92) plus a smooth trend so that every grid
    size gives non-degenerate survey/drill data (same formula as tests/golden/make_golden.py)."""
    x3, y3, z3 = grid.voxel_centres()
    rad = grid.yL / 18.
    rc1 = ((y3 - grid.yL / 1.3 - rad) ** 2) + ((z3 + grid.zLcube / 4 - rad) ** 2)
    rc2 = ((y3 - grid.yL / 4. - rad) ** 2) + ((z3 + grid.zLcube / 4 - rad) ** 2)
    rho = x3 * 0. + 0.1
    rho[rc2 <= rad ** 2] = 1.
    rho[rc1 <= rad ** 2] = 1.
    rho[(x3 < grid.xL / 5.) | (x3 > grid.xL * 4. / 5.)] = 0.1
    rho = rho + 0.02 * (x3 / grid.xL + 2. * y3 / grid.yL - z3 / grid.zLcube)
    return rho, grid.gp_coeff[1] * rho


def synthetic_survey(grid: Grid, md, A=None):
    rho, chi = synthetic_truth(grid)
    loc = grid.sensor_locations()
    if A is None:
        e = grid.edges()
        A = (a_sens(grid, grid.B * 0., loc, e, "grav"), a_sens(grid, grid.B, loc, e, "magn"))
    grav = (A[0] @ rho.flatten()).astype(np.float32).astype(np.float64)
    mag = (A[1] @ chi.flatten()).astype(np.float32).astype(np.float64)
    drill0 = np.zeros_like(rho)
    if md > 0:
        sel = np.random.default_rng(2020).choice(rho.size, md, replace=False)
        drill0.reshape(-1)[sel] = rho.reshape(-1)[sel]
    return dict(gravfield=grav, magfield=mag, sensor_locations=loc, drilldata0=drill0, rho=rho, chi=chi, A=A)
