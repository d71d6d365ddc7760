"""Spectral (real-DFT) form of the covariance x forward-operator product  AK = A_s K_sj  on a regular grid.

SURVEY.md section 8(f) row f2 ("structure-exploiting assembly: same results, fewer flops").  On the grid of
`calcGridPoints3D` (kernels.py:27-42) every block K_sj of `create_cov` (kernels.py:158-195) is a symmetric
three-level Toeplitz matrix: K(p,q) = k(|dy|,|dx|,|dz|).  Embedded in a symmetric three-level circulant of size
(2ny, 2nx, 2nz) it is diagonalised by the Kronecker product of the REAL transforms

    G_a  (P x n, P = 2n):   rows  sqrt(w_o) cos(2 pi o z / P), o = 0..P/2   and   sqrt(2) sin(2 pi o z / P), o = 1..P/2-1

(w_o = 1 for o in {0, P/2}, else 2):   K = crop[ (Gy x Gx x Gz)^T diag(Lambda / PyPxPz) (Gy x Gx x Gz) ],
Lambda(oy,ox,oz) = sum_d k(d) prod_a c_a(d_a) cos(2 pi o_a d_a / P_a)  (c = 1 for d = 0, 2 otherwise) -- all real, because
k is even in every coordinate.  One row of A_s (a ny x nx x nz volume) therefore costs three small-k GEMM passes forward
and three per property block backward (cropping on the way back) instead of an N x N contraction:
2.25e15 -> ~1.2e13 flop at 64^3.  Every pass is a batch of MFMA GEMMs against a fixed cosine/sine matrix
(`geobo_gemm_batched`); no FFT library, no complex arithmetic.

Numerics: the transform matrices are orthogonal up to scaling, so the result differs from the dense path by a few
1e-16 of max|A| max|K| per entry (normwise); the posterior cubes stay far inside the 1e-8 contract (tests).
"""
import math
import os

import numpy as np
import torch

from . import hip

F64 = hip.F64
SLACK = 4096  # doubles of slack behind every buffer: compute tiles may overhang the valid region (reads only)


def base_modes(n):
    """Base rows b = 0 .. n-1 of the real eigenvector basis of symmetric circulants of size P = 2n, as (kind, omega).  Every base row
    g_b comes with its MIRROR row (-1)^i g_b, which is the eigenvector of frequency n - omega (cos) / minus that eigenvector (sin); for
    the middle frequency n/2 the pair is (c + s, c - s) instead of (c, s) -- any rotation inside an eigenspace is as good a basis.
    Spectral position 2b holds base row b, position 2b+1 its mirror: the two rows differ only in the sign of the odd inputs, which is
    what the radix-2 (folded) transform kernels exploit -- per pair ONE even-input and ONE odd-input partial sum, outputs E + O and
    E - O (half the multiply-adds of the plain matrix product).

    ORDER (round 5): base rows come in groups of four, one per frequency omega = 0 .. n/4 - 1,
        b = 4 omega + (0: cos omega, 1: sin omega, 2: cos(n/2 - omega), 3: sin(n/2 - omega)),     omega >= 1
        b = 0 .. 3: cos 0, the middle pair (n/2), cos(n/4), sin(n/4)
    so that the eight spectral positions 8 omega .. 8 omega + 7 hold the frequencies omega, n - omega, n/2 - omega, n/2 + omega: the
    orbit of omega under a shift by a QUARTER period.  On the inputs i = 4j + rho of one residue class the rows of frequency
    n/2 -+ omega are +-(cos | sin) of frequency omega, so a transform needs only the cos / sin rows of omega < n/4 per class
    (RADIX 4: a quarter of the multiply-adds of the plain product; xz2d_fold.hip uses it for the synthesis along z).  The radix-2
    kernels see a pair-interleaved basis as before."""
    assert n % 4 == 0
    q = n // 4
    modes = [("cos", 0), ("mid", n // 2), ("cos", q), ("sin", q)]
    for om in range(1, q):
        modes += [("cos", om), ("sin", om), ("cos", n // 2 - om), ("sin", n // 2 - om)]
    return modes


INTEGER_BASIS_N = (32, 64)      # extents whose fused kernels (xz2d_fold.hip: n = 64, and the four-plane form of 32 x 32 planes) build the
                                # integer-frequency basis of base_modes() internally; every other extent takes the half-integer basis below


def half_integer(n):
    """True where the axis runs on the HALF-INTEGER (skew-circulant) basis (round 6): a symmetric Toeplitz block of size n is also the
    leading block of the skew-circulant of size P = 2n (first column k_0 .. k_{n-1}, *, -k_{n-1} .. -k_1), diagonalised by cos / sin of
    the frequencies kappa = 1/2 .. n - 1/2 with eigenvalues sum_d w_d k(d) cos(2 pi kappa d / P).  No frequency is its own mirror or
    quarter-period image, so all n frequencies fall into n/4 orbits {kappa, n - kappa, n/2 + kappa, n/2 - kappa} of ONE shape -- what
    the radix-4 axis kernels of spectral_y.hip want.  Every consumer of G / G^T / E treats the spectral index as opaque (the eigen-data
    go through the same matrices), and the pair structure row 2b+1 = (-1)^i row 2b of the radix-2 passes holds for every pair."""
    return n not in INTEGER_BASIS_N


def half_modes(n):
    """Base rows of the half-integer basis as (kind, 2 kappa): orbit omega = 0 .. n/4 - 1 (kappa = omega + 1/2) holds the base rows
    4 omega + (0: cos kappa, 1: sin kappa, 2: cos(n/2 + kappa), 3: sin(n/2 + kappa)); their mirrors (-1)^i g_b are the eigenvectors of
    n - kappa and n/2 - kappa: spectral positions 8 omega .. 8 omega + 7 = cos k, cos(n - k), sin k, -sin(n - k), cos(n/2 + k),
    cos(n/2 - k), sin(n/2 + k), -sin(n/2 - k)."""
    assert n % 4 == 0
    modes = []
    for om in range(n // 4):
        k2 = 2 * om + 1
        modes += [("cos", k2), ("sin", k2), ("cos", n + k2), ("sin", n + k2)]
    return modes


def _base_row(kind, om, n):
    P = 2 * n
    z = np.arange(n)
    ang = 2.0 * np.pi * ((om * z) % P) / P                        # exact integer phase index
    if kind == "cos":
        return (1.0 if om == 0 else math.sqrt(2.0)) * np.cos(ang)
    if kind == "sin":
        return math.sqrt(2.0) * np.sin(ang)
    return np.array([1.0, 1.0, -1.0, -1.0])[z % 4]                # cos(pi z / 2) + sin(pi z / 2), exactly


def _half_row(kind, k2, n):
    """sqrt(2) cos / sin(2 pi kappa z / P), kappa = k2 / 2, with the phase reduced in integers (index over 2P)."""
    P = 2 * n
    ang = 2.0 * np.pi * ((k2 * np.arange(n)) % (2 * P)) / (2 * P)
    return math.sqrt(2.0) * (np.cos(ang) if kind == "cos" else np.sin(ang))


def forward_matrix(n):
    """G (P x n): real eigenvector basis of symmetric circulants (integer basis) / skew-circulants (half-integer basis) of size P = 2n,
    restricted to the first n inputs, pair-interleaved: row 2b = base row b (base_modes / half_modes), row 2b+1 = (-1)^z times it."""
    assert n % 4 == 0
    G = np.empty((2 * n, n))
    alt = 1.0 - 2.0 * (np.arange(n) % 2)
    if half_integer(n):
        for b, (kind, k2) in enumerate(half_modes(n)):
            G[2 * b] = _half_row(kind, k2, n)
            G[2 * b + 1] = alt * G[2 * b]
        return G
    for b, (kind, om) in enumerate(base_modes(n)):
        G[2 * b] = _base_row(kind, om, n)
        G[2 * b + 1] = alt * G[2 * b]
    return G


def eigen_matrix(n):
    """E (P x n): Lambda' = E k for a half table k(d), d = 0..n-1, laid out on the same row index as G: row 2b carries the
    eigenvalue of the frequency of base row b, row 2b+1 that of its mirror n - frequency."""
    P = 2 * n
    d = np.arange(n)
    if half_integer(n):
        f2 = np.empty(P, dtype=np.int64)                          # twice the frequency
        for b, (kind, k2) in enumerate(half_modes(n)):
            f2[2 * b], f2[2 * b + 1] = k2, 2 * n - k2
        E = np.cos(2.0 * np.pi * ((f2[:, None] * d[None, :]) % (2 * P)) / (2 * P))
        E[:, 1:] *= 2.0
        return E
    om = np.empty(P, dtype=np.int64)
    for b, (kind, w) in enumerate(base_modes(n)):
        om[2 * b], om[2 * b + 1] = w, n - w
    E = np.cos(2.0 * np.pi * ((om[:, None] * d[None, :]) % P) / P)
    E[:, 1:] *= 2.0
    return E


def folded_matrices(n):
    """(Fe, Fo), each n x n/2: even / odd input columns of the base rows, Fe[b][j] = g_b[2j], Fo[b][j] = g_b[2j+1].
    forward:  out[2b] = E_b + O_b, out[2b+1] = E_b - O_b  with  E_b = sum_j Fe[b][j] x[2j], O_b = sum_j Fo[b][j] x[2j+1];
    inverse:  x[2j] = sum_b Fe[b][j] (s[2b] + s[2b+1]),   x[2j+1] = sum_b Fo[b][j] (s[2b] - s[2b+1])."""
    G = forward_matrix(n)
    return G[0::2, 0::2].copy(), G[0::2, 1::2].copy()


def _pad_rows(M, mult=128):
    r = (M.shape[0] + mult - 1) // mult * mult + mult      # one extra tile of zero rows: row-offset views may overhang
    out = np.zeros((r, M.shape[1]))
    out[:M.shape[0]] = M
    return out


def _buf(n, device):
    return torch.empty(int(n) + SLACK, dtype=F64, device=device)


class LatticeRows:
    """Operator rows of a lattice survey that are never materialised: plane y of row r is a window of the stencil table Q
    (hip.a_sens_lattice_stencil) at row_off[r] + y * q_plane, the two boundary planes come from `edge` ([rows][2][nx*nz]).
    forward_zx feeds the radix-2 forward kernel from this directly (geobo_xz2d_fold_lattice)."""

    def __init__(self, Q, row_off, q_plane, edge, r0=0):
        self.Q, self.row_off, self.q_plane, self.edge, self.r0 = Q, row_off, int(q_plane), edge, int(r0)

    def rows(self, r0):
        return LatticeRows(self.Q, self.row_off, self.q_plane, self.edge, self.r0 + r0)


class SpectralProduct:
    """AK rows by the real-DFT route for one grid; holds the transform matrices and the work buffers."""

    def __init__(self, nx, ny, nz, device, rows_per_batch=None, plane_pad=0, opts=None):
        if nx % 16 or ny % 16 or nz % 16:
            raise ValueError("spectral path needs grid extents that are multiples of 16")
        self.nx, self.ny, self.nz = nx, ny, nz
        # behaviour switches: resolved ONCE by the planner (plan.switches; the engine hands its route's options in); a product built on
        # its own (tests, tools) asks the planner with the process environment
        from . import plan
        self.opts = dict(plan.switches(os.environ)) if opts is None else dict(opts)
        self.Px, self.Py, self.Pz = 2 * nx, 2 * ny, 2 * nz
        self.N = nx * ny * nz
        self.P3 = self.Px * self.Py * self.Pz
        self.device = device
        dev = lambda a: hip.to_dev(a, device)
        self.G = {a: dev(_pad_rows(forward_matrix(n))) for a, n in (("x", nx), ("y", ny), ("z", nz))}
        self.GT = {a: dev(_pad_rows(forward_matrix(n).T.copy())) for a, n in (("x", nx), ("y", ny), ("z", nz))}
        self.E = {a: dev(_pad_rows(eigen_matrix(n))) for a, n in (("x", nx), ("y", ny), ("z", nz))}
        # folded (radix-2) matrices [n][n/2][(Fe, Fo)] for the axes the radix-2 kernels are instantiated for
        self.F = {a: dev(np.stack(folded_matrices(n), axis=2)) for a, n in (("x", nx), ("y", ny), ("z", nz)) if n in hip.XZ2D_FOLD_N}
        self.fold = self.opts["fold"]
        # y axis: applied as Toeplitz blocks per (x, z) mode (geobo_toeplitz_y) when the kernel has the extent, else carried
        # through the spectrum like x and z
        self.dense_y = ny in hip.TOEPLITZ_NY and self.opts["dense_y"]
        # ... and, where instantiated, through the y axis's own spectrum INSIDE the kernel, on the matrix pipe (round 6: geobo_spectral_y,
        # same arguments, same sums to rounding; GEOBO_Y_MFMA=0 keeps the direct vector-pipe kernels: the A/B of profiles/r06_*)
        self.y_mfma = self.dense_y and ny in hip.SPECTRAL_Y_NY and self.opts["y_mfma"]
        # x passes of the unfused (x, z) transforms: radix-4 axis kernels on the half-integer basis (geobo_spectral_axis) where instantiated
        self.x_mfma = self.axis_mfma(nx)
        # x and z: one fused kernel per direction (geobo_xz2d) where it is instantiated, else two batched GEMM passes
        self.fused_xz = (nx, nz) in hip.XZ2D_SHAPES and self.opts["fused_xz"]
        # 32 x 32 planes: two consecutive y-planes stacked along x go through the (64, 32) instance with diag(Mx, Mx) -- the z step
        # is row-wise anyway, the x step spends half its MFMAs on the zero blocks (cheap next to two GEMM passes through HBM)
        self.pair_xz = (nx, nz) == (32, 32) and (64, 32) in hip.XZ2D_SHAPES and ny % 2 == 0 and self.opts["fused_xz"]
        # 32 x 32 planes, better: FOUR consecutive y-planes as one 64 x 64 plane of the radix-2 kernels with the folded matrices of
        # diag(G32, G32) (geobo_xz2d_fold_quad): no zero blocks in z, folded in both axes, four waves per workgroup
        self.quad_xz = ((nx, nz) == (32, 32) and ny % 4 == 0 and 64 in hip.XZ2D_FOLD_N and self.fold
                        and self.opts["quad"] and self.opts["fused_xz"])
        if self.quad_xz:
            fe, fo = folded_matrices(32)
            fq = np.zeros((64, 32, 2))
            fq[:32, :16, 0], fq[:32, :16, 1] = fe, fo
            fq[32:, 16:, 0], fq[32:, 16:, 1] = fe, fo
            self.Fq = dev(fq)
        self._pairs = {}
        # plane stride of the (x, z)-spectrum work buffers [row][y][Px*Pz].  Px*Pz is a power of two at 64^3 (16384 doubles =
        # 128 KiB); a bare copy with the y stage's access pattern (512-byte runs, one per plane) streams 3.95 TB/s at that stride and
        # 5.2 TB/s with 2 KiB of padding (tools/hbm_copy_runs.hip, profiles/r03_hbm_copy_runs.txt) -- but the kernels of this
        # pipeline do not move: toeplitz_y 1.514 / 1.506 ms, inverse transform 0.372 / 0.375 ms, forward 0.571 / 0.573 ms with / without
        # the padding (profiles/r03_spectral_plane_pad_ab.txt): they are held by the fp64 VALU / the matrix pipe at the clock a
        # memory-bound launch runs at, not by channel aliasing.  The stride stays a parameter (`plane_pad`, doubles; the A/B switch
        # of round 3 is retired); default dense.  Fused kernels only: they take explicit row / plane strides.
        C = self.Px * self.Pz
        pad = int(plane_pad)
        self.Cp = C + pad if (self.fused_xz and self.dense_y and C % 2048 == 0 and pad > 0) else C
        # operator rows fed straight from a lattice survey's stencil table (LatticeRows): needs the radix-2 forward kernel
        self.lattice_feed = self.fused_xz and self.fold and nx == nz and "x" in self.F and ny >= 3 and self.dense_y
        if rows_per_batch is None:
            per_row = (ny * self.Cp if self.dense_y else self.P3) * 8
            # ~3 GB per work buffer; ny = 128 (67 MB of spectrum per row): 48 rows.  Measured on the 128^3 rank step with the round-4 y
            # stage (several output chunks per staged row), A K -> AkA + posterior: 16 rows 1.69 + 2.94 s, 24: 1.62 + 2.92, 32: 1.58 + 2.82,
            # 48: 1.55 + 2.78, 64: 1.58 + 2.77, 96: 1.53 + 2.76, 128: 1.50 + 2.75 (flat from 48; 108 / 117 / 127 / 146 GB peak)
            cap = 3 << 30
            rows_per_batch = max(1, min(256 if self.dense_y else 128, cap // per_row))
        g = 128 // math.gcd(nx * ny, 128)
        self.R = max(g, rows_per_batch // g * g)
        self._bufs = {}
        self.kernel_timer = None     # callable(name, algorithmic_bytes, fn) -> fn(): the engine's HIP-event bracket for single kernels

    def axis_mfma(self, n):
        """True where a pass along a strided axis of extent n runs as a radix-4 axis kernel (geobo_spectral_axis)."""
        return n in hip.SPECTRAL_AXIS_N and half_integer(n) and self.opts.get("axis_mfma", True)

    def _ystage(self, ny, C, R, src, tabs, outs, y0, y1, plane, accumulate=False):
        """geobo_toeplitz_y launch, bracketed for the bench's per-kernel roofline when a timer is set: algorithmic bytes = the rows'
        (x, z)-spectrum read once + one output slab per property block (read as well when the launch accumulates)."""
        # (the long-axis matrix-pipe kernel computes every output of a row whatever the slab; the windowed direct kernel only the chunks of
        # 16 that cover it: narrow slabs -- the y-slab shards of the column form -- stay with the direct kernel: 0.55 against 1.33 ms for 16
        # of 128 planes)
        if self.y_mfma and (ny > 64 or not accumulate) and (ny <= 64 or 2 * (y1 - y0) >= ny):
            fn = lambda: hip.spectral_y(ny, C, R, src, tabs, outs, y0, y1, plane=plane, accumulate=accumulate)
        else:
            fn = lambda: hip.toeplitz_y(ny, C, R, src, tabs, outs, y0, y1, plane=plane, accumulate=accumulate)
        if self.kernel_timer is None:
            return fn()
        # (one name per kernel symbol: single-block launches run toeplitz_y_kernel<ny, 2>, the others <ny, 1>)
        name = "kernel:toeplitz_y" if len(tabs) >= 2 or ny > 64 else "kernel:toeplitz_y_single"
        fl, fv = self.y_stage_flop(1, len(tabs), y1 - y0)
        return self.kernel_timer(name, 8.0 * R * C * (ny + (2 if accumulate else 1) * len(tabs) * (y1 - y0)), fn, valu=R * fv, flop=R * fl)

    def buf(self, name, n):
        b = self._bufs.get(name)
        if b is None or b.numel() < n + SLACK:
            b = self._bufs[name] = _buf(n, self.device)
        return b

    def _paired(self, Mx, rows, cols):
        """diag(Mx, Mx) for the stacked-pair form of the fused kernel (Mx: the valid rows x cols of a padded matrix)."""
        key = (Mx.data_ptr(), rows, cols)
        m2 = self._pairs.get(key)
        if m2 is None:
            m2 = torch.zeros((2 * rows, 2 * cols), dtype=Mx.dtype, device=Mx.device)
            m2[:rows, :cols] = Mx[:rows, :cols]
            m2[rows:, cols:] = Mx[:rows, :cols]
            self._pairs[key] = m2
        return m2

    # ---- axis passes, z (contiguous) then x [then y]: [R][ny][nx][nz] -> [R][ny][Px][Pz] [-> [R][Py][Px][Pz]] -----------------
    def forward_zx(self, src, R, M, src_row_stride=None, out_name="T2"):
        """src: R volumes of ny*nx*nz doubles, `src_row_stride` doubles apart (default: contiguous)."""
        nx, ny, nz, Px, Pz, Cp = self.nx, self.ny, self.nz, self.Px, self.Pz, self.Cp
        rows = R * ny * nx
        lds = self.N if src_row_stride is None else int(src_row_stride)
        if isinstance(src, LatticeRows):
            assert self.lattice_feed and M is self.G
            t2 = self.buf(out_name, R * ny * Cp)
            hip.xz2d_fold_lattice(nx, R, ny, src.Q, src.row_off[src.r0:], src.q_plane, src.edge[src.r0:], src.edge.stride(0),
                                  self.F["x"], self.F["z"], t2, ny * Cp, Cp)
            return t2
        if self.fused_xz:
            t2 = self.buf(out_name, R * ny * Cp)       # planes Cp >= Px*Pz doubles apart (padded: see __init__)
            if self.fold and M is self.G and nx == nz and "x" in self.F:     # radix-2 kernels: half the MFMAs (xz2d_fold.hip)
                hip.xz2d_fold(False, nx, R, ny, src, lds, nx * nz, self.F["x"], self.F["z"], t2, ny * Cp, Cp)
            else:
                hip.xz2d(False, nx, nz, R, ny, src, lds, nx * nz, M["x"], M["z"], t2, ny * Cp, Cp)
            return t2
        if self.quad_xz and M is self.G:
            t2 = self.buf(out_name, R * ny * Px * Pz)
            hip.xz2d_fold_quad(False, nx, R, ny // 4, src, lds, nx * nz, self.Fq, self.Fq, t2, ny * Px * Pz, Px * Pz)
            return t2
        if self.pair_xz:
            t2 = self.buf(out_name, R * ny * Px * Pz)
            hip.xz2d(False, 2 * nx, nz, R, ny // 2, src, lds, 2 * nx * nz, self._paired(M["x"], Px, nx), M["z"], t2, ny * Px * Pz,
                     2 * Px * Pz)
            return t2
        # any other extent: two batched passes, radix-2 (geobo_gemm_fold: half the multiply-adds) against the pair-interleaved basis
        fold = self.fold and M is self.G
        t1 = self.buf("T1", rows * Pz)
        hip.axis_pass(fold, False, False, hip.pad_n(ny * nx), hip.pad_n(Pz), nz, src, nz, lds, M["z"], nz, 0, t1, Pz, ny * nx * Pz, ny * nx, Pz, R)
        t2 = self.buf(out_name, R * ny * Px * Pz)
        if self.x_mfma and M is self.G and Pz % 16 == 0:
            hip.spectral_axis(False, nx, Pz, Pz, Pz, nx * Pz, Px * Pz, R * ny, t1, t2)
        else:
            hip.axis_pass(fold, True, False, hip.pad_n(Px), hip.pad_n(Pz), nx, M["x"], nx, 0, t1, Pz, nx * Pz, t2, Pz, Px * Pz, Px, Pz, R * ny)
        return t2

    def forward(self, src, R, M, out_name="T3", src_row_stride=None):
        ny, Px, Py, Pz = self.ny, self.Px, self.Py, self.Pz
        t2 = self.forward_zx(src, R, M, src_row_stride)
        t3 = self.buf(out_name, R * self.P3)
        hip.gemm_batched(True, hip.pad_n(Py), hip.pad_n(Px * Pz), ny, M["y"], ny, 0, t2, Px * Pz, ny * Px * Pz, t3, Px * Pz,
                         self.P3, Py, Px * Pz, R)
        return t3

    # ---- back: y (crop to the slab [y0,y1)), x, z; writes rows of `out` (leading dimension ldo) -----------------------------
    def backward(self, spec, R, y0, y1, out, ldo):
        Px, Py, Pz = self.Px, self.Py, self.Pz
        slab = y1 - y0
        gyt = self.GT["y"][y0:]                                  # rows y0.. of G_y^T (padded rows behind are zero / slack)
        u2 = self.buf("U2", R * slab * Px * Pz)
        hip.gemm_batched(True, hip.pad_n(slab), hip.pad_n(Px * Pz), Py, gyt, Py, 0, spec, Px * Pz, self.P3, u2, Px * Pz,
                         slab * Px * Pz, slab, Px * Pz, R)
        self.backward_xz(u2, R, y0, y1, [(y0, y1, out, ldo)])

    def backward_xz(self, u2, R, ylo, yhi, targets):
        """u2: [R][yhi-ylo][Px][Pz] (y already in the space domain).  x pass over every (row, y), then one z pass per target
        (ya, yb, out, ldo): rows of `out` receive the y-slab [ya, yb) of every row."""
        nx, nz, Px, Pz = self.nx, self.nz, self.Px, self.Pz
        Ly = yhi - ylo
        if self.fused_xz:
            Cp = self.Cp if self.dense_y else Px * Pz       # (y through the spectrum: the batched GEMM of backward() writes dense planes)
            for ya, yb, out, ldo in targets:
                if self.fold and nx == nz and "x" in self.F:
                    hip.xz2d_fold(True, nx, R, yb - ya, u2[(ya - ylo) * Cp:], Ly * Cp, Cp, self.F["x"], self.F["z"],
                                  out, ldo, nx * nz)
                else:
                    hip.xz2d(True, nx, nz, R, yb - ya, u2[(ya - ylo) * Cp:], Ly * Cp, Cp, self.GT["x"], self.GT["z"],
                             out, ldo, nx * nz)
            return
        if self.quad_xz and all((yb - ya) % 4 == 0 for ya, yb, _, _ in targets):
            for ya, yb, out, ldo in targets:
                hip.xz2d_fold_quad(True, nx, R, (yb - ya) // 4, u2[(ya - ylo) * Px * Pz:], Ly * Px * Pz, Px * Pz, self.Fq, self.Fq,
                                   out, ldo, nx * nz)
            return
        if self.pair_xz and all((yb - ya) % 2 == 0 for ya, yb, _, _ in targets):
            for ya, yb, out, ldo in targets:
                hip.xz2d(True, 2 * nx, nz, R, (yb - ya) // 2, u2[(ya - ylo) * Px * Pz:], Ly * Px * Pz, 2 * Px * Pz,
                         self._paired(self.GT["x"], nx, Px), self.GT["z"], out, ldo, 2 * nx * nz)
            return
        u1 = self.buf("U1", R * Ly * nx * Pz)
        if self.x_mfma and Pz % 16 == 0:
            hip.spectral_axis(True, nx, Pz, Pz, Pz, Px * Pz, nx * Pz, R * Ly, u2, u1)
        else:
            hip.axis_pass(self.fold, True, True, hip.pad_n(nx), hip.pad_n(Pz), Px, self.GT["x"], Px, 0, u2, Pz, Px * Pz, u1, Pz, nx * Pz, nx, Pz,
                          R * Ly)
        for ya, yb, out, ldo in targets:
            slab = yb - ya
            hip.axis_pass(self.fold, False, True, hip.pad_n(slab * nx), hip.pad_n(nz), Pz, u1[(ya - ylo) * nx * Pz:], Pz, Ly * nx * Pz,
                          self.GT["z"], Pz, 0, out, nz, ldo, slab * nx, nz, R)

    def flops(self, rows, nblocks, slab):
        """Executed flop of product(): forward passes once, the rest per property block (compute extents)."""
        nx, ny, nz, Px, Py, Pz = self.nx, self.ny, self.nz, self.Px, self.Py, self.Pz
        pn = hip.pad_n
        fwd = 2.0 * (ny * nx * pn(Pz) * nz + ny * pn(Px) * pn(Pz) * nx)
        bwd = 2.0 * (slab * pn(nx) * pn(Pz) * Px + pn(slab * nx) * pn(nz) * Pz)
        if self.fold and not (self.fused_xz or self.quad_xz or self.pair_xz):
            fwd, bwd = 0.5 * fwd, 0.5 * bwd         # radix-2 batched passes (geobo_gemm_fold)
        if self.quad_xz:                            # four planes per 64-point radix-2 plane: folded, both axes over the zero blocks
            fwd = 0.25 * ny * (64.0 * 64 * 128 + 128.0 * 64 * 128)
            bwd = 0.25 * slab * (128.0 * 128 * 64 + 64.0 * 128 * 64)
        elif self.pair_xz:                          # diag(Mx, Mx) on stacked plane pairs: the x steps run over the zero blocks too
            fwd += 2.0 * ny * Px * Pz * nx
            bwd += 2.0 * slab * nx * Pz * Px
        if self.fused_xz and self.fold and "x" in self.F and nx == nz:
            # radix-2 / radix-4 kernels (xz2d_fold.hip): forward z step one even- and one odd-input sum per spectral pair (half the plain
            # multiply-adds), forward x step and both inverse steps one cos / sin row per frequency group and residue class (a quarter)
            fwd = 2.0 * ny * (0.5 * nx * pn(Pz) * nz + 0.25 * pn(Px) * pn(Pz) * nx)
            bwd = 0.25 * bwd
        if not self.dense_y:
            fwd += 2.0 * pn(Py) * Px * Pz * ny
            bwd += 2.0 * pn(slab) * Px * Pz * Py
        return rows * (fwd + nblocks * bwd + (self.y_stage_flop(1, nblocks, slab)[0] if self.dense_y else 0.0))

    def y_stage_flop(self, terms, nblocks, slab=None, shared=False):
        """(executed flop, its vector-pipe part) of the y stage per row: `terms` input spectra, `nblocks` output blocks.
        Direct kernels (toeplitz.hip): ny multiply-adds per output, block and term on the vector pipe (ny <= 64: every output y is
        computed and the slab stored; ny > 64: chunks of 16 outputs covering the slab); `shared`: the two-term rows of a block pair
        whose cross blocks coincide cost three products instead of four.
        In-kernel spectral product (spectral_y.hip): ny^2 / 2 multiply-adds per transform and mode on the matrix pipe -- one analysis
        per term, one synthesis per block -- plus the orbit butterflies on the vector pipe: 16 additions per orbit and analysis,
        4 mul + 8 fma + 8 add per orbit and block (one term) or 8 add + 8 mul + 16 fma + 32 add per orbit for a block pair of two-term
        rows (ny / 4 orbits per mode)."""
        ny, C = self.ny, self.Px * self.Pz
        if self.y_mfma and (ny <= 64 or slab is None or 2 * slab >= ny):
            valu = C * ny * (4.0 * terms + (7.0 * nblocks if terms == 1 else 10.0 * nblocks))
            return C * 1.0 * ny * ny * (terms + nblocks) + valu, valu
        outs = ny if (ny <= 64 or slab is None) else (slab + 15) // 16 * 16
        nprod = terms * nblocks * (0.75 if shared and terms == 2 and nblocks == 2 else 1.0)
        f = 2.0 * ny * outs * C * nprod
        return f, f

    def flops_valu(self, rows, nblocks, slab=None):
        """The part of flops() executed on the fp64 VALU (the y stage's vector-pipe work); everything else is MFMA."""
        return rows * self.y_stage_flop(1, nblocks, slab)[1] if self.dense_y else 0.0

    def eigenvalues(self, table_mirrored):
        """What product() needs of one covariance block, from the (z-mirrored) lattice table of geobo_cov_table:
        dense-y route: the Toeplitz generators t[d][ox][oz] = (E_x E_z k)(d, ox, oz) / (Px Pz)   ([ny][Px*Pz]);
        otherwise the full eigenvalue cube Lambda'/(Py Px Pz) on the G row index."""
        nx, ny, nz = self.nx, self.ny, self.nz
        T = table_mirrored[:ny * nx * 2 * nz].view(ny, nx, 2 * nz)[:, :, nz - 1:2 * nz - 1].contiguous()
        src = self.buf("Tsrc", self.N)
        src[:self.N] = T.reshape(-1)
        if self.dense_y:
            C = self.Px * self.Pz
            Cp = self.Cp if self.fused_xz else C          # plane stride of what forward_zx wrote
            gen = self.forward_zx(src, 1, self.E, out_name="Lam")[:ny * Cp].view(ny, Cp)[:, :C].clone().view(-1)   # own dense [ny][C] table (the work buffer is reused)
            gen.mul_(1.0 / float(C))
            return gen
        lam = self.forward(src, 1, self.E, out_name="Lam")[:self.P3].clone()
        lam.mul_(1.0 / float(self.P3))
        return lam

    def product(self, A, Ms, lam_list, outs, y0=0, y1=None, slabs=None):
        """outs[j][r, :] = (A[r, :N] convolved with block j) restricted to y-slab [y0,y1), r < Ms.
        A: (>=Ms x N) row-major; outs[j]: 2-D views (rows x (y1-y0)*nx*nz).
        `slabs` = [(y0, y1, outs), ...] crops the SAME scaled spectrum to several y-slabs (one per destination rank of the
        row-sharded multi-GPU form); the forward passes and the scaling are done once."""
        y1 = self.ny if y1 is None else y1
        if slabs is None:
            slabs = [(y0, y1, outs)]
        N = self.N
        if isinstance(A, LatticeRows):
            assert self.lattice_feed
        else:
            assert A.stride(1) == 1 and A.stride(0) >= N and A.stride(0) % 2 == 0
        if self.dense_y:
            return self._product_dense_y(A, Ms, lam_list, slabs)
        for r0 in range(0, Ms, self.R):
            R = min(self.R, Ms - r0)
            spec = self.forward(A[r0:], R, self.G, src_row_stride=A.stride(0))
            n = R * self.P3
            j = 0
            while j < len(lam_list):
                if j + 1 < len(lam_list):      # two property blocks per read of the spectrum
                    s0, s1 = self.buf("S", n), self.buf("S1", n)
                    hip.scale_broadcast2(spec[:n], lam_list[j], lam_list[j + 1], s0[:n], s1[:n])
                    for ya, yb, o in slabs:
                        self.backward(s0, R, ya, yb, o[j][r0:], o[j].stride(0))
                        self.backward(s1, R, ya, yb, o[j + 1][r0:], o[j + 1].stride(0))
                    j += 2
                else:
                    s0 = self.buf("S", n)
                    hip.scale_broadcast(spec[:n], lam_list[j], s0[:n])
                    for ya, yb, o in slabs:
                        self.backward(s0, R, ya, yb, o[j][r0:], o[j].stride(0))
                    j += 1

    def _product_dense_y(self, A, Ms, gens, slabs):
        """z and x through the spectrum, y as Toeplitz blocks: one read of the (x, z)-spectrum per pair of property blocks."""
        ny, C = self.ny, self.Px * self.Pz
        Cp = self.Cp if self.fused_xz else C
        ylo, yhi = min(s[0] for s in slabs), max(s[1] for s in slabs)
        n_out = (yhi - ylo) * Cp
        for r0 in range(0, Ms, self.R):
            R = min(self.R, Ms - r0)
            if isinstance(A, LatticeRows):
                t2 = self.forward_zx(A.rows(r0), R, self.G)
            else:
                t2 = self.forward_zx(A[r0:], R, self.G, src_row_stride=A.stride(0))
            for j in range(0, len(gens), 3):              # up to three property blocks per read of the (x, z)-spectrum
                js = list(range(j, min(j + 3, len(gens))))
                u2 = [self.buf(("S", "S1", "S2")[i], R * n_out) for i in range(len(js))]
                self._ystage(ny, C, R, t2, [gens[jj] for jj in js], u2, ylo, yhi, Cp)
                for i, jj in enumerate(js):
                    self.backward_xz(u2[i], R, ylo, yhi, [(ya, yb, o[jj][r0:], o[jj].stride(0)) for ya, yb, o in slabs])

    # ---- posterior variance in the transposed order (engine._posterior_zpath) ------------------------------------------------------
    def fused_ss(self):
        """True where the inverse transform itself squares and sums its output planes (geobo_xz2d_fold_inv_ss: the radix-2 kernels of
        nx = nz = 64 with the Toeplitz y stage); elsewhere reduce_ss stores a batch of V rows and geobo_sumsq_accum reduces it."""
        return self.fused_xz and self.fold and self.nx == self.nz and "x" in self.F and self.dense_y

    GENERIC_SS_SLOTS = 8

    def ss_slots(self):
        """Partial cubes per property block that reduce_ss adds to (sum over them afterwards)."""
        return hip.xz2d_fold_inv_ss_slots(self.nx, self.R, self.ny) if self.fused_ss() else self.GENERIC_SS_SLOTS

    def y2s_tables(self, gens_g, gens_m, pair=(0, 1)):
        """Tables of the three-product form of the two-term rows for the block pair (a, b) = `pair` (b = a + 1, a even), or None where
        the kernel does not apply.  The caller states that gens_g[b] and gens_m[a] are the SAME covariance block -- K_01 = K_10:
        create_cov's prior is symmetric (kernels.py:181-195) -- and the statement is verified on the device: `sym_residual`
        (max |gens_g[b] - gens_m[a]| over max |gens_g[b]|, a 0-d device tensor) is left for the caller to read at its next
        synchronisation (engine.posterior raises above 1e-12).  Returns (a, D0 = gens_g[a] - X, X = gens_g[b], D1 = gens_m[b] - X)."""
        self.sym_residual = None
        a, b = pair
        if not (self.fused_ss() and self.ny in hip.TOEPLITZ_Y2T_NY and b == a + 1 and a % 2 == 0 and b < len(gens_g)
                and self.opts["y2s"]):
            return None
        x = gens_g[b]
        # (cross weight 0 -- gp_coeff = [.., .., 0], a legal prior -- makes both cross blocks exactly zero: 0 / 0 must not read as
        # "asymmetric", so the residual is the difference against a denominator that is never zero)
        den = torch.maximum(x.abs().max(), gens_m[a].abs().max())
        self.sym_residual = (x - gens_m[a]).abs().max() / den.clamp_min(torch.finfo(x.dtype).tiny)
        return (a, gens_g[a] - x, x, gens_m[b] - x)

    def reduce_ss(self, Zg, Mg, gens_g, Zm, m_first, gens_m, ss, y2s=None):
        """ss[j][slot][y][x*z] += sum_m V_j[m]^2  with  V_j[m] = Zg[m] * K_0j  (+ Zm[m - m_first] * K_1j for m >= m_first),  m < Mg:
        rows of L^-1 A_s carried through the covariance product and squared on the way out of the inverse transform -- V is never
        written.  Zg: (>= Mg x N) rows, Zm: (>= Mg - m_first x N) rows or None; gens_g[j] / gens_m[j]: Toeplitz generators of the
        blocks (0, j) / (1, j) (eigenvalues()); ss[j]: zeroed partial cubes (ss_slots() slots each).
        Batches never straddle m_first (a batch is entirely one-term or entirely two-term).  Grids without the fused reduction
        (fused_ss): the storing product of every batch into a scratch of R rows, then geobo_sumsq_accum.
        y2s: tables of y2s_tables() for a pair of blocks whose cross blocks coincide -- the two-term rows of that pair then cost three
        y-stage products instead of four (geobo_toeplitz_y2s)."""
        nx, ny, nz, C, Cp, R = self.nx, self.ny, self.nz, self.Px * self.Pz, self.Cp, self.R
        P_c = len(gens_g)
        y2s_tabs = y2s
        if Zm is None:
            m_first = Mg
        starts = list(range(0, min(m_first, Mg), R)) + list(range(m_first, Mg, R))
        if not self.fused_ss():
            N = self.N
            add_y = self.dense_y and ny in hip.TOEPLITZ_ADD_NY
            for r0 in starts:
                two = r0 >= m_first
                Rb = min(R, (Mg if two else min(m_first, Mg)) - r0)
                vg = [self.buf("SSV%d" % jj, R * N)[:R * N].view(R, N) for jj in range(P_c)]
                if two and add_y:
                    # both terms meet in the (x, z)-spectrum: the second y stage adds into the first one's output, ONE inverse
                    # transform per block instead of two (what geobo_toeplitz_y2t does for the register-table kernel)
                    Cq = Cp if self.fused_xz else C
                    t2g = self.forward_zx(Zg[r0:], Rb, self.G, src_row_stride=Zg.stride(0), out_name="T2")
                    t2m = self.forward_zx(Zm[r0 - m_first:], Rb, self.G, src_row_stride=Zm.stride(0), out_name="T2b")
                    for j in range(0, P_c, 3):
                        js = list(range(j, min(j + 3, P_c)))
                        u2 = [self.buf(("S", "S1", "S2")[i], Rb * ny * Cq) for i in range(len(js))]
                        if self.y_mfma and ny in hip.SPECTRAL_Y3T_NY:
                            # both terms in ONE pass, meeting in the y spectrum (geobo_spectral_y3t): 2 + len(js) transforms instead of
                            # 2 (1 + len(js)), no read-modify-write of the outputs
                            fn = lambda: hip.spectral_y3t(ny, C, Rb, t2g, t2m, [gens_g[jj] for jj in js], [gens_m[jj] for jj in js], u2, plane=Cq)
                            if self.kernel_timer is None:
                                fn()
                            else:
                                fl, fv = self.y_stage_flop(2, len(js))
                                self.kernel_timer("kernel:toeplitz_y", 8.0 * Rb * C * ny * (2 + len(js)), fn, valu=Rb * fv, flop=Rb * fl)
                        else:
                            self._ystage(ny, C, Rb, t2g, [gens_g[jj] for jj in js], u2, 0, ny, Cq)
                            self._ystage(ny, C, Rb, t2m, [gens_m[jj] for jj in js], u2, 0, ny, Cq, accumulate=True)
                        for i, jj in enumerate(js):
                            self.backward_xz(u2[i], Rb, 0, ny, [(0, ny, vg[jj], vg[jj].stride(0))])
                    for jj in range(P_c):
                        hip.sumsq_accum(vg[jj], None, Rb, ss[jj].view(ss[jj].shape[0], -1))
                    continue
                self.product(Zg[r0:], Rb, gens_g, vg)
                vm = None
                if two:
                    vm = [self.buf("SSW%d" % jj, R * N)[:R * N].view(R, N) for jj in range(P_c)]
                    self.product(Zm[r0 - m_first:], Rb, gens_m, vm)
                for jj in range(P_c):
                    hip.sumsq_accum(vg[jj], vm[jj] if two else None, Rb, ss[jj].view(ss[jj].shape[0], -1))
            return
        for r0 in starts:
            two = r0 >= m_first
            Rb = min(R, (Mg if two else min(m_first, Mg)) - r0)
            t2g = self.forward_zx(Zg[r0:], Rb, self.G, src_row_stride=Zg.stride(0), out_name="T2")
            t2m = self.forward_zx(Zm[r0 - m_first:], Rb, self.G, src_row_stride=Zm.stride(0), out_name="T2b") if two else None
            for j in range(0, P_c, 2):
                js = list(range(j, min(j + 2, P_c)))
                sg = [self.buf(("S", "S1")[i], Rb * ny * Cp) for i in range(len(js))]
                if two and len(js) == 2 and ny in hip.TOEPLITZ_Y2T_NY:
                    # both terms in ONE y-stage pass: one output spectrum per block, one input of the inverse; with the shared cross
                    # block three products per mode (geobo_toeplitz_y2s), otherwise four (geobo_toeplitz_y2t)
                    if y2s_tabs is not None and y2s_tabs[0] == j:
                        y2s_fn = hip.spectral_y2s if self.y_mfma else hip.toeplitz_y2s
                        fn = lambda: y2s_fn(ny, C, Rb, t2g, t2m, y2s_tabs[1], y2s_tabs[2], y2s_tabs[3], sg, plane=Cp)
                        kname, nprod = "kernel:toeplitz_y2s", 3
                    else:
                        fn = lambda: hip.toeplitz_y2t(ny, C, Rb, t2g, t2m, [gens_g[jj] for jj in js], [gens_m[jj] for jj in js], sg, plane=Cp)
                        kname, nprod = "kernel:toeplitz_y2t", 4
                    if self.kernel_timer is None:
                        fn()
                    else:
                        self.kernel_timer(kname, 8.0 * Rb * C * (2 * ny + 2 * ny), fn, valu=Rb * self.y_stage_flop(2, 2, shared=nprod == 3)[1],
                                          flop=Rb * self.y_stage_flop(2, 2, shared=nprod == 3)[0])
                    for i, jj in enumerate(js):
                        hip.xz2d_fold_inv_ss(nx, Rb, ny, sg[i], ny * Cp, Cp, self.F["x"], self.F["z"], ss[jj])
                    continue
                self._ystage(ny, C, Rb, t2g, [gens_g[jj] for jj in js], sg, 0, ny, Cp)
                sm = None
                if two and ny in hip.TOEPLITZ_ADD_NY:
                    self._ystage(ny, C, Rb, t2m, [gens_m[jj] for jj in js], sg, 0, ny, Cp, accumulate=True)     # the terms meet in the spectrum
                elif two:
                    sm = [self.buf(("Sb", "S1b")[i], Rb * ny * Cp) for i in range(len(js))]
                    self._ystage(ny, C, Rb, t2m, [gens_m[jj] for jj in js], sm, 0, ny, Cp)
                for i, jj in enumerate(js):
                    hip.xz2d_fold_inv_ss(nx, Rb, ny, sg[i], ny * Cp, Cp, self.F["x"], self.F["z"], ss[jj],
                                         src2=sm[i] if sm is not None else None, in2_row=ny * Cp, r2_first=0)

    def flops_ss(self, rows_one, rows_two, nblocks, shared=False):
        """Executed flop of reduce_ss: rows_one one-term rows, rows_two two-term rows (forward transform per term, y stage, one second
        inverse step per row; the first inverse step per term)."""
        nx, ny, nz, Px, Pz = self.nx, self.ny, self.nz, self.Px, self.Pz
        fwd = 1.0 * ny * (nx * nz * Pz + 0.5 * Px * nx * Pz)           # radix 2 along z (half the plain products), radix 4 along x (a quarter)
        inv1, inv2 = 0.5 * ny * Px * Pz * nz, 0.5 * ny * nx * Px * nz  # radix 4 both ways
        terms = rows_one + 2 * rows_two
        y = rows_one * self.y_stage_flop(1, nblocks)[0] + rows_two * self.y_stage_flop(2, nblocks, shared=shared)[0]
        return terms * (fwd + nblocks * inv1) + (rows_one + rows_two) * nblocks * inv2 + y

    def valu_ss(self, rows_one, rows_two, nblocks, shared=False):
        """The vector-pipe part of flops_ss (the y stage's share on the fp64 VALU)."""
        return rows_one * self.y_stage_flop(1, nblocks)[1] + rows_two * self.y_stage_flop(2, nblocks, shared=shared)[1]
