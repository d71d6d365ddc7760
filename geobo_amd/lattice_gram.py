"""AkA = (A K) A^T on a lattice survey without the N-deep GEMM  ("structure-exploiting assembly").

With the sensors on the cube's own x-y lattice (hip.lattice_plan) the operator entry for sensor (jy', jx') and voxel
(iy, ix, iz) of an interior y-slab is a stencil table  Q[iy - jy'][ix - jx'][iz]  (geobo_a_sens_lattice).  A block column of
AkA is then, row by row, a two-level correlation with the z axis as a channel:

    AkA[r, (jy', jx')] = sum_{iy in 1..ny-2} sum_ix sum_iz  X_r[iy, ix, iz] * Q[iy - jy', ix - jx', iz]   +   boundary slabs,

X_r = row r of A K (the same product that the N-deep GEMM contracts).  Q is even in both offsets for the vertical-component
operators (checked on the device; otherwise the GEMM is used), so the (y, x) circulant embedding is diagonalised by the same real
transforms G as the covariance (spectral.py), channel by channel:

    AkA[r, :] = crop (Gy x Gx)^T [ sum_iz Lambda[:, :, iz] * ((Gy x Gx) X_r[:, :, iz]) ]

  1. y step     Y1_r = Gy0 X_r            batched MFMA GEMM (geobo_gemm_batched); Gy0 = Gy with the columns of the two
                                           boundary slabs zeroed
  2. x step + scaling + channel sum        geobo_xcorr_reduce (fused MFMA kernel, one (row, y-mode) plane per workgroup step)
  3. back       C_r = Gy^T S_r Gx          geobo_xz2d (inverse), written straight into the AkA row segment
  4. the two 1e6-padded boundary slabs     (3 % of the contraction, 20 % of the time as plain GEMMs over their 2 nx nz columns)
     The padding shifts node x AND y offsets of the plane by 1e6 m (sensormodel.py:63-68): the slab's stencil is no longer even
     in x -- no real-cosine diagonalisation -- and differs from sensor row to sensor row in y, but it is still a function of
     (jy, ix - jx, iz): for every jy a one-level Toeplitz matrix in x with the z axis as a channel.  Round 3: cross-correlation
     along x through the full real DFT (cosine AND sine parts, P = 2 nx points, nothing assumed about symmetry),
         out[r, jy, jx] = 1/P sum_w c_w [ C_w cos(2 pi w jx / P) + S_w sin(2 pi w jx / P) ],
         C_w = sum_iz (Xc Kc + Xs Ks),   S_w = sum_iz (Xs Kc - Xc Ks)
     as three batched MFMA GEMMs (edge_rows): x-DFT of the slab planes, per-frequency contraction over (cos / sin, iz) against
     the stencil's spectrum, inverse DFT; 4.3e6 instead of 3.4e7 flop per row and slab.

2 x 1.9e8 flop per row instead of 2.1e9, and no N-deep pass over A."""
import numpy as np

import torch

from . import hip

F64 = hip.F64


class LatticeGram:
    def __init__(self, sp, device):
        """sp: SpectralProduct of the same grid (transform and eigen matrices)."""
        self.sp, self.device = sp, device
        self.nx, self.ny, self.nz = sp.nx, sp.ny, sp.nz
        self.Px, self.Py = sp.Px, sp.Py
        gy0 = sp.G["y"].clone()                       # (P x n, padded rows): boundary slabs iy = 0, ny-1 do not enter
        gy0[:, 0] = 0.0
        gy0[:, self.ny - 1] = 0.0
        self.Gy0 = gy0
        # rows per batch: 256 where the fused kernels run (1 GB of y-step output at 64^3); the shape-independent x step keeps
        # D = Gx X for a whole batch (Py Px nz doubles per row: 67 MB at 128^3), ~2 GB per buffer
        per_row = self.Py * self.Px * self.nz * 8
        dflt = 256 if self.fast(self.nx, self.ny, self.nz) else max(8, min(256, (2 << 30) // per_row))
        self.R = dflt
        # rows per batch of the transposed application (W = Lambda * lhat is Px nz Py doubles per row: 8.4 MB at 64^3, 67 MB at 128^3)
        self.Rz = self.R if self.fast(self.nx, self.ny, self.nz) else max(8, min(256, (2 << 30) // per_row))
        self.J = (self.ny + 63) // 64 * 64                          # jy slots of the boundary-slab spectra (whole 64-row halves of a GEMM tile)
        self._lam_oz = {}

    # ---- boundary slabs: x-correlation through the full real DFT (module docstring, item 4) -----------------------------------
    EDGE_ROWS = 1024     # rows per batch: 64 frequency slots x 8 column tiles = 512 tiles per batched GEMM (two per CU)

    def _edge_consts(self):
        """F2 [2 nx comps][nx]: slot 0 = (cos_0, cos_nx) -- the two frequencies without a sine part --, slot w = (cos_w, sin_w);
        FiT [nx (padded to 128 rows)][2 nx]: the inverse with the weights c_w / P folded in."""
        if getattr(self, "_edge_c", None) is None:
            nx = self.nx
            P = 2 * nx
            i = np.arange(nx)
            F2 = np.zeros((hip.pad_n(2 * nx) // 2, 2, nx))           # (zero rows behind 2 nx: compute extents are multiples of 128)
            F2[0, 0], F2[0, 1] = 1.0, (-1.0) ** i
            Fi = np.zeros((hip.pad_n(nx), nx, 2))
            Fi[:nx, 0, 0], Fi[:nx, 0, 1] = 1.0 / P, ((-1.0) ** i) / P
            for w in range(1, nx):
                ang = 2.0 * np.pi * ((w * i) % P) / P
                F2[w, 0], F2[w, 1] = np.cos(ang), np.sin(ang)
                Fi[:nx, w, 0], Fi[:nx, w, 1] = 2.0 / P * np.cos(ang), 2.0 / P * np.sin(ang)
            self._edge_c = (hip.to_dev(F2.reshape(-1, nx), self.device), hip.to_dev(Fi.reshape(-1, 2 * nx), self.device))
        return self._edge_c

    def edge_supported(self):
        return True

    def edge_eigen(self, E):
        """Spectrum of one boundary slab of an operator.  E: (>= ny*nx rows) x (nx*nz) view of the slab's columns, row (jy, jx),
        column (ix, iz), with E[(jy, jx), (ix, iz)] = kappa_jy(ix - jx, iz) exactly (lattice survey: the node offsets are bit-identical
        for equal index differences, hip.lattice_plan).  Returns V [nx slots][2 x J (C / S, jy)][2 x nz (cos / sin, iz)], J = ny rounded
        up to 64."""
        nx, ny, nz, J = self.nx, self.ny, self.nz, self.J
        Kc_s, Ks_s, K64 = self._edge_spectra(E)
        V = torch.zeros((nx, 2, J, 2, nz), dtype=F64, device=self.device)           # [slot][C | S][jy (J slots)][cos | sin][iz]
        V[1:, 0, :ny, 0], V[1:, 0, :ny, 1] = Kc_s[1:], Ks_s[1:]
        V[1:, 1, :ny, 0], V[1:, 1, :ny, 1] = -Ks_s[1:], Kc_s[1:]
        V[0, 0, :ny, 0] = Kc_s[0]                                                   # frequency 0
        V[0, 1, :ny, 1] = K64                                                       # frequency nx (a cosine, stored in the slot's second place)
        return V.view(nx, 2 * J, 2 * nz)

    def _edge_spectra(self, E):
        """x-DFT of the slab's stencil kappa_jy(d, iz): (Kc [slot][jy][iz], Ks [slot][jy][iz], K_nx [jy][iz]); slot 0 carries frequency 0."""
        nx, ny, nz = self.nx, self.ny, self.nz
        F2, _ = self._edge_consts()
        Ev = E[:ny * nx]
        nzp = hip.pad_n(nz)
        kap = torch.zeros((2, ny, nx, nzp), dtype=F64, device=self.device)         # [+ / -][jy][d][iz | compute-extent padding]
        kap[0, :, :, :nz] = Ev[0::nx].reshape(ny, nx, nz)                          # d = ix >= 0   (sensor column jx = 0)
        kap[1, :, 1:, :nz] = Ev[:, :nz].reshape(ny, nx, nz)[:, 1:]                 # d = -jx < 0   (voxel column ix = 0)
        T = torch.empty((2 * ny * 2 * nx * nz + 4096,), dtype=F64, device=self.device)[:2 * ny * 2 * nx * nz].view(2, ny, 2 * nx, nz)   # [+ / -][jy][(slot, cs)][iz]
        hip.gemm_batched(True, hip.pad_n(2 * nx), nzp, nx, F2, nx, 0, kap, nzp, nx * nzp, T, nz, 2 * nx * nz, 2 * nx, nz, 2 * ny)
        T = T.view(2, ny, nx, 2, nz)
        even = T[0] + T[1]                                                          # cosine parts: kappa(d) and kappa(-d) add
        Ks = (T[0] - T[1])[:, :, 1]                                                 # sine parts: they subtract
        Kc = even[:, :, 0]
        return Kc.permute(1, 0, 2).contiguous(), Ks.permute(1, 0, 2).contiguous(), even[:, 0, 1].contiguous()

    def edge_rows(self, X, nrows, V, out):
        """out[r, :ny*nx] += X[r, :nx*nz] . E^T for r < nrows, E given by its spectrum V (edge_eigen).  X: view starting at the slab's
        first column (unit column stride, even row stride); the compute extent of the first GEMM overhangs the plane by up to
        pad128(nz) doubles (inside the next slab / the next row everywhere but at the end of the buffer: staged there)."""
        nx, ny, nz, sp, J = self.nx, self.ny, self.nz, self.sp, self.J
        pl, RB = nx * nz, self.EDGE_ROWS
        F2, FiT = self._edge_consts()
        assert X.stride(1) == 1 and X.stride(0) % 2 == 0 and out.stride(1) == 1
        nzp, nxp = hip.pad_n(nz), hip.pad_n(nx)
        end = X.storage_offset() + (nrows - 1) * X.stride(0) + pl + nzp
        if end > X.untyped_storage().nbytes() // 8:                                 # no slack behind the last row: stage the rows
            Xc = sp.buf("LG_EX", nrows * pl + nzp)[:nrows * pl].view(nrows, pl)
            Xc.copy_(X[:nrows, :pl])
            X = Xc
        for r0 in range(0, nrows, RB):
            R = min(RB, nrows - r0)
            Rp = (R + 127) // 128 * 128
            Xh = sp.buf("LG_EXh", RB * 2 * nx * max(nz, J))                          # [row][(slot, cs)][iz]
            hip.gemm_batched(True, hip.pad_n(2 * nx), nzp, nx, F2, nx, 0, X[r0:], nz, X.stride(0), Xh, nz, 2 * nx * nz, 2 * nx, nz, R)
            o2 = sp.buf("LG_EO2", nx * 2 * max(J, nz) * RB)                          # [slot][(C | S, jy)][row]
            hip.gemm_batched(False, 2 * J, Rp, 2 * nz, V, 2 * nz, 2 * J * 2 * nz, Xh, 2 * nx * nz, 2 * nz, o2, Rp, 2 * J * Rp, 2 * J, R, nx)
            oT = sp.buf("LG_EOT", max(J, nz) * nx * RB)                              # [jy][jx][row]
            hip.gemm_batched(True, nxp, Rp, 2 * nx, FiT, 2 * nx, 0, o2, J * Rp, Rp, oT, Rp, nx * Rp, nx, R, ny)
            out[r0:r0 + R, :ny * nx] += oT[:ny * nx * Rp].view(ny * nx, Rp)[:, :R].t()

    # ---- the TRANSPOSED application: rows of L^-1 (one operator's columns) -> rows of L^-1 A  ---------------------------------------
    # Z[r][iy, ix, iz] = sum_(jy, jx) l_r[jy, jx] A[(jy, jx), (iy, ix, iz)]:  on the interior slabs a (y, x) convolution of the row's
    # sensor image with the stencil table Q (z as a channel, Q even: the same eigen-data Lambda as the Gram),
    #     Z_r[:, :, iz] = crop (Gy x Gx)^T [ Lambda[:, :, iz] * ((Gy x Gx) l_r) ],
    # on the two boundary slabs an x-convolution per sensor row jy through the full real DFT.  2e8 flop per row instead of the
    # 2 Ms N = 2.1e9 of the GEMM  L^-1[:, operator columns] A  that the transposed posterior path would otherwise spend.
    #   1. lhat = Gy l Gx^T                                   geobo_xz2d(_fold), one 64 x 64 plane per row
    #   2. W[kx][iz][ky] = LambdaW[kx][iz][ky] * lhat[ky][kx]    geobo_lattice_wbuild (8.4 MB per row, write bound)
    #   3. U[ix][(iz, ky)] = Gx^T[ix][kx] W[kx][(iz, ky)]        geobo_gemm_batched (NN, one 128 x 8192 x 128 product per row)
    #   4. Z[iy][(ix, iz)] = Gy0^T[iy][ky] U[(ix, iz)][ky]       geobo_gemm_batched (NT; rows iy = 0, ny-1 of Gy^T zeroed)
    #   5. boundary slabs: edge_apply_transpose (three batched GEMMs per 1024 rows)
    def transpose_tables(self, lam):
        """LambdaW[kx][iz][ky] from the Gram's eigen-data lam = Lambda^T[ky][z][kx] / (Py Px)."""
        return lam.view(self.Py, self.nz, self.Px).permute(2, 1, 0).contiguous().view(-1)

    def _lhat(self, Lrows, r0, Rb, lh):
        """lh[r] (Py x Px) = Gy l_r Gx^T for the sensor images l_r = Lrows[r0 + r] (ny x nx): the radix-2 / fused two-axis kernels where
        they are instantiated, two batched GEMM passes otherwise."""
        nx, ny, Px, Py, sp = self.nx, self.ny, self.Px, self.Py, self.sp
        if sp.fold and ny == nx and "y" in sp.F:
            hip.xz2d_fold(False, ny, Rb, 1, Lrows[r0:], Lrows.stride(0), ny * nx, sp.F["y"], sp.F["x"], lh, Py * Px, Py * Px)
        elif (ny, nx) in hip.XZ2D_SHAPES:
            hip.xz2d(False, ny, nx, Rb, 1, Lrows[r0:], Lrows.stride(0), ny * nx, sp.G["y"], sp.G["x"], lh, Py * Px, Py * Px)
        else:
            src = Lrows[r0:]
            nyp = hip.pad_n(ny)
            end = src.storage_offset() + (Rb - 1) * src.stride(0) + nyp * nx
            if end > src.untyped_storage().nbytes() // 8:                             # compute rows of the first pass overhang the image
                Lc = sp.buf("LG_LX", Rb * ny * nx + nyp * nx)[:Rb * ny * nx].view(Rb, ny * nx)
                Lc.copy_(src[:Rb, :ny * nx])
                src = Lc
            t1 = sp.buf("LG_Lt", Rb * ny * Px + nyp * Px)                             # [row][jy][kx]
            hip.axis_pass(sp.fold, False, False, nyp, hip.pad_n(Px), nx, src, nx, src.stride(0), sp.G["x"], nx, 0, t1, Px, ny * Px, ny, Px, Rb)
            hip.axis_pass(sp.fold, True, False, hip.pad_n(Py), hip.pad_n(Px), ny, sp.G["y"], ny, 0, t1, Px, ny * Px, lh, Px, Py * Px, Py, Px, Rb)

    def apply_transpose(self, Lrows, nrows, lamW, out):
        """out[r, :ny*nx*nz] = interior-slab part of  sum_c Lrows[r, c] A[c, :]  (boundary slabs zero), r < nrows.
        Lrows: (>= nrows x ny*nx) view of L^-1's columns of this operator (16-byte aligned, even row stride)."""
        nx, ny, nz, Px, Py, sp = self.nx, self.ny, self.nz, self.Px, self.Py, self.sp
        assert Lrows.stride(1) == 1 and Lrows.stride(0) % 2 == 0 and out.stride(1) == 1
        if getattr(self, "_GyT0", None) is None:
            g = sp.GT["y"].clone()                     # (ny x Py, padded rows): output rows of the two boundary slabs do not come from Q
            g[0] = 0.0
            g[ny - 1] = 0.0
            self._GyT0 = g
        R = self.Rz
        for r0 in range(0, nrows, R):
            Rb = min(R, nrows - r0)
            lh = sp.buf("LG_Lh", R * Py * Px)
            self._lhat(Lrows, r0, Rb, lh)
            W = sp.buf("LG_W", R * Px * nz * Py)
            hip.lattice_wbuild(Rb, Py, Px, nz, lamW, lh, W)
            U = sp.buf("LG_U", R * nx * nz * Py)
            if sp.x_mfma and (nz * Py) % 16 == 0 and 2 * nx * nz * Py * 8 < (1 << 31):
                # x synthesis as a radix-4 axis pass (geobo_spectral_axis): nz Py contiguous modes per plane, Px -> nx planes per row
                hip.spectral_axis(True, nx, nz * Py, nz * Py, nz * Py, Px * nz * Py, nx * nz * Py, Rb, W, U)
            else:
                hip.axis_pass(sp.fold, True, True, hip.pad_n(nx), nz * Py, Px, sp.GT["x"], Px, 0, W, nz * Py, Px * nz * Py, U, nz * Py, nx * nz * Py, nx, nz * Py, Rb)
            hip.axis_pass(sp.fold and 2, False, 2, hip.pad_n(ny), nx * nz, Py, self._GyT0, Py, 0, U, Py, nx * nz * Py, out[r0:], nx * nz, out.stride(0), ny, nx * nz, Rb)

    def transpose_tables3(self, lam):
        """Lambda3[iz][ky][kx] (spectral planes per z channel) from the Gram's eigen-data lam = Lambda^T[ky][z][kx] / (Py Px)."""
        return lam.view(self.Py, self.nz, self.Px).permute(1, 0, 2).contiguous().view(-1)

    def zx_supported(self):
        return self.nx == self.ny == self.nz == 64 and self.sp.fold and "y" in self.sp.F and self.sp.opts["z_fused"]

    def apply_transpose_zx(self, Lrows, nrows, lam3, out):
        """The same product as apply_transpose with steps 2-4 as ONE fused inverse two-axis transform per (row, z channel) plane
        (geobo_xz2d_fold_inv_strided: the radix-2 kernel of the covariance product, 17 ms for the 786 432 planes of a 64^3 step where the
        two batched-GEMM inverse steps -- half of whose 128-row tiles is padding -- took 85 ms), writing rows in the layout
        out[r][iy][iz][ix]: x and z trade places in every plane, which the covariance product that follows does not mind (square planes,
        identical transform matrices; its Toeplitz tables and the final sums are transposed accordingly).  Boundary slabs: as computed
        from the stencil table (wrong) -- edge_apply_transpose(..., zx=True) overwrites them.  nx = ny = nz = 64."""
        nx, ny, nz, Px, Py, sp = self.nx, self.ny, self.nz, self.Px, self.Py, self.sp
        assert self.zx_supported() and Lrows.stride(1) == 1 and Lrows.stride(0) % 2 == 0 and out.stride(1) == 1
        R = self.R
        for r0 in range(0, nrows, R):
            Rb = min(R, nrows - r0)
            lh = sp.buf("LG_Lh", R * Py * Px)
            hip.xz2d_fold(False, ny, Rb, 1, Lrows[r0:], Lrows.stride(0), ny * nx, sp.F["y"], sp.F["x"], lh, Py * Px, Py * Px)
            if self.sp.opts["z_mul"]:
                # W[r][iz] = Lambda3[iz] * lhat_r is formed inside the inverse kernel, chunk by chunk, from the two cache-resident factors
                hip.xz2d_fold_inv_mul(ny, Rb, nz, lam3, Py * Px, lh, Py * Px, sp.F["y"], sp.F["x"], out[r0:], out.stride(0), nx, nz * nx)
                continue
            W = sp.buf("LG_W", R * nz * Py * Px)
            hip.lattice_wplanes(Rb, Py, Px, nz, lam3, lh, W)
            hip.xz2d_fold_inv_strided(ny, Rb, nz, W, nz * Py * Px, Py * Px, sp.F["y"], sp.F["x"], out[r0:], out.stride(0), nx, nz * nx)

    def edge_eigen_t(self, E):
        """Spectrum of one boundary slab for the transposed application: Vt [nx slots][2 x nz (C / S, iz)][2 x J (cos / sin, jy)]
        (same transforms Kc, Ks of the slab's x-Toeplitz stencil as edge_eigen; the convolution theorem instead of the correlation's)."""
        nx, ny, nz, J = self.nx, self.ny, self.nz, self.J
        Kc_s, Ks_s, K64 = self._edge_spectra(E)                                      # [slot][jy][iz], [slot][jy][iz], [jy][iz]
        # (spare slots of zeros behind the last one: the compute rows of the per-slot GEMM, pad128(2 nz), overhang it by pad128(2 nz) - 2 nz
        #  rows -- THREE slots at nz = 16; with the one spare slot of rounds 3-4 the last slot's operand tile read 64 KB past the
        #  allocation there: harmless values, never stored, but an unmapped page behind the tensor aborts the process -- found in round 5
        #  when a larger Cholesky workspace moved the allocations of the 16^3 row-form tests)
        spare = (hip.pad_n(2 * nz) - 2 * nz + 2 * nz - 1) // (2 * nz)
        Vt = torch.zeros((nx + max(spare, 1), 2, nz, 2, J), dtype=F64, device=self.device)       # [slot][C | S][iz][cos | sin][jy (J slots)]
        Kc_t, Ks_t = Kc_s.transpose(1, 2), Ks_s.transpose(1, 2)                      # [slot][iz][jy]
        Vt[1:nx, 0, :, 0, :ny], Vt[1:nx, 0, :, 1, :ny] = Kc_t[1:], -Ks_t[1:]         # Outc = Lc Kc - Ls Ks
        Vt[1:nx, 1, :, 0, :ny], Vt[1:nx, 1, :, 1, :ny] = Ks_t[1:], Kc_t[1:]          # Outs = Lc Ks + Ls Kc
        Vt[0, 0, :, 0, :ny] = Kc_t[0]                                                # frequency 0
        Vt[0, 1, :, 1, :ny] = K64.t()                                                # frequency nx
        return Vt.view(-1, 2 * nz, 2 * J)

    def edge_apply_transpose(self, Lrows, nrows, Vt, out, zx=False):
        """out[r, (ix, iz)] = sum_(jy, jx) Lrows[r, jy*nx+jx] kappa_jy(ix - jx, iz)  for r < nrows: one boundary slab of L^-1 A.
        out: (>= nrows x nx*nz) view of the slab's columns; zx: the slab is stored as [iz][ix] (apply_transpose_zx's row layout)."""
        nx, ny, nz, sp, J = self.nx, self.ny, self.nz, self.sp, self.J
        RB = self.EDGE_ROWS
        F2, FiT = self._edge_consts()
        nyp, nxp = hip.pad_n(ny), hip.pad_n(nx)
        end = Lrows.storage_offset() + (nrows - 1) * Lrows.stride(0) + nyp * nx
        if end > Lrows.untyped_storage().nbytes() // 8:                               # compute extents of the first GEMM overhang the row
            Lc = sp.buf("LG_EX", nrows * ny * nx + nyp * nx)[:nrows * ny * nx].view(nrows, ny * nx)
            Lc.copy_(Lrows[:nrows, :ny * nx])
            Lrows = Lc
        for r0 in range(0, nrows, RB):
            R = min(RB, nrows - r0)
            Rp = (R + 127) // 128 * 128
            Lh = sp.buf("LG_EXh", RB * 2 * nx * max(nz, J))                           # [row][(slot, cs)][jy]
            if ny < J:
                Lh.zero_()                                                            # (jy slots >= ny meet zero columns of Vt: keep them finite)
            hip.gemm_batched(False, hip.pad_n(2 * nx), nyp, nx, F2, nx, 0, Lrows[r0:], nx, Lrows.stride(0), Lh, J, 2 * nx * J, 2 * nx, ny, R)
            o2 = sp.buf("LG_EO2", nx * 2 * max(J, nz) * RB)                           # [slot][(C | S, iz)][row]
            hip.gemm_batched(False, hip.pad_n(2 * nz), Rp, 2 * J, Vt, 2 * J, 2 * nz * 2 * J, Lh, 2 * nx * J, 2 * J, o2, Rp, 2 * nz * Rp, 2 * nz, R, nx)
            oT = sp.buf("LG_EOT", max(J, nz) * nx * RB)                               # [iz][ix][row]
            hip.gemm_batched(True, nxp, Rp, 2 * nx, FiT, 2 * nx, 0, o2, nz * Rp, Rp, oT, Rp, nx * Rp, nx, R, nz)
            res = oT[:nz * nx * Rp].view(nz, nx, Rp)[:, :, :R]
            if zx:
                out[r0:r0 + R, :nx * nz].view(R, nz, nx).copy_(res.permute(2, 0, 1))
            else:
                out[r0:r0 + R, :nx * nz].view(R, nx, nz).copy_(res.permute(2, 1, 0))

    @staticmethod
    def fast(nx, ny, nz):
        """Grids whose x step and back-transform run on the fused kernels (geobo_xcorr_reduce(_fold), geobo_xz2d(_fold)): the planner's
        table (plan.lattice_gram_fast) is the one definition."""
        from .plan import lattice_gram_fast
        return lattice_gram_fast(nx, ny, nz)

    @staticmethod
    def supported(nx, ny, nz):
        """Any grid of the spectral route (extents in multiples of 16): the stages without a fused instance for the extent run as
        batched MFMA GEMMs + geobo_lamdot_z (plan.lattice_gram_supported)."""
        from .plan import lattice_gram_supported
        return lattice_gram_supported(nx, ny, nz)

    def eigen(self, Q, tol=1e-11):
        """Lambda^T[ky][z][kx] / (Py Px) from the stencil table Q[(2ny-3)][(2nx-1)][nz]; None if Q is not even in both offsets."""
        nx, ny, nz, Px, Py = self.nx, self.ny, self.nz, self.Px, self.Py
        cy, cx = ny - 2, nx - 1                                         # index of offset 0
        sp = self.sp
        Qh = sp.buf("LG_Qh", ny * nx * nz)[:ny * nx * nz].view(ny, nx, nz)   # (slack behind it: compute tiles overhang)
        Qh.zero_()
        Qh[:ny - 1] = Q[cy:cy + ny - 1, cx:cx + nx]
        # evenness in both offsets, checked on the device with ONE host read: max deviation of the three mirrored quadrants / max |Q|
        dev = [Q.abs().max()]
        for sy, sx in ((-1, 1), (1, -1), (-1, -1)):
            ys = torch.arange(0, ny - 1, device=self.device) * sy + cy
            xs = torch.arange(0, nx, device=self.device) * sx + cx
            dev.append((Q[ys][:, xs] - Qh[:ny - 1]).abs().max())
        dev = torch.stack(dev).tolist()
        if max(dev[1:]) > tol * dev[0]:
            return None
        T = sp.buf("LG_T", ny * Px * nz)
        hip.gemm_batched(True, hip.pad_n(Px), hip.pad_n(nz), nx, sp.E["x"], nx, 0, Qh, nz, nx * nz, T, nz, Px * nz, Px, nz, ny)
        lam = torch.empty(Py * Px * nz + 4096, dtype=F64, device=self.device)
        hip.gemm_batched(True, hip.pad_n(Py), hip.pad_n(Px * nz), ny, sp.E["y"], ny, 0, T, Px * nz, 0, lam, Px * nz, 0, Py, Px * nz, 1)
        lam[:Py * Px * nz].mul_(1.0 / float(Py * Px))
        return lam[:Py * Px * nz].view(Py, Px, nz).transpose(1, 2).contiguous().view(-1)       # z-major planes: [ky][z][kx]

    def flops(self, rows, Ly=None):
        nx, ny, nz, Px, Py = self.nx, self.ny, self.nz, self.Px, self.Py
        Ly = ny if Ly is None else Ly
        f = self.sp.fold and nx == nz == 64
        h = 0.25 if f else 1.0            # radix-4 x step: a quarter of the MFMAs of the plain product
        hb = 0.25 if f else 1.0           # back-transform on the radix-4 inverse kernel: a quarter
        zsum = 0.0 if self.fast(nx, ny, nz) else 2.0 * Py * Px * nz  # (the stand-alone scaling + channel sum of the batched-GEMM form, fp64 VALU)
        hy = 0.5 if (f and (Py, Ly) in hip.YMUL_SHAPES) else 1.0     # y step on geobo_ymul_fold: radix 2
        return rows * (2.0 * (hy * hip.pad_n(Py) * nx * nz * Ly + h * Py * Px * nx * nz + hb * (ny * Py * Px + ny * nx * Px)) + zsum)

    def gram_rows(self, X, nrows, lam, out, y0=0, y1=None):
        """out[r, :ny*nx] = interior-slab part of (A K)[r] . A^T for r < nrows.  X: (>= nrows x >= (y1-y0)*nx*nz) rows of A K for
        the block's property, voxel columns of the y-slab [y0, y1) only (a rank of a column-sharded run holds just its slab: the
        partial correlations add up in the all-reduce of AkA); out: (>= nrows x >= ny*nx) view of the AkA block column."""
        nx, ny, nz, Px, Py = self.nx, self.ny, self.nz, self.Px, self.Py
        y1 = ny if y1 is None else y1
        Ly = y1 - y0
        assert Ly % 16 == 0
        sp = self.sp
        plane = nx * nz
        assert X.stride(1) == 1 and out.stride(1) == 1 and X.stride(0) % 2 == 0 and out.stride(0) % 2 == 0
        gy = self.Gy0[:, y0:]                        # columns of this slab (row stride ny; rows behind are padding / slack)
        for r0 in range(0, nrows, self.R):
            R = min(self.R, nrows - r0)
            y1b = sp.buf("LG_Y1", R * Py * plane)
            if (Py, Ly) in hip.YMUL_SHAPES and plane % 64 == 0:
                # G_y in registers, rows streamed once; radix 2 where the slab starts on an even y (Gy0 keeps the pair structure: its
                # zeroed columns are zero in both rows of a pair)
                hip.ymul(Py, Ly, plane, R, gy, X[r0:], X.stride(0), y1b, Py * plane, fold=sp.fold and y0 % 2 == 0)
            elif sp.axis_mfma(ny) and y0 == 0 and Ly == ny and plane % 16 == 0 and 2 * ny * plane * 8 < (1 << 31) and X.stride(0) >= ny * plane:
                # the whole y axis: radix-4 axis pass, the two boundary slabs masked inside the kernel (what Gy0's zeroed columns do)
                hip.spectral_axis(False, ny, plane, plane, plane, X.stride(0), Py * plane, R, X[r0:], y1b, mask_ends=True)
            else:
                # (radix-2 when the slab starts on an even y: the parity of the local input index is the basis row pair's)
                hip.axis_pass(sp.fold and y0 % 2 == 0, True, False, hip.pad_n(Py), hip.pad_n(plane), Ly, gy, ny, 0, X[r0:], plane, X.stride(0),
                              y1b, plane, Py * plane, Py, plane, R)
            s = sp.buf("LG_S", R * Py * Px)
            if sp.fold and nx == nz and "x" in sp.F:
                hip.xcorr_reduce_fold(nx, R, Py, y1b, Py * plane, plane, sp.F["x"], lam, s, Py * Px, Px)
            elif (nx, nz) == (64, 64):
                hip.xcorr_reduce(nx, nz, R, Py, y1b, Py * plane, plane, sp.G["x"], lam, s, Py * Px, Px)
            else:
                # any other extent: D[r, ky] = Gx X[r, ky] as a batch of small GEMMs, then the eigenvalue scaling and the channel sum
                if sp.fold and nz <= 128:
                    # one launch: radix-2 analysis with the scaling and the channel sum as its epilogue, D never written
                    hip.gemm_fold_lamdot(Px, nz, nx, sp.G["x"], nx, y1b, nz, plane, self.lam_oz(lam), Py, s, R * Py)
                else:
                    D = sp.buf("LG_D", R * Py * Px * nz)
                    hip.axis_pass(sp.fold, True, False, hip.pad_n(Px), hip.pad_n(nz), nx, sp.G["x"], nx, 0, y1b, nz, plane, D, nz, Px * nz, Px, nz, R * Py)
                    hip.lamdot_z(R * Py, Py, Px, nz, D, self.lam_oz(lam), s)
            # "x" of geobo_xz2d is this grid's y axis, its "z" this grid's x axis: S_r (Py x Px) -> Gy^T S_r Gx (ny x nx)
            if sp.fold and ny == nx and "y" in sp.F:
                hip.xz2d_fold(True, ny, R, 1, s, Py * Px, Py * Px, sp.F["y"], sp.F["x"], out[r0:], out.stride(0), ny * nx)
            elif (ny, nx) in hip.XZ2D_SHAPES:
                hip.xz2d(True, ny, nx, R, 1, s, Py * Px, Py * Px, sp.GT["y"], sp.GT["x"], out[r0:], out.stride(0), ny * nx)
            else:
                t1 = sp.buf("LG_Bt", R * ny * Px + hip.pad_n(ny) * Px)             # [row][iy][kx]
                hip.axis_pass(sp.fold, True, True, hip.pad_n(ny), hip.pad_n(Px), Py, sp.GT["y"], Py, 0, s, Px, Py * Px, t1, Px, ny * Px, ny, Px, R)
                hip.gemm_batched(True, hip.pad_n(ny), hip.pad_n(nx), Px, t1, Px, ny * Px, sp.G["x"], nx, 0, out[r0:], nx, out.stride(0), ny, nx, R)

    def lam_oz(self, lam):
        """The Gram's eigen-data as [ky][kx][z] (what geobo_lamdot_z reads) from eigen()'s [ky][z][kx]."""
        hit = self._lam_oz.get(lam.data_ptr())
        if hit is None or hit[0] is not lam:
            self._lam_oz = {k: v for k, v in self._lam_oz.items() if v[0] is not lam}
            if len(self._lam_oz) > 4:
                self._lam_oz = {}
            hit = self._lam_oz[lam.data_ptr()] = (lam, lam.view(self.Py, self.nz, self.Px).transpose(1, 2).contiguous().view(-1))
        return hit[1]
