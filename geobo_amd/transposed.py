"""The one-rank structured step on the fused n = 64 kernels (plan.Route.family "single") and the pieces of the transposed posterior
every family shares: mixin of engine.PosteriorEngine.

  _sym_ok / _assemble_AkA_sym   AkA from the three blocks of A K its lower triangle needs, through the lattice Gram (inversion.py:96)
  _lattice_Z                    rows of L^-1 A as lattice convolutions with the operator's stencil table (inversion.py:114 re-associated)
  _posterior_zpath              V = (L^-1 A3) K through the covariance kernels, squared and summed on the way out (inversion.py:114-117,238)
  _mean_rows, _drill_rows_ss    the mean as three rows through the covariance product; the rows behind the sensor rows
"""

import torch

from . import hip
from .operators import StreamedOperator

F64 = hip.F64


class TransposedPosteriorMixin:
    def _sym_ok(self, A_g, A_m):
        """The symmetric plan of A K / AkA (see _assemble_AK): the step will run in the transposed order (_zpath_static_ok) and AkA is
        the lattice Gram for both operators, boundary slabs through their spectra."""
        if not self._zpath_static_ok():
            return False
        return (self._gram is not None and self._gram.edge_supported() and self.Ms_pad == self.nx * self.ny
                and all(self._lam.get(f) is not None and self._lam[f][0] is A for f, A in (("grav", A_g), ("magn", A_m))))

    def _assemble_AkA_sym(self, AkA, AK, M_pad, A_g, A_m, sel_t, lengths, name, amp, gp_sigma, props):
        """AkA on one device from the blocks of A K the symmetric plan keeps, row block by row block through the lattice Gram:
        [grav rows -> grav columns], [grav rows -> magn columns] (transposed into the lower-left block, which is what the
        factorisation reads), [magn rows -> magn columns], drill rows -> both.  A quarter of the Gram's and of A K's work less than the
        block-column form, which computes the lower-left block from A_m K_10 as well."""
        gram, pl, ny, Msp, nc = self._gram, self.nx * self.nz, self.ny, self.Ms_pad, self.nc
        Md, off_d = 0 if sel_t is None else sel_t.numel(), 2 * self.Ms_pad
        lam = {0: self._lam["grav"][1], 1: self._lam["magn"][1]}
        ops = {0: A_g, 1: A_m}

        def edge_cols(sp_, k, iy):
            A = ops[sp_]
            if isinstance(A, StreamedOperator):
                return A.edge[:, k * pl:(k + 1) * pl] if A.lattice is not None else A.slab_into(self._workspace2d("op_slab", Msp, pl), iy, iy + 1)
            return A[:, iy * pl:(iy + 1) * pl]

        def rows_times_AT(X, nrows, sp_, out):
            gram.gram_rows(X, nrows, lam[sp_], out, 0, ny)
            for k, iy in enumerate((0, ny - 1)):
                gram.edge_rows(X[:, iy * pl:], nrows, self._edge_spectrum(("grav", "magn")[sp_], k, edge_cols(sp_, k, iy)), out)
        blk = lambda r0, j: AK[r0:, props.index(j) * nc:(props.index(j) + 1) * nc]

        def run():
            rows_times_AT(blk(0, 0), self.Ms, 0, AkA[0:, 0:Msp])
            rows_times_AT(blk(0, 1), self.Ms, 1, AkA[0:, Msp:2 * Msp])
            rows_times_AT(blk(Msp, 1), self.Ms, 1, AkA[Msp:, Msp:2 * Msp])
            AkA[Msp:2 * Msp, :Msp] = AkA[:Msp, Msp:2 * Msp].t()
            if Md:
                rows_times_AT(blk(off_d, 0), Md, 0, AkA[off_d:, 0:Msp])
                rows_times_AT(blk(off_d, 1), Md, 1, AkA[off_d:, Msp:2 * Msp])
        self._timed("aka_lattice", gram.flops(3 * self.Ms + 2 * Md, ny), run)
        return self._finish_AkA(AkA, M_pad, sel_t, lengths, name, amp, gp_sigma)

    def _zpath_static_ok(self):
        """The one-rank transposed posterior on the fused kernels (plan.Route.single): one rank, fp64 A K, the radix-2 transform kernels
        and the Toeplitz y stage of this grid, unpadded sensor rows and voxel columns."""
        if not self.route.single or self.exchange:
            return False
        self._spectral_product()
        return True

    def _zpath_ok(self, AK, props, A_g, A_m):
        """... and the covariance generators of the last A K assembly."""
        return (self._zpath_static_ok() and AK is not None and AK.dtype == F64
                and all((s_, j) in self._gens for s_ in (0, 1) for j in props))

    def _resident_operator(self, A, func):
        """A materialised copy of a forward operator that the route so far only kept implicitly (stencil table + boundary slabs)."""
        if not isinstance(A, StreamedOperator):
            return A
        R = self._workspace2d("A_" + func, self.Ms_pad, self.N_pad)
        xed, yed, zed = A.axes_dev
        hip.a_sens(A.func, A.Bv, A.locd, self.nx, self.ny, self.nz, xed, yed, zed, A.mul, A.div, R, plan=A.plan, ws=A.lws)
        return R

    def _lattice_Z(self, Lview, nrows, func, A, out, zx=False, edge=None):
        """out[r, :N] = sum_c Lview[r, c] A[c, :]  for a lattice-survey operator, without touching A: interior slabs through the stencil
        table's eigen-data, the two boundary slabs through their x-DFT spectra (lattice_gram.apply_transpose / edge_apply_transpose)."""
        gram, pl, ny = self._gram, self.nx * self.nz, self.ny
        lam = self._lam[func][1]
        hit = self._lamW.get((func, zx))
        if hit is None or hit[0] is not lam:
            hit = self._lamW[(func, zx)] = (lam, gram.transpose_tables3(lam) if zx else gram.transpose_tables(lam))
        if zx:
            gram.apply_transpose_zx(Lview, nrows, hit[1], out)      # rows as [iy][iz][ix]
        else:
            gram.apply_transpose(Lview, nrows, hit[1], out)
        for k, iy in enumerate((0, ny - 1)):
            if edge is not None:                      # (row form: the two boundary slabs of every sensor, engine.operator)
                ycols = edge[k]
            elif isinstance(A, StreamedOperator) and A.lattice is not None:
                ycols = A.edge[:, k * pl:(k + 1) * pl]
            elif isinstance(A, StreamedOperator):
                ycols = A.slab_into(self._workspace2d("op_slab", self.Ms_pad, pl), iy, iy + 1)
            else:
                ycols = A[:, iy * pl:(iy + 1) * pl]
            key = (func, k)
            vt = self._edgeVt.get(key)
            if vt is None or vt[0] != ycols.data_ptr():
                vt = self._edgeVt[key] = (ycols.data_ptr(), gram.edge_eigen_t(ycols))
            gram.edge_apply_transpose(Lview, nrows, vt[1], out[:, iy * pl:(iy + 1) * pl], zx=zx)

    def _posterior_zpath(self, Linv, AK, u, A_g, A_m, sel_t, lengths, W, name, amp, props, M_pad):
        """Posterior mean and variance in the TRANSPOSED order (round 3).  V = L^-1 (A3 K) is (L^-1 A3) K as well, and A3 is block
        diagonal: applying L^-1 to the forward operators costs M x Ms x N per operator -- independent of the number of property blocks
        and only over the operator's own columns of L^-1 -- where applying it to A K costs M^2 / 2 x N per property block:
            Z_g = Linv[:, grav columns] A_g,   Z_m = Linv[:, magn columns] A_m                (fp64 MFMA GEMMs, triangular X: 1.8e13 flop
                                                                                               at 64^3 instead of 3.6e13)
            V_j = Z_g K_0j + Z_m K_1j  (+ the drill term in the last rows only: L^-1 is lower triangular)
        and the covariance products run through the same spectral kernels as A K, with the inverse transform squaring and summing
        its output planes over the rows instead of storing them (geobo_xz2d_fold_inv_ss): V is never written.  The mean needs no V
        at all: mu = (A K)^T (L^-T u), two weighted column sums.  Same arithmetic up to summation order (inversion.py:114-117)."""
        sp, N, Msp, P_c, Md = self._spectral, self.N, self.Ms_pad, len(props), 0 if sel_t is None else sel_t.numel()
        nx, ny, nz = self.nx, self.ny, self.nz
        cws = self._workspace("colgemv_ws", (max(hip.colgemv_ws_doubles(M_pad, M_pad), hip.colgemv_ws_doubles(Msp, self.N_pad)),))
        w = self._timed("posterior_mean", 0.0, lambda: hip.colgemv(Linv, u, ws=cws))
        Zg, Zm = self._workspace2d("Zg", 2 * Msp, N), self._workspace2d("Zm", Msp, N)
        # Z = L^-1[:, operator columns] A: on a lattice survey a (y, x) convolution of every row's sensor image with the operator's
        # stencil table (lattice_gram.apply_transpose: 2e8 flop per row), otherwise two triangular MFMA GEMMs (2.1e9 flop per row)
        lat = (self._gram is not None and self._gram.edge_supported() and Msp == nx * ny and self.route.opt("z_lattice")
               and all(self._lam.get(f) is not None and self._lam[f][0] is A for f, A in (("grav", A_g), ("magn", A_m))))
        if lat:
            gram = self._gram
            fl = 3 * Msp * (gram.flops(1, ny) + 2 * 3 * 2.0 * 128 * 128 * 64)

            zx = gram.zx_supported()          # rows of Z as [iy][iz][ix]: the fused inverse transform writes them, the products below follow

            def zlattice():
                self._lattice_Z(Linv[:2 * Msp, :Msp], 2 * Msp, "grav", A_g, Zg, zx=zx)
                self._lattice_Z(Linv[Msp:2 * Msp, Msp:2 * Msp], Msp, "magn", A_m, Zm, zx=zx)
            self._timed("posterior_zlattice", fl, zlattice)
            Ag = Am = None
            vec_of = lambda func, wv, out: self._lattice_Z(wv.view(1, -1), 1, func, A_g if func == "grav" else A_m, out)
        else:
            Ag = self._timed("a_sens_grav", 0.0, lambda: self._resident_operator(A_g, "grav"))
            Am = self._timed("a_sens_magn", 0.0, lambda: self._resident_operator(A_m, "magn"))
            tri = sum(min(256 * (bi + 1), Msp) for bi in range(Msp // 256)) * 256.0      # executed k-extent x rows of a triangular block
            fl = 2.0 * N * (2 * tri + 1.0 * Msp * Msp)
            alg = 2.0 * N * (2 * (Msp * (Msp + 1) / 2.0) + 1.0 * Msp * Msp)

            def zgemm():
                hip.gemm_nn(Linv[:2 * Msp, :Msp], Ag[:Msp, :N], Zg, x_lower=True)
                hip.gemm_nn(Linv[Msp:2 * Msp, Msp:2 * Msp], Am[:Msp, :N], Zm, x_lower=True)
            self._timed("posterior_zgemm", fl, zgemm, alg=alg)
            vec_of = lambda func, wv, out: hip.colgemv((Ag if func == "grav" else Am)[:Msp, :N], wv, out=out[0], ws=cws)
        mu_l = self._timed("posterior_mean", 0.0, lambda: self._mean_rows(w, sel_t, lengths, W, name, amp, props, vec_of)).reshape(-1)
        slots = hip.xz2d_fold_inv_ss_slots(nx, sp.R, ny)
        ss = [self._workspace("post_ss_%d" % jj, (slots, ny, nx * nz)) for jj in range(P_c)]
        for t in ss:
            t.zero_()
        gens_g, gens_m = [self._gens[(0, j)] for j in props], [self._gens[(1, j)] for j in props]
        zx = lat and zx
        swap = (lambda g: g.view(ny, sp.Px, sp.Pz).transpose(1, 2).contiguous().view(-1)) if zx else (lambda g: g)   # tables of transposed planes
        tg, tm = [swap(g) for g in gens_g], [swap(g) for g in gens_m]
        # blocks (0, 1) and (1, 0) of a symmetric prior coincide: three y-stage products per two-term row instead of four
        y2s = sp.y2s_tables(tg, tm) if tuple(props[:2]) == (0, 1) else None
        shared = y2s is not None and P_c == 2
        self._timed("posterior_spectral", sp.flops_ss(Msp, Msp, P_c, shared), lambda: sp.reduce_ss(Zg, 2 * Msp, tg, Zm, Msp, tm, ss, y2s=y2s),
                    valu=sp.valu_ss(Msp, Msp, P_c, shared))
        if zx:
            ssum = torch.stack([t.sum(0).view(ny, nz, nx).transpose(1, 2).reshape(-1) for t in ss])   # planes came out as [iz][ix]
        else:
            ssum = torch.stack([t.sum(0).reshape(-1) for t in ss])                        # (P_c, N), voxel order (iy, ix, iz)
        if Md:
            ssum = ssum + self._timed("posterior_drill_rows", 0.0, lambda: self._drill_rows_ss(
                Linv, 0, Md, sel_t, lengths, W, name, amp, props, gens_g, gens_m,
                (lambda Lv, n, func, out: self._lattice_Z(Lv, n, func, A_g if func == "grav" else A_m, out)) if lat else None, Ag, Am))
        return mu_l, (amp * 1.0 - ssum).reshape(-1)

    def _mean_rows(self, w, sel_t, lengths, W, name, amp, props, vec_of):
        """Posterior mean (P_c, N):  mu_j = (A3 K)[:, block j]^T w  re-associated as  K_.j (A3^T w)  -- the covariance blocks are
        symmetric, so three N-vectors (A_g^T w_g, A_m^T w_m, the drill weights scattered to their voxels) go through the covariance
        product as ONE row each (0.3 ms) where the weighted column sums of A K read all of it (35 GB at 64^3: 6 ms).
        vec_of(func, weights, out (1 x N)): out = A_func^T weights.  w = L^-T u (inversion.py:105,115)."""
        sp, N, Msp, P_c = self._spectral, self.N, self.Ms_pad, len(props)
        Md = 0 if sel_t is None else sel_t.numel()
        V = self._workspace2d("mean_rows", 4, N)
        vec_of("grav", w[:Msp], V[0:1])
        vec_of("magn", w[Msp:2 * Msp], V[1:2])
        terms = [(V[0:1], 0), (V[1:2], 1)]
        if Md:
            V[2].zero_()
            V[2][sel_t] = w[2 * Msp:2 * Msp + Md]
            terms.append((V[2:3], 2))
        mu = torch.zeros((P_c, N), dtype=F64, device=self.device)
        tmp = [self._workspace2d("mean_tmp_%d" % jj, 2, N) for jj in range(P_c)]
        for rows, s_ in terms:
            gens = [self._gens[(s_, j)] if s_ < 2 else
                    sp.eigenvalues(self._cov_table(hip.kernel_id(name, s_ != j), lengths[j], lengths[s_], W[s_][j], amp)) for j in props]
            sp.product(rows, 1, gens, tmp)
            for jj in range(P_c):
                mu[jj].add_(tmp[jj][0])
        return mu

    def _drill_rows_ss(self, Linv, d0, nd, sel_t, lengths, W, name, amp, props, gens_g, gens_m, lattice_Z, Ag, Am):
        """(P_c, N) sums of squares of V = L^-1 (A3 K) over the drill rows d0 .. d0 + nd of the row block behind the sensor rows:
        L^-1 is lower triangular, so only THESE rows see the drill columns.  Tiles of up to 128 rows through the storing covariance
        product, three terms (gravity, magnetic, drill block rows of K), squared and summed here.  lattice_Z(Lview, n, func, out):
        rows of L^-1 A on a lattice survey; None: MFMA GEMMs against the resident operators Ag / Am (whole 128-row tiles)."""
        sp, N, Msp, P_c, Md, T = self._spectral, self.N, self.Ms_pad, len(props), sel_t.numel(), 128
        Zgd, Zmd, Zdd = (self._workspace2d(nm, T, N) for nm in ("Zg_d", "Zm_d", "Zd_d"))
        Vd = [self._workspace2d("Vd_%d" % jj, T, N) for jj in range(P_c)]
        tmp = [self._workspace2d("Vt_%d" % jj, T, N) for jj in range(P_c)]
        gens_d = [sp.eigenvalues(self._cov_table(hip.kernel_id(name, 2 != j), lengths[j], lengths[2], W[2][j], amp)) for j in props]
        acc = torch.zeros((P_c, N), dtype=F64, device=self.device)
        for c0 in range(d0, d0 + nd, T):
            n = min(T, d0 + nd - c0)
            b0 = 2 * Msp + c0
            if lattice_Z is not None:
                lattice_Z(Linv[b0:b0 + n, :Msp], n, "grav", Zgd)
                lattice_Z(Linv[b0:b0 + n, Msp:2 * Msp], n, "magn", Zmd)
            else:
                nt = min(T, Linv.shape[0] - b0)          # (M_pad is a multiple of 256: whole tiles unless the caller's share starts mid-tile)
                hip.gemm_nn(Linv[b0:b0 + nt, :Msp], Ag[:Msp, :N], Zgd)
                hip.gemm_nn(Linv[b0:b0 + nt, Msp:2 * Msp], Am[:Msp, :N], Zmd)
            Zdd[:n].zero_()
            Zdd[:n, sel_t] = Linv[b0:b0 + n, 2 * Msp:2 * Msp + Md]
            sp.product(Zgd, n, gens_g, Vd)
            for Zx, gx in ((Zmd, gens_m), (Zdd, gens_d)):
                sp.product(Zx, n, gx, tmp)
                for jj in range(P_c):
                    Vd[jj][:n].add_(tmp[jj][:n])
            for jj in range(P_c):
                acc[jj].add_((Vd[jj][:n] ** 2).sum(0))
        return acc
