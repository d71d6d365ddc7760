"""The structured inversion step sharded by SENSOR ROWS (plan.Route.family == "rows"): mixin of engine.PosteriorEngine.

Reference arithmetic replaced: inversion.py:92-117 (predict3: K, A K A^T + S, Cholesky, V = L^-1 A K, mu, diag(K - V^T V)) on a survey
that sits on the cube's own x-y lattice (run_geobo.py:56-65), for world >= 1 ranks and ANY grid of the spectral route.  Rank r owns
Ms / G sensor rows of each forward operator and never holds a voxel-column shard, A K, or V:

  A K -> AkA   the rank's rows of A_g K_00, A_g K_01, A_m K_11 (all that AkA's lower triangle contracts) go through the covariance
               product a CHUNK of rows at a time and straight on through the lattice Gram (rows x A^T as a (y, x) correlation with the
               operator's stencil table): a chunk of A K lives in a scratch of a few GB and is overwritten by the next one
               (_rows_aka_local).  One all-gather of the (rows_r x 3 Ms) row blocks; the (magn, grav) block is the transpose of
               (grav, magn); drill rows replicated.
  Cholesky     replicated (keeps ranks bit-identical, no broadcast).
  posterior    transposed order V = (L^-1 A3) K: rows of Z = L^-1[own rows, operator columns] A as lattice convolutions, a chunk at a
               time, through the same covariance product, squared and summed on the way out (spectral.reduce_ss: fused into the
               inverse transform at n = 64, a stored batch + geobo_sumsq_accum elsewhere); the mean as three rows through the
               covariance product on every rank; ONE all-reduce of the P_c N partial sums of squares.

Chunking bounds the footprint by the scratch budget instead of by the cube: 128^3 x 3 properties (BASELINE config 5) runs in
~50 GB per rank where the column form needed a 104 GB A K shard.  With one chunk (64^3 from 2 ranks) the launches are those of
round 3's resident form, bit for bit.  Covariance tables are rounded through fp32 in the fp32-assembly mode (engine._cov_table);
everything else is fp64."""
import os

import torch

from . import hip
from .sharding import EmulatedGroup, agree, allreduce_sum_, gather_rows

F64 = hip.F64
ROW_SCRATCH_BYTES = 8 << 30       # per stage: the chunk's A K blocks (stage 1) / rows of Z (stage 3)


class RowFormMixin:
    # ---- what a rank holds of the operators ----------------------------------------------------------------------------------------
    def _rows_chunk(self, nblocks):
        """Rows per chunk: `nblocks` row blocks of N doubles each inside the scratch budget, whole transform batches."""
        sp = self._spectral
        budget = ROW_SCRATCH_BYTES
        ck = int(os.environ.get("GEOBO_ROW_CHUNK", "0")) or max(1, budget // (nblocks * self.N * 8))      # (GEOBO_ROW_CHUNK: tests)
        return max(sp.R, ck // sp.R * sp.R)

    def _rows_product(self, func, g0, n, lams, outs):
        """outs[j][:n] = rows g0 .. g0 + n (global sensor index) of operator `func`, over all voxels, through the covariance blocks
        `lams`.  Row sources (engine.operator): a resident tensor (the whole operator on one rank, the rank's row shard otherwise), windows
        of the stencil table (lattice feed of the radix-2 forward kernel) or a streamed operator (rows generated a transform batch
        at a time, never resident)."""
        sp = self._spectral
        kind, src, first = self._rowsrc[func]
        if kind == "tensor":
            return sp.product(src[g0 - first:g0 - first + n], n, lams, outs)
        if kind == "lattice":
            return sp.product(src.rows(g0), n, lams, outs)
        buf = self._op_rows_buffer()
        for b0 in range(0, n, sp.R):
            nb = min(sp.R, n - b0)
            sp.product(src.rows_into(buf, g0 + b0, nb), nb, lams, [o[b0:] for o in outs])

    def _rows_ok(self, A_g, A_m):
        """This step can run in the row form: both operators were built for it (lattice survey, even stencils: engine.operator)."""
        if not self.rows_static or self._rows_denied or self._gram is None or not self._gram.edge_supported():
            return False
        for f, A in (("grav", A_g), ("magn", A_m)):
            lam = self._lam.get(f)
            if lam is None or lam[0] is not A or f not in self._rowsrc or f not in self._Aedge:
                return False
        return True

    def _rows_agree(self, flag):
        """Every rank must take the same form (a rank-divergent decision would deadlock in the first collective): `sharding.agree`,
        one all-reduce and one read-back; the answer only changes when operators are (re)built, which is where this is called."""
        return agree(flag, self.world, self.group, self.device, force=self.force_collectives)

    # ---- A K -> AkA ------------------------------------------------------------------------------------------------------------------
    def _rows_times_AT(self, X, nrows, sp_, out):
        """out[:nrows, :Ms] = X[:nrows] . A_sp^T: interior y-slabs by the (y, x) correlation with the stencil table, the two 1e6-padded
        slabs through their x-DFT spectra."""
        gram, pl = self._gram, self.nx * self.nz
        func = ("grav", "magn")[sp_]
        gram.gram_rows(X, nrows, self._lam[func][1], out, 0, self.ny)
        for k, iy in enumerate((0, self.ny - 1)):
            gram.edge_rows(X[:, iy * pl:], nrows, self._edge_spectrum(func, k, self._Aedge[func][k]), out)

    def _rows_aka_local(self, props, sel_t, lengths, W, name, amp):
        """Row blocks of AkA this rank owns: (rows_r, 3 Ms_pad) = its gravity rows against grav | magn columns and its magnetic rows
        against magn columns, and the drill rows (every rank computes those few rows itself: cheaper than shipping them).  The rank's
        rows of A K exist a chunk at a time only."""
        sp, G, Msp = self._spectral, self.world, self.Ms_pad
        rows_r, Md = self.Ms // G, 0 if sel_t is None else sel_t.numel()
        g_first = self.rank * rows_r
        loc = self._workspace("aka_rows_local", (rows_r, 3 * Msp))
        loc.zero_()
        ck = min(self._rows_chunk(2), (rows_r + sp.R - 1) // sp.R * sp.R)
        X = [self._workspace2d("rows_ak_%d" % i, ck, self.N_pad) for i in range(2)]
        pre = self._fullrows if (self._fullrows and not self._rowpath) else None
        for s_, func, blocks in ((0, "grav", ((0, 0), (1, 1))), (1, "magn", ((1, 2),))):       # (column block j, slot k of loc)
            if pre is not None:
                # column form with the row exchange (GEOBO_POSTERIOR=dense on a lattice survey, >= 4 ranks): the rank's rows of A K were
                # kept whole from its send buffers (engine._keep_full_rows)
                for j, k in blocks:
                    self._timed("aka_lattice", self._gram.flops(rows_r, self.ny),
                                lambda: self._rows_times_AT(pre[(s_, j)][:, :self.N], rows_r, j, loc[:, k * Msp:(k + 1) * Msp]))
                continue
            for j in props:
                gen = sp.eigenvalues(self._cov_table(hip.kernel_id(name, s_ != j), lengths[j], lengths[s_], W[s_][j], amp))
                self._gens[(s_, j)] = gen            # (the transposed posterior applies the same blocks to L^-1 A_s)
            lams = [self._gens[(s_, j)] for j, _ in blocks]
            for c0 in range(0, rows_r, ck):
                n = min(ck, rows_r - c0)
                outs = [X[i][:n] for i in range(len(blocks))]
                self._timed("spectral_product", sp.flops(n, len(lams), self.ny),
                            lambda: self._rows_product(func, g_first + c0, n, lams, outs), valu=sp.flops_valu(n, len(lams)))
                for i, (j, k) in enumerate(blocks):
                    self._timed("aka_lattice", self._gram.flops(n, self.ny),
                                lambda: self._rows_times_AT(X[i][:, :self.N], n, j, loc[c0:, k * Msp:(k + 1) * Msp]))
        drill = None
        if Md:
            Mdp = (Md + 127) // 128 * 128
            drill = self._workspace("aka_rows_drill", (Mdp, 2 * Msp))
            drill.zero_()
            Xd = self._workspace2d("fullrows_drill", Mdp, self.N_pad)

            def drill_rows():
                for sp_ in (0, 1):
                    Xd.zero_()
                    self._cov_rows(name, 2, sp_, lengths, W, amp, sel_t, 0, Xd[:Md, :self.N])
                    self._rows_times_AT(Xd[:, :self.N], Md, sp_, drill[:, sp_ * Msp:(sp_ + 1) * Msp])
            self._timed("aka_lattice", self._gram.flops(2 * Md, self.ny), drill_rows)
        return loc, drill

    def _rows_aka_place(self, AkA, allrows, drill, sel_t):
        """The gathered row blocks (world, rows_r, 3 Ms_pad) and the replicated drill rows into the lower triangle of AkA."""
        rows_r, Md, off_d, Msp = self.Ms // self.world, 0 if sel_t is None else sel_t.numel(), 2 * self.Ms_pad, self.Ms_pad
        for src in range(self.world):
            r0 = src * rows_r
            AkA[r0:r0 + rows_r, :off_d].copy_(allrows[src][:, :off_d])
            AkA[Msp + r0:Msp + r0 + rows_r, Msp:off_d].copy_(allrows[src][:, off_d:])
        AkA[Msp:off_d, :Msp] = AkA[:Msp, Msp:off_d].t()
        if Md:
            AkA[off_d:off_d + Md, :off_d].copy_(drill[:Md])
        return AkA

    def _assemble_AkA_rows(self, AkA, M_pad, sel_t, lengths, name, amp, gp_sigma, props):
        """AkA from row blocks: local correlation of this rank's sensor rows, one all-gather."""
        loc, drill = self._rows_aka_local(props, sel_t, lengths, self._W, name, amp)
        allrows = self._timed("xgmi_all_gather", 0.0, lambda: gather_rows(loc, self.world, self.group, force=self.force_collectives))
        self._rows_aka_place(AkA, allrows, drill, sel_t)
        return self._finish_AkA(AkA, M_pad, sel_t, lengths, name, amp, gp_sigma)

    # ---- posterior ---------------------------------------------------------------------------------------------------------------
    def _posterior_rows(self, Linv, u, sel_t, lengths, W, name, amp, props, M_pad):
        """The transposed posterior (engine._posterior_zpath) sharded by ROWS of L^-1 over the ranks: rank r carries the rows of its own
        Ms / G gravity and Ms / G magnetic sensors (2 + 1 row blocks of Z = L^-1 A, a chunk at a time) and a 1/G share of the drill
        rows through the covariance product; the partial sums of squares meet in ONE all-reduce of P_c N doubles (4 MB at 64^3); the
        mean is three rows through the covariance product and every rank forms it whole (_mean_rows).
        Returns (mu, var), (P_c, N) each, complete on every rank."""
        sp, N, Msp, P_c, Md = self._spectral, self.N, self.Ms_pad, len(props), 0 if sel_t is None else sel_t.numel()
        nx, ny, nz, G, r = self.nx, self.ny, self.nz, self.world, self.rank
        rows_r = self.Ms // G
        a0, a1 = r * rows_r, Msp + r * rows_r
        ssq = self._workspace("rows_reduce", (P_c, N))
        cws = self._workspace("colgemv_ws", (hip.colgemv_ws_doubles(M_pad, M_pad),))
        Eg, Em = self._Aedge["grav"], self._Aedge["magn"]
        edge_of = lambda func: Eg if func == "grav" else Em

        def mean():
            # every rank forms the whole mean itself (three rows through the covariance product: cheaper than an all-reduce of it)
            w = hip.colgemv(Linv, u, ws=cws)                                   # L^-T u
            return self._mean_rows(w, sel_t, lengths, W, name, amp, props,
                                   lambda func, wv, out: self._lattice_Z(wv.view(1, -1), 1, func, None, out, edge=edge_of(func)))
        mu = self._timed("posterior_mean", 0.0, mean)
        gram = self._gram
        zx = gram.zx_supported() and sp.fused_ss()           # rows of Z as [iy][iz][ix]: what the fused (row, z)-plane inverse writes
        zc = min(self._rows_chunk(2), (rows_r + sp.R - 1) // sp.R * sp.R)
        Zg, Zm = self._workspace2d("Zg", zc, N), self._workspace2d("Zm", zc, N)
        slots = sp.ss_slots()
        ss = [self._workspace("post_ss_%d" % jj, (slots, ny, nx * nz)) for jj in range(P_c)]
        for t in ss:
            t.zero_()
        gens_g, gens_m = [self._gens[(0, j)] for j in props], [self._gens[(1, j)] for j in props]
        swap = (lambda g: g.view(ny, sp.Px, sp.Pz).transpose(1, 2).contiguous().view(-1)) if zx else (lambda g: g)
        tg, tm = [swap(g) for g in gens_g], [swap(g) for g in gens_m]
        # blocks (0, 1) and (1, 0) of a symmetric prior coincide: three y-stage products per two-term row instead of four
        y2s = sp.y2s_tables(tg, tm) if tuple(props[:2]) == (0, 1) else None
        fl_z = gram.flops(1, ny) + 2 * 3 * 2.0 * 128 * 128 * 64
        for two in (False, True):
            for c0 in range(0, rows_r, zc):
                n = min(zc, rows_r - c0)
                b0 = (a1 if two else a0) + c0

                def zlattice():
                    self._lattice_Z(Linv[b0:b0 + n, :Msp], n, "grav", None, Zg, zx=zx, edge=Eg)
                    if two:
                        self._lattice_Z(Linv[b0:b0 + n, Msp:2 * Msp], n, "magn", None, Zm, zx=zx, edge=Em)
                self._timed("posterior_zlattice", (2 if two else 1) * n * fl_z, zlattice)
                shared = y2s is not None and P_c == 2
                self._timed("posterior_spectral", sp.flops_ss(0 if two else n, n if two else 0, P_c, shared),
                            lambda: sp.reduce_ss(Zg, n, tg, Zm if two else None, 0, tm, ss, y2s=y2s),
                            valu=sp.valu_ss(0 if two else n, n if two else 0, P_c, shared))
        for jj, t in enumerate(ss):
            if zx:
                ssq[jj].copy_(t.sum(0).view(ny, nz, nx).transpose(1, 2).reshape(-1))
            else:
                ssq[jj].copy_(t.sum(0).reshape(-1))
        # the rows behind the sensor rows (only they see the drill columns of L^-1): an equal share per rank
        dper = -(-Md // G)
        d0 = min(Md, r * dper)
        nd = min(Md, d0 + dper) - d0
        if nd:
            def drill_rows():
                part = self._drill_rows_ss(Linv, d0, nd, sel_t, lengths, W, name, amp, props, gens_g, gens_m,
                                           lambda Lv, n, func, out: self._lattice_Z(Lv, n, func, None, out, edge=edge_of(func)),
                                           None, None)
                for jj in range(P_c):
                    ssq[jj].add_(part[jj])
            self._timed("posterior_drill_rows", 0.0, drill_rows)
        self._timed("xgmi_all_reduce", 0.0, lambda: allreduce_sum_(ssq, G, self.group, force=self.force_collectives))
        return mu, amp * 1.0 - ssq
