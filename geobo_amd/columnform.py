"""Column form with the row exchange (plan.Route.family "columns", exchange = True; rounds 1-2): mixin of engine.PosteriorEngine.

Every rank transforms its own Ms / G sensor rows of both operators for ALL voxels and one all-to-all per operator hands each peer the
block columns of A K it owns; the chunked variant serves fp32 assembly / streamed operators when the row form
(rowform.py) is not available for the survey.  Reference arithmetic: inversion.py:96,114 (np.dot(Asens3, kcov))."""
import torch

from . import hip
from .operators import StreamedOperator
from .sharding import backend_of, exchange_blocks, exchange_blocks_finish, exchange_blocks_start, shard_columns

F64 = hip.F64


class ColumnExchangeMixin:
    def _assemble_AK_spectral_exchange(self, AK, lengths, W, name, amp, props):
        """Row-sharded spectral product + all-to-all (multi-GPU): rank r transforms sensor rows [r*Ms/G, (r+1)*Ms/G) of both
        operators for every voxel, cropping the backward passes once per destination y-slab straight into the send buffer of
        the operator, [dest][block][row][col]; one all_to_all_single over xGMI per operator (the first one runs under the second
        operator's transforms); the received blocks are this rank's columns of every sensor row."""
        if self.f32 or self.streamed:
            return self._exchange_chunked(AK, lengths, W, name, amp, props)
        self._finish_exchange()          # (an exchange left over by a call that failed between its start and its factorisation)
        # one exchange per operator: the gravity rows travel while the magnetic rows are being transformed
        sends, pending = [], []
        for s_, func in ((0, "grav"), (1, "magn")):
            send = self._exchange_send(s_, func, lengths, W, name, amp, props)
            sends.append(send)
            out = self._workspace("xchg_recv_%d" % s_, tuple(send.shape)) if backend_of(self._xgroup) == "nccl" else None
            if self.async_exchange:
                pending.append(exchange_blocks_start(send, self.world, self._xgroup, out=out))
            else:
                pending.append((self._timed("xgmi_all_to_all", 0.0, lambda: exchange_blocks(send, self.world, self._xgroup)), None))
        self._keep_full_rows(sends, props)
        self._pending_exchange = (AK, pending, props)
        if not self._row_gram():
            self._finish_exchange()      # AkA by the GEMM reads the received columns of A K

    def _finish_exchange(self):
        """Wait for the row exchange and put the received blocks into A K.  With the row-sharded lattice Gram nothing reads those
        columns before the posterior reduction (AkA comes from this rank's own rows, kept from the send buffers), so posterior()
        calls this after the factorisation: the all-to-all runs under the Gram, the all-gather and the Cholesky."""
        if self._pending_exchange is None:
            return
        AK, pending, props = self._pending_exchange
        self._pending_exchange = None
        for s_, (recv, work) in enumerate(pending):
            self._timed("xgmi_all_to_all", 0.0, lambda: exchange_blocks_finish(work))
            self._exchange_place(AK, recv, props, s_)

    def _exchange_chunked(self, AK, lengths, W, name, amp, props):
        """The row exchange for the large-cube modes (fp32 assembly and / or streamed operators, BASELINE config 5): the rank's
        sensor rows go through the transform in chunks of a few row batches; each chunk is cropped per destination into a send
        buffer of ~1.5 GB (fp32 in the fp32 mode: converted from an fp64 scratch of one batch), exchanged by its own
        all_to_all_single and written straight into the A K shard -- no rank-sized send / receive buffers (3 x 104 GB at 128^3)."""
        sp, nc, G = self._spectral, self.nc, self.world
        plane = self.nx * self.nz
        rows_r, P_c, Rb = self.Ms // G, len(props), self._spectral.R
        esize = 4 if self.f32 else 8
        Rc = Rb * max(1, int((3 << 29) // (G * P_c * Rb * nc * esize)))
        send = self._workspace("xchg_send_chunk", (G, P_c, Rc, nc), dtype=hip.F32 if self.f32 else F64)
        scr = self._workspace("xchg_scratch64", (G, P_c, Rb, nc)) if self.f32 else None
        slabs_of = [tuple(c // plane for c in shard_columns(self.N_pad, G, d)) for d in range(G)]
        self._fullrows = {}
        for s_, func in ((0, "grav"), (1, "magn")):
            lams = [sp.eigenvalues(self._cov_table(hip.kernel_id(name, s_ != j), lengths[j], lengths[s_], W[s_][j], amp)) for j in props]
            A = [v for k, v in self._A.items() if k[0] == func][0]
            streamed = isinstance(A, StreamedOperator)
            abuf = self._op_rows_buffer() if streamed else None
            for r0 in range(0, rows_r, Rc):
                R = min(Rc, rows_r - r0)

                def transform():
                    for rb in range(0, R, Rb):
                        n = min(Rb, R - rb)
                        g0 = self.rank * rows_r + r0 + rb                 # first sensor row of this batch
                        src = A.rows_into(abuf, g0, n) if streamed else self._Arows[func][r0 + rb:r0 + rb + n]
                        dst = scr if self.f32 else send[:, :, rb:rb + Rb]
                        sp.product(src, n, lams, None, slabs=[(slabs_of[d][0], slabs_of[d][1], [dst[d, jj] for jj in range(P_c)])
                                                              for d in range(G)])
                        if self.f32:
                            for d in range(G):
                                for jj in range(P_c):
                                    hip.convert(scr[d, jj, :n], send[d, jj, rb:rb + n])
                self._timed("spectral_product", sp.flops(R, P_c, self.ny), transform, valu=sp.flops_valu(R, P_c))
                recv = self._timed("xgmi_all_to_all", 0.0, lambda: exchange_blocks(send.view(G, -1), self.world, self.group))
                for srcr in range(G):
                    blk = recv[srcr].view(P_c, Rc, nc)
                    a0 = s_ * self.Ms_pad + srcr * rows_r + r0
                    for jj in range(P_c):
                        AK[a0:a0 + R, jj * nc:(jj + 1) * nc].copy_(blk[jj, :R])

    def _row_gram(self):
        """True when AkA is assembled from row blocks: row exchange + lattice Gram available for both operators."""
        return self.exchange and all(self._lam.get(f) is not None for f in ("grav", "magn"))

    def _keep_full_rows(self, sends, props):
        """This rank's own sensor rows of A K over ALL voxels (block columns 0 and 1), gathered from the per-destination slabs of
        the two send buffers: the input of the row-sharded lattice Gram."""
        self._fullrows = {}
        if not self._row_gram():
            return
        G, nc = self.world, self.nc
        rows_r, P_c = self.Ms // G, len(props)
        for s_ in (0, 1):
            v = sends[s_].view(G, P_c, rows_r, nc)
            for sp_ in (0, 1):
                full = self._workspace2d("fullrows_%d%d" % (s_, sp_), rows_r, G * nc)
                for d in range(G):
                    full[:, d * nc:(d + 1) * nc].copy_(v[d, props.index(sp_)])
                self._fullrows[(s_, sp_)] = full

    def _exchange_send(self, s_, func, lengths, W, name, amp, props):
        """Send buffer of one operator, (G, P_c * rows_r * nc): [destination][block][row][col] -- this rank's sensor rows of A_s K,
        every voxel, cropped per destination y-slab by the backward passes themselves."""
        sp, nc, G = self._spectral, self.nc, self.world
        plane = self.nx * self.nz
        rows_r, P_c = self.Ms // G, len(props)
        send = self._workspace("xchg_send_%d" % s_, (G, P_c * rows_r * nc))
        slabs_of = [tuple(c // plane for c in shard_columns(self.N_pad, G, d)) for d in range(G)]
        lams = []
        for j in props:
            tab = self._cov_table(hip.kernel_id(name, s_ != j), lengths[j], lengths[s_], W[s_][j], amp)
            lams.append(sp.eigenvalues(tab))
        slabs = [(slabs_of[d][0], slabs_of[d][1], [send[d].view(P_c, rows_r, nc)[jj] for jj in range(P_c)]) for d in range(G)]
        Ar = self._Arows[func]
        self._timed("spectral_product", sp.flops(rows_r, P_c, self.ny), lambda: sp.product(Ar, rows_r, lams, None, slabs=slabs),
                    valu=sp.flops_valu(rows_r, P_c))
        return send

    def _exchange_place(self, AK, recv, props, s_):
        G, nc = self.world, self.nc
        rows_r, P_c = self.Ms // G, len(props)
        for src in range(G):
            blocks = recv[src].view(P_c, rows_r, nc)
            r0 = s_ * self.Ms_pad + src * rows_r
            for jj in range(P_c):
                AK[r0:r0 + rows_r, jj * nc:(jj + 1) * nc].copy_(blocks[jj])
