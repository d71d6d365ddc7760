"""Soak test of the hand-synchronised kernels (LDS-DMA rings with counted vmcnt waits): many launches with varying plane counts
against torch references; any race shows up as a sporadic mismatch.

    python tools/soak_kernels.py [seed] [seconds]        # long form
    tests/test_hip_kernels.py::test_soak_hand_synchronised_kernels runs soak(iters=25) in the -m gpu tier."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def soak(seed=0, seconds=None, iters=60, verbose=True):
    import numpy as np
    import torch
    from geobo_amd import hip
    rng = np.random.default_rng(seed)
    n, P = 64, 128
    rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device="cuda") * 2 - 1
    Gx, Gz, GxT, GzT = rnd(P, n), rnd(P, n), rnd(n, P), rnd(n, P)
    from geobo_amd.spectral import folded_matrices, forward_matrix
    Gf = hip.to_dev(forward_matrix(n))
    GfT = Gf.t().contiguous()
    Ff = hip.to_dev(np.stack(folded_matrices(n), axis=2))
    G32 = hip.to_dev(forward_matrix(32))
    fe32, fo32 = folded_matrices(32)
    fq = np.zeros((64, 32, 2))
    fq[:32, :16, 0], fq[:32, :16, 1], fq[32:, 16:, 0], fq[32:, 16:, 1] = fe32, fo32, fe32, fo32
    Fq = hip.to_dev(fq)
    bad = 0
    t0 = time.time()
    it = 0
    while (time.time() - t0 < seconds) if seconds else (it < iters):
        it += 1
        rows, ppr = int(rng.integers(1, 40)), int(rng.integers(1, 200))
        # xz2d forward / inverse
        for inverse in (False, True):
            ix, iz, ox, oz = (P, P, n, n) if inverse else (n, n, P, P)
            Mx, Mz = (GxT, GzT) if inverse else (Gx, Gz)
            src = rnd(rows, ppr * ix * iz + 16)
            out = torch.empty((rows, ppr * ox * oz), dtype=torch.float64, device="cuda")
            hip.xz2d(inverse, n, n, rows, ppr, src, src.stride(0), ix * iz, Mx, Mz, out, out.stride(0), ox * oz)
            ref = torch.einsum("ai,rpik,bk->rpab", Mx, src[:, :ppr * ix * iz].reshape(rows, ppr, ix, iz), Mz)
            e = (out.reshape(rows, ppr, ox, oz) - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("xz2d inverse=%s rows=%d ppr=%d err %.3e" % (inverse, rows, ppr, e), flush=True)
            # the radix-2 kernels on the same plane counts (pair-interleaved basis)
            M = GfT if inverse else Gf
            out = torch.empty((rows, ppr * ox * oz), dtype=torch.float64, device="cuda")
            hip.xz2d_fold(inverse, n, rows, ppr, src, src.stride(0), ix * iz, Ff, Ff, out, out.stride(0), ox * oz)
            ref = torch.einsum("ai,rpik,bk->rpab", M, src[:, :ppr * ix * iz].reshape(rows, ppr, ix, iz), M)
            e = (out.reshape(rows, ppr, ox, oz) - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("xz2d_fold inverse=%s rows=%d ppr=%d err %.3e" % (inverse, rows, ppr, e), flush=True)
        # xcorr
        planes = int(rng.integers(1, 129))
        src = rnd(rows, planes * n * n)
        lam = rnd(planes * n * P)
        out = torch.empty((rows, planes * P), dtype=torch.float64, device="cuda")
        hip.xcorr_reduce(n, n, rows, planes, src, src.stride(0), n * n, Gx, lam, out, out.stride(0), P)
        Z = torch.einsum("ox,rpxz->rpzo", Gx, src.reshape(rows, planes, n, n))
        ref = (Z * lam.reshape(planes, n, P)[None]).sum(2)
        e = (out.reshape(rows, planes, P) - ref).abs().max().item() / ref.abs().max().item()
        if not e < 1e-13:
            bad += 1; print("xcorr rows=%d planes=%d err %.3e" % (rows, planes, e), flush=True)
        out = torch.empty((rows, planes * P), dtype=torch.float64, device="cuda")
        hip.xcorr_reduce_fold(n, rows, planes, src, src.stride(0), n * n, Ff, lam, out, out.stride(0), P)
        Z = torch.einsum("ox,rpxz->rpzo", Gf, src.reshape(rows, planes, n, n))
        ref = (Z * lam.reshape(planes, n, P)[None]).sum(2)
        e = (out.reshape(rows, planes, P) - ref).abs().max().item() / ref.abs().max().item()
        if not e < 1e-13:
            bad += 1; print("xcorr_fold rows=%d planes=%d err %.3e" % (rows, planes, e), flush=True)
        # toeplitz
        C, R = 64 * int(rng.integers(1, 9)), int(rng.integers(1, 30))
        srct = rnd(R, n, C)
        tabs = [rnd(n, C), rnd(n, C)]
        outs = [torch.empty((R, n, C), dtype=torch.float64, device="cuda") for _ in range(2)]
        hip.toeplitz_y(n, C, R, srct.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outs])
        idx = (torch.arange(n)[:, None] - torch.arange(n)[None, :]).abs().cuda()
        for j in range(2):
            ref = torch.einsum("ypc,rpc->ryc", tabs[j][idx], srct)
            e = (outs[j] - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("toeplitz R=%d C=%d err %.3e" % (R, C, e), flush=True)
        # single-block launch (every half of the outputs split between two waves), random output slab
        ya = int(rng.integers(0, n - 1)); yb = int(rng.integers(ya + 1, n + 1))
        out1 = torch.empty((R, yb - ya, C), dtype=torch.float64, device="cuda")
        hip.toeplitz_y(n, C, R, srct.reshape(-1), [tabs[0].reshape(-1)], [out1.reshape(-1)], ya, yb)
        ref = torch.einsum("ypc,rpc->ryc", tabs[0][idx], srct)[:, ya:yb]
        e = (out1 - ref).abs().max().item() / ref.abs().max().item()
        if not e < 1e-13:
            bad += 1; print("toeplitz single block R=%d C=%d slab [%d, %d) err %.3e" % (R, C, ya, yb, e), flush=True)
        # ---- round 3: the inverse transform's other modes on the same plane counts -------------------------------------------
        rows2, ppr2 = int(rng.integers(1, 24)), int(rng.integers(1, 96))
        s1 = rnd(rows2, ppr2 * P * P + 16)
        ref1 = torch.einsum("ai,rpik,bk->rpab", GfT, s1[:, :ppr2 * P * P].reshape(rows2, ppr2, P, P), GfT)
        # (a) strided rows: planes written as [y][p][x] instead of [p][y][x]
        outs_ = torch.empty((rows2, n * ppr2 * n), dtype=torch.float64, device="cuda")
        hip.xz2d_fold_inv_strided(n, rows2, ppr2, s1, s1.stride(0), P * P, Ff, Ff, outs_, outs_.stride(0), n, ppr2 * n)
        e = (outs_.reshape(rows2, n, ppr2, n).permute(0, 2, 1, 3) - ref1).abs().max().item() / ref1.abs().max().item()
        if not e < 1e-13:
            bad += 1; print("fold_inv_strided rows=%d ppr=%d err %.3e" % (rows2, ppr2, e), flush=True)
        # (b) product input: plane (r, p) = a[p] * b[r], formed in the kernel
        fa_, fb_ = rnd(ppr2, P * P), rnd(rows2, P * P)
        refm = torch.einsum("ai,rpik,bk->rpab", GfT, (fa_[None, :, :] * fb_[:, None, :]).reshape(rows2, ppr2, P, P), GfT)
        outm = torch.empty((rows2, n * ppr2 * n), dtype=torch.float64, device="cuda")
        hip.xz2d_fold_inv_mul(n, rows2, ppr2, fa_, P * P, fb_, P * P, Ff, Ff, outm, outm.stride(0), n, ppr2 * n)
        e = (outm.reshape(rows2, n, ppr2, n).permute(0, 2, 1, 3) - refm).abs().max().item() / refm.abs().max().item()
        if not e < 1e-13:
            bad += 1; print("fold_inv_mul rows=%d ppr=%d err %.3e" % (rows2, ppr2, e), flush=True)
        # (c) sum of squares over the rows, rows >= r2 with a second term
        s2 = rnd(rows2, ppr2 * P * P + 16)
        r2 = int(rng.integers(0, rows2 + 1))
        slots = hip.xz2d_fold_inv_ss_slots(n, rows2, ppr2)
        ss = torch.zeros((slots, ppr2, n * n), dtype=torch.float64, device="cuda")
        hip.xz2d_fold_inv_ss(n, rows2, ppr2, s1, s1.stride(0), P * P, Ff, Ff, ss, src2=s2 if r2 < rows2 else None, in2_row=s2.stride(0),
                             r2_first=r2)
        tot = ref1.clone()
        if r2 < rows2:
            tot[r2:] += torch.einsum("ai,rpik,bk->rpab", GfT, s2[:rows2 - r2, :ppr2 * P * P].reshape(rows2 - r2, ppr2, P, P), GfT)
        refs_ = (tot ** 2).sum(0).reshape(ppr2, n * n)
        e = (ss.sum(0) - refs_).abs().max().item() / refs_.abs().max().item()
        if not e < 1e-12:
            bad += 1; print("fold_inv_ss rows=%d ppr=%d r2=%d err %.3e" % (rows2, ppr2, r2, e), flush=True)
        # ---- round 4: the two-term y stage (eight waves, partial sums exchanged through the consumed LDS stage) ---------------------
        nyt = int(rng.choice([32, 48, 64]))
        Ct, Rt = 64 * int(rng.integers(1, 7)), int(rng.integers(1, 40))
        sg_, sm_ = rnd(Rt, nyt, Ct), rnd(Rt, nyt, Ct)
        tg_, tm_ = [rnd(nyt, Ct), rnd(nyt, Ct)], [rnd(nyt, Ct), rnd(nyt, Ct)]
        o2 = [torch.empty((Rt, nyt, Ct), dtype=torch.float64, device="cuda") for _ in range(2)]
        hip.toeplitz_y2t(nyt, Ct, Rt, sg_.reshape(-1), sm_.reshape(-1), [t.reshape(-1) for t in tg_], [t.reshape(-1) for t in tm_],
                         [o.reshape(-1) for o in o2])
        idt = (torch.arange(nyt)[:, None] - torch.arange(nyt)[None, :]).abs().cuda()
        for j in range(2):
            ref = torch.einsum("ypc,rpc->ryc", tg_[j][idt], sg_) + torch.einsum("ypc,rpc->ryc", tm_[j][idt], sm_)
            e = (o2[j] - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("toeplitz_y2t ny=%d R=%d C=%d err %.3e" % (nyt, Rt, Ct, e), flush=True)
        # ---- round 5: the same rows with the shared cross block (three products; waves of two sizes, dedicated exchange area) --------
        o3 = [torch.full((Rt, nyt, Ct), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
        hip.toeplitz_y2s(nyt, Ct, Rt, sg_.reshape(-1), sm_.reshape(-1), (tg_[0] - tg_[1]).reshape(-1), tg_[1].reshape(-1),
                         (tm_[1] - tg_[1]).reshape(-1), [o.reshape(-1) for o in o3])
        for j, (ta, tb) in enumerate(((tg_[0], tg_[1]), (tg_[1], tm_[1]))):
            ref = torch.einsum("ypc,rpc->ryc", ta[idt], sg_) + torch.einsum("ypc,rpc->ryc", tb[idt], sm_)
            e = (o3[j] - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("toeplitz_y2s ny=%d R=%d C=%d err %.3e" % (nyt, Rt, Ct, e), flush=True)
        # ---- round 4: 32 x 32 planes four at a time on the n = 64 radix-2 kernels ---------------------------------------------------
        rq, gq = int(rng.integers(1, 30)), int(rng.integers(1, 12))
        for inverse in (False, True):
            iq, oq = (64, 32) if inverse else (32, 64)
            srcq = rnd(rq, 4 * gq * iq * iq + 16)
            outq = torch.empty((rq, 4 * gq * oq * oq), dtype=torch.float64, device="cuda")
            hip.xz2d_fold_quad(inverse, 32, rq, gq, srcq, srcq.stride(0), iq * iq, Fq, Fq, outq, outq.stride(0), oq * oq)
            Mq = G32.t().contiguous() if inverse else G32
            ref = torch.einsum("ai,rpik,bk->rpab", Mq, srcq[:, :4 * gq * iq * iq].reshape(rq, 4 * gq, iq, iq), Mq)
            e = (outq.reshape(rq, 4 * gq, oq, oq) - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("xz2d_fold_quad inverse=%s rows=%d groups=%d err %.3e" % (inverse, rq, gq, e), flush=True)
        # ---- round 4: the windowed y stage (ny > 64) with several output chunks per workgroup, slabs ending inside a chunk, three blocks
        #      as 2 + 1, and the accumulating form (second term of a two-term row) --------------------------------------------------
        nyw = int(rng.choice([80, 96, 112, 128]))
        Cw, Rw, npw = 64 * int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(1, 4))
        ya = int(rng.integers(0, nyw - 1)); yb = int(rng.integers(ya + 1, nyw + 1))
        if rng.integers(0, 2):
            ya, yb = 0, nyw
        sw, sw2 = rnd(Rw, nyw, Cw), rnd(Rw, nyw, Cw)
        tw, tw2 = [rnd(nyw, Cw) for _ in range(npw)], [rnd(nyw, Cw) for _ in range(npw)]
        ow = [torch.full((Rw, yb - ya, Cw), float("nan"), dtype=torch.float64, device="cuda") for _ in range(npw)]
        flat = lambda ts: [t.reshape(-1) for t in ts]
        hip.toeplitz_y(nyw, Cw, Rw, sw.reshape(-1), flat(tw), flat(ow), ya, yb)
        hip.toeplitz_y(nyw, Cw, Rw, sw2.reshape(-1), flat(tw2), flat(ow), ya, yb, accumulate=True)
        idw = (torch.arange(nyw)[:, None] - torch.arange(nyw)[None, :]).abs().cuda()
        for j in range(npw):
            ref = (torch.einsum("ypc,rpc->ryc", tw[j][idw], sw) + torch.einsum("ypc,rpc->ryc", tw2[j][idw], sw2))[:, ya:yb]
            e = (ow[j] - ref).abs().max().item() / ref.abs().max().item()
            if not e < 1e-13:
                bad += 1; print("toeplitz windowed ny=%d R=%d C=%d blocks=%d slab [%d, %d) err %.3e" % (nyw, Rw, Cw, npw, ya, yb, e), flush=True)
        # ---- round 6: the y stage on the matrix pipe.  ny <= 64: a wave per 16 modes, register-prefetched rows (no LDS hand-off between
        #      waves); ny > 64: eight waves per tile in two roles (matrix waves / orbit waves), double-buffered LDS exchanges ordered by ONE barrier per (row, block) step -- a missing barrier or an
        #      exchange area rewritten too early shows up as a sporadic mismatch against the direct kernels / torch, NaN-poisoned outputs
        for nys in (int(rng.choice([32, 48, 64])), nyw):
            Cs, Rs, nps = 16 * int(rng.integers(1, 24)), int(rng.integers(1, 12)), int(rng.integers(1, 4))
            ss1, ss2 = rnd(Rs, nys, Cs), rnd(Rs, nys, Cs)
            ts1, ts2 = [rnd(nys, Cs) for _ in range(nps)], [rnd(nys, Cs) for _ in range(nps)]
            ids = (torch.arange(nys)[:, None] - torch.arange(nys)[None, :]).abs().cuda()
            r1 = [torch.einsum("ypc,rpc->ryc", ts1[j][ids], ss1) for j in range(nps)]
            os_ = [torch.full((Rs, nys, Cs), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nps)]
            hip.spectral_y(nys, Cs, Rs, ss1.reshape(-1), flat(ts1), flat(os_))
            for j in range(nps):
                e = (os_[j] - r1[j]).abs().max().item() / r1[j].abs().max().item()
                if not e < 1e-13:
                    bad += 1; print("spectral_y ny=%d R=%d C=%d blocks=%d err %.3e" % (nys, Rs, Cs, nps, e), flush=True)
            r2 = [r1[j] + torch.einsum("ypc,rpc->ryc", ts2[j][ids], ss2) for j in range(nps)]
            if nys > 64:
                ot = [torch.full((Rs, nys, Cs), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nps)]
                hip.spectral_y3t(nys, Cs, Rs, ss1.reshape(-1), ss2.reshape(-1), flat(ts1), flat(ts2), flat(ot))
                hip.spectral_y(nys, Cs, Rs, ss2.reshape(-1), flat(ts2), flat(os_), accumulate=True)
                for j in range(nps):
                    e = max((ot[j] - r2[j]).abs().max().item(), (os_[j] - r2[j]).abs().max().item()) / r2[j].abs().max().item()
                    if not e < 1e-13:
                        bad += 1; print("spectral_y3t / accumulate ny=%d R=%d C=%d blocks=%d err %.3e" % (nys, Rs, Cs, nps, e), flush=True)
            else:
                t00, t01, t11 = rnd(nys, Cs), rnd(nys, Cs), rnd(nys, Cs)
                o2 = [torch.full((Rs, nys, Cs), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
                hip.spectral_y2s(nys, Cs, Rs, ss1.reshape(-1), ss2.reshape(-1), (t00 - t01).reshape(-1), t01.reshape(-1), (t11 - t01).reshape(-1), flat(o2))
                want = [torch.einsum("ypc,rpc->ryc", t00[ids], ss1) + torch.einsum("ypc,rpc->ryc", t01[ids], ss2),
                        torch.einsum("ypc,rpc->ryc", t01[ids], ss1) + torch.einsum("ypc,rpc->ryc", t11[ids], ss2)]
                for j in range(2):
                    e = (o2[j] - want[j]).abs().max().item() / want[j].abs().max().item()
                    if not e < 1e-13:
                        bad += 1; print("spectral_y2s ny=%d R=%d C=%d err %.3e" % (nys, Rs, Cs, e), flush=True)
        # ---- round 5: the persistent tile-DAG factorisation (geobo_potrf_inv from m = 1024): inter-workgroup hand-offs through
        #      agent-scope counters.  Random block counts, the result buffers poisoned with NaN (a tile or a zero that is read before
        #      its producer's write-through stores have landed shows up as NaN), every other iteration under uneven load: a long
        #      GEMM on a second stream takes CUs away while the tasks are drawn ------------------------------------------------------
        nbd = int(rng.integers(8, 25)) if it % 8 else int(rng.integers(41, 45))     # (round 6: <= 40 block columns run the chain walker)
        md = 128 * nbd
        Bd = rnd(md, 256)
        Sd = Bd @ Bd.t() / 256 + 0.3 * torch.eye(md, dtype=torch.float64, device="cuda")
        Lref = torch.linalg.cholesky(Sd)
        Ld = Sd.clone()
        Xd = torch.full((md, md), float("nan"), dtype=torch.float64, device="cuda")
        if it % 2:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                big = rnd(4096, 4096)
                for _ in range(3):
                    big = (big @ big) * 1e-3
        _, info = hip.potrf_inv(Ld, Xd)
        if it % 2:
            torch.cuda.current_stream().wait_stream(side)
        eL = (torch.tril(Ld) - Lref).abs().max().item()
        eX = (Xd @ Lref - torch.eye(md, dtype=torch.float64, device="cuda")).abs().max().item()
        up = torch.triu(Xd, 1).abs().max().item()
        if not (int(info.item()) == 0 and eL < 1e-12 and eX < 1e-10 and up == 0.0):
            bad += 1; print("potrf_inv (tile DAG) nb=%d loaded=%d info=%d errs %.3e %.3e upper %.1e" % (nbd, it % 2, int(info.item()), eL, eX, up), flush=True)
    if verbose:
        print("soak: %d iterations, %d mismatches" % (it, bad))
    return it, bad


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else None)
