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
    if verbose:
        print("soak: %d iterations, %d mismatches" % (it, bad))
    return it, bad


if __name__ == "__main__":
    soak(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else None)
