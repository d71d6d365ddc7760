"""One launch of each kernel of the spectral route / lattice Gram at the 64^3 batch shapes (for rocprofv3 PMC passes):
xz2d forward and inverse (256 sensor rows x 64 y-planes), toeplitz_y (256 rows, 2 property blocks), xcorr (256 rows x 128 y-modes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
n, R = 64, 256
P = 2 * n
dev = "cuda"
rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device=dev) * 2 - 1
def timed(name, flop, gbytes, f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    print("%s: %.4f s, %.1f TF/s executed, flop %.0f; algorithmic bytes %.3e (%.2f TB/s)" % (name, t, flop / t / 1e12, flop, gbytes, gbytes / t / 1e12), flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
Gx, Gz, GxT, GzT = rnd(P, n), rnd(P, n), rnd(n, P), rnd(n, P)
if which in ("all", "xz2d_fwd"):
    src, out = rnd(R, n * n * n), torch.empty((R, n * P * P), dtype=torch.float64, device=dev)
    timed("xz2d_fwd", R * n * 2.0 * (n * n * P + P * n * P), R * n * (n * n + P * P) * 8.0,
          lambda: hip.xz2d(False, n, n, R, n, src, src.stride(0), n * n, Gx, Gz, out, out.stride(0), P * P))
    del src, out
if which in ("all", "xz2d_bwd"):
    src, out = rnd(R, n * P * P), torch.empty((R, n * n * n), dtype=torch.float64, device=dev)
    timed("xz2d_bwd", R * n * 2.0 * (P * P * n + n * P * n), R * n * (n * n + P * P) * 8.0,
          lambda: hip.xz2d(True, n, n, R, n, src, src.stride(0), P * P, GxT, GzT, out, out.stride(0), n * n))
    del src, out
if which in ("all", "toeplitz"):
    C = P * P
    src = rnd(R * n * C)
    tabs = [rnd(n * C), rnd(n * C)]
    outs = [torch.empty(R * n * C, dtype=torch.float64, device=dev) for _ in range(2)]
    timed("toeplitz_y", R * C * 2.0 * n * n * 2, R * n * C * 8.0 * 3, lambda: hip.toeplitz_y(n, C, R, src, tabs, outs))
    del src, tabs, outs
if which in ("all", "toeplitz2t"):
    C = P * P
    src_g, src_m = rnd(R * n * C), rnd(R * n * C)
    tg, tm = [rnd(n * C), rnd(n * C)], [rnd(n * C), rnd(n * C)]
    outs = [torch.empty(R * n * C, dtype=torch.float64, device=dev) for _ in range(2)]
    timed("toeplitz_y2t", R * C * 2.0 * n * n * 4, R * n * C * 8.0 * 4, lambda: hip.toeplitz_y2t(n, C, R, src_g, src_m, tg, tm, outs))
    del src_g, src_m, tg, tm, outs
if which in ("all", "toeplitz2s"):
    C = P * P
    src_g, src_m = rnd(R * n * C), rnd(R * n * C)
    t3 = [rnd(n * C) for _ in range(3)]
    outs = [torch.empty(R * n * C, dtype=torch.float64, device=dev) for _ in range(2)]
    timed("toeplitz_y2s", R * C * 2.0 * n * n * 3, R * n * C * 8.0 * 4, lambda: hip.toeplitz_y2s(n, C, R, src_g, src_m, t3[0], t3[1], t3[2], outs))
    del src_g, src_m, t3, outs
if which in ("all", "spectral_y", "spectral_y1", "spectral_y2s"):
    # the y stage through its own spectrum on the matrix pipe (geobo_spectral_y*): executed flop = ny^2 / 2 multiply-adds per transform
    # and mode on MFMA + the orbit butterflies (adds count 1, multiplies 1, fused multiply-adds 2)
    C = P * P
    src_g, src_m = rnd(R * n * C), rnd(R * n * C)
    t3 = [rnd(n * C) for _ in range(3)]
    outs = [torch.empty(R * n * C, dtype=torch.float64, device=dev) for _ in range(2)]
    if which in ("all", "spectral_y"):
        timed("spectral_y", R * C * (3.0 * n * n + 18.0 * n), R * n * C * 8.0 * 3, lambda: hip.spectral_y(n, C, R, src_g, t3[:2], outs))
    if which in ("all", "spectral_y1"):
        timed("spectral_y1", R * C * (2.0 * n * n + 11.0 * n), R * n * C * 8.0 * 2, lambda: hip.spectral_y(n, C, R, src_g, t3[:1], outs[:1]))
    if which in ("all", "spectral_y2s"):
        timed("spectral_y2s", R * C * (4.0 * n * n + 28.0 * n), R * n * C * 8.0 * 4, lambda: hip.spectral_y2s(n, C, R, src_g, src_m, t3[0], t3[1], t3[2], outs))
    del src_g, src_m, t3, outs
if which in ("spectral_y128", "spectral_y128_1", "spectral_y3t128"):
    # the long-axis form (four waves per 16-mode tile) at the 128^3 batch shape: 32 rows of 128 planes x 65536 modes
    ny_, Rb, C = 128, 32, 4 * 128 * 128
    src_g, src_m = rnd(Rb * ny_ * C), rnd(Rb * ny_ * C)
    t6 = [rnd(ny_ * C) for _ in range(6)]
    outs = [torch.empty(Rb * ny_ * C, dtype=torch.float64, device=dev) for _ in range(3)]
    if which == "spectral_y128":
        timed("spectral_y128", Rb * C * (4.0 * ny_ * ny_ + 25.0 * ny_), Rb * ny_ * C * 8.0 * 4, lambda: hip.spectral_y(ny_, C, Rb, src_g, t6[:3], outs))
    elif which == "spectral_y128_1":
        timed("spectral_y128_1", Rb * C * (2.0 * ny_ * ny_ + 11.0 * ny_), Rb * ny_ * C * 8.0 * 2, lambda: hip.spectral_y(ny_, C, Rb, src_g, t6[:1], outs[:1]))
    else:
        timed("spectral_y3t128", Rb * C * (5.0 * ny_ * ny_ + 38.0 * ny_), Rb * ny_ * C * 8.0 * 5, lambda: hip.spectral_y3t(ny_, C, Rb, src_g, src_m, t6[:3], t6[3:], outs))
    del src_g, src_m, t6, outs
if which in ("axis128_fwd", "axis128_inv"):
    # radix-4 axis passes (geobo_spectral_axis) at the x-pass shape of a 128^3 batch: 16 rows x 128 y-planes = 2048 items of
    # 128 <-> 256 planes x 256 contiguous modes
    n_, Pz_, items = 128, 256, 16 * 128
    lo, hi = rnd(items * n_ * Pz_), rnd(items * 2 * n_ * Pz_)
    if which == "axis128_fwd":
        timed("axis128_fwd", items * Pz_ * 2.0 * n_ * n_, items * Pz_ * 8.0 * 3 * n_, lambda: hip.spectral_axis(False, n_, Pz_, Pz_, Pz_, n_ * Pz_, 2 * n_ * Pz_, items, lo, hi))
    else:
        timed("axis128_inv", items * Pz_ * 2.0 * n_ * n_, items * Pz_ * 8.0 * 3 * n_, lambda: hip.spectral_axis(True, n_, Pz_, Pz_, Pz_, 2 * n_ * Pz_, n_ * Pz_, items, hi, lo))
    del lo, hi
if which in ("all", "xcorr"):
    src = rnd(R, P * n * n)
    lam = rnd(P * n * P)
    out = torch.empty((R, P * P), dtype=torch.float64, device=dev)
    timed("xcorr", R * P * 2.0 * P * n * n, R * P * (n * n + P) * 8.0,
          lambda: hip.xcorr_reduce(n, n, R, P, src, src.stride(0), n * n, Gx, lam, out, out.stride(0), P))
# ---- radix-2 / radix-4 kernels: flop = what the folded kernels execute on the matrix pipe (forward: half of the plain product along z,
# a quarter along x; inverse and xcorr_fold: a quarter) -----------------------------------------------------------------
if which in ("all", "fold_fwd", "fold_bwd", "xcorr_fold", "fold_inv_ss", "fold_inv_strided", "fold_inv_mul"):
    import numpy as np
    from geobo_amd.spectral import folded_matrices
    F = hip.to_dev(np.stack(folded_matrices(n), axis=2))
if which in ("all", "fold_fwd"):
    src, out = rnd(R, n * n * n), torch.empty((R, n * P * P), dtype=torch.float64, device=dev)
    timed("xz2d_fold_fwd", R * n * (1.0 * n * n * P + 0.5 * P * n * P), R * n * (n * n + P * P) * 8.0,
          lambda: hip.xz2d_fold(False, n, R, n, src, src.stride(0), n * n, F, F, out, out.stride(0), P * P))
    del src, out
if which in ("all", "fold_bwd"):
    src, out = rnd(R, n * P * P), torch.empty((R, n * n * n), dtype=torch.float64, device=dev)
    timed("xz2d_fold_bwd", R * n * 0.5 * (P * P * n + n * P * n), R * n * (n * n + P * P) * 8.0,
          lambda: hip.xz2d_fold(True, n, R, n, src, src.stride(0), P * P, F, F, out, out.stride(0), n * n))
    del src, out
if which in ("all", "xcorr_fold"):
    src = rnd(R, P * n * n)
    lam = rnd(P * n * P)
    out = torch.empty((R, P * P), dtype=torch.float64, device=dev)
    timed("xcorr_fold", R * P * 0.5 * P * n * n, R * P * (n * n + P) * 8.0,
          lambda: hip.xcorr_reduce_fold(n, R, P, src, src.stride(0), n * n, F, lam, out, out.stride(0), P))
if which in ("all", "ymul"):
    # y step of the lattice Gram: the (128 x 64) matrix from the left of every (64 x 4096) row; flop = MFMA, bytes = in + out
    G = rnd(128, 64)
    src, out = rnd(R, 64 * n * n), torch.empty((R, 128 * n * n), dtype=torch.float64, device=dev)
    timed("ymul", R * 2.0 * 128 * 64 * n * n, R * (64 + 128) * n * n * 8.0, lambda: hip.ymul(128, 64, n * n, R, G, src, src.stride(0), out, out.stride(0)))
    timed("ymul_fold", R * 1.0 * 128 * 64 * n * n, R * (64 + 128) * n * n * 8.0, lambda: hip.ymul(128, 64, n * n, R, G, src, src.stride(0), out, out.stride(0), fold=True))
if which in ("all", "ymul_gemm"):   # the same product as a batched GEMM (what geobo_ymul replaced)
    G = rnd(128, 64)
    src, out = rnd(R, 64 * n * n), torch.empty((R, 128 * n * n), dtype=torch.float64, device=dev)
    timed("ymul_as_gemm_batched", R * 2.0 * 128 * 64 * n * n, R * (64 + 128) * n * n * 8.0,
          lambda: hip.gemm_batched(True, 128, n * n, 64, G, 64, 0, src, n * n, src.stride(0), out, n * n, out.stride(0), 128, n * n, R))
if which in ("all", "xz2d32_fwd", "xz2d32_bwd"):
    # BASELINE config 2 (32^3): 32 x 32 planes through the (64, 32) instance of geobo_xz2d two at a time (M_x -> diag(M_x, M_x))
    m, ny32 = 32, 32
    Mx, Mz = rnd(2 * m, m), rnd(2 * m, m)
    Mx2 = torch.zeros((4 * m, 2 * m), dtype=torch.float64, device=dev); Mx2[:2 * m, :m] = Mx; Mx2[2 * m:, m:] = Mx
    MxT2 = Mx2.t().contiguous(); MzT = Mz.t().contiguous()
if which in ("all", "xz2d32_fwd"):
    src, out = rnd(R, ny32 * m * m), torch.empty((R, ny32 * 4 * m * m), dtype=torch.float64, device=dev)
    timed("xz2d32_fwd", R * ny32 * 2.0 * (m * m * 2 * m + 2 * 2 * m * m * 2 * m), R * ny32 * 5 * m * m * 8.0,
          lambda: hip.xz2d(False, 2 * m, m, R, ny32 // 2, src, src.stride(0), 2 * m * m, Mx2, Mz, out, out.stride(0), 8 * m * m))
if which in ("all", "xz2d32_bwd"):
    src, out = rnd(R, ny32 * 4 * m * m), torch.empty((R, ny32 * m * m), dtype=torch.float64, device=dev)
    timed("xz2d32_bwd", R * ny32 * 2.0 * (2 * m * 2 * m * m + 2 * m * 2 * m * m), R * ny32 * 5 * m * m * 8.0,
          lambda: hip.xz2d(True, 2 * m, m, R, ny32 // 2, src, src.stride(0), 8 * m * m, MxT2, MzT, out, out.stride(0), 2 * m * m))
if which in ("all", "toeplitz128"):
    # BASELINE config 5 (128^3 x 3 properties), the rank-0-of-8 batch: 26 sensor rows, all 128 planes in, the rank's 16 planes out,
    # three property blocks per read of the (x, z)-spectrum (geobo_toeplitz_y3, windowed register table)
    ny, C, Rb = 128, 256 * 256, 26
    src = rnd(Rb * ny * C)
    tabs = [rnd(ny * C) for _ in range(3)]
    outs = [torch.empty(Rb * 16 * C, dtype=torch.float64, device=dev) for _ in range(3)]
    timed("toeplitz_y_win", Rb * C * 2.0 * ny * 16 * 3, Rb * C * 8.0 * (ny + 3 * 16), lambda: hip.toeplitz_y(ny, C, Rb, src, tabs, outs, 0, 16))
    del src, tabs, outs
if which in ("all", "edge_rows"):
    # lattice Gram, one boundary slab for 1024 rows: x-DFT, per-frequency contraction, inverse DFT (three batched GEMMs)
    from geobo_amd.lattice_gram import LatticeGram
    from geobo_amd.spectral import SpectralProduct
    gram = LatticeGram(SpectralProduct(n, n, n, dev), dev)
    E = rnd(n * n + 256, n * n)
    V = gram.edge_eigen(E)
    X, out = rnd(1025, 2 * n * n), torch.zeros((1024, n * n), dtype=torch.float64, device=dev)
    timed("edge_rows", 1024 * 3 * 2.0 * 128 * 128 * 64, 1024 * (n * n + n * n) * 8.0, lambda: gram.edge_rows(X[:, :n * n + 64], 1024, V, out))
# ---- transposed posterior (round 3) ---------------------------------------------------------------------------------------------
if which in ("all", "fold_inv_ss"):
    # inverse transform fused with the sum of squares over rows, two-term input (V = Zg K0j + Zm K1j): reads both y-stage outputs of a
    # 256-row batch, writes nothing but the 32 partial cubes it adds into
    s1, s2 = rnd(R, n * P * P), rnd(R, n * P * P)
    slots = hip.xz2d_fold_inv_ss_slots(n, R, n)
    ss = torch.zeros((slots, n, n * n), dtype=torch.float64, device=dev)
    timed("xz2d_fold_inv_ss", R * n * 0.5 * (2 * P * P * n + n * P * n), 2.0 * R * n * P * P * 8.0,
          lambda: hip.xz2d_fold_inv_ss(n, R, n, s1, n * P * P, P * P, F, F, ss, src2=s2, in2_row=n * P * P, r2_first=0))
    del s1, s2, ss
if which in ("all", "fold_inv_strided"):
    # rows of L^-1 A on a lattice survey: one inverse two-axis transform per (row, z) plane of W = Lambda[iz] * lhat_r, written as [iy][iz][ix]
    W, out = rnd(R, n * P * P), torch.empty((R, n * n * n), dtype=torch.float64, device=dev)
    timed("xz2d_fold_inv_strided", R * n * 0.5 * (P * P * n + n * P * n), R * n * (n * n + P * P) * 8.0,
          lambda: hip.xz2d_fold_inv_strided(n, R, n, W, n * P * P, P * P, F, F, out, out.stride(0), n, n * n))
    del W, out
if which in ("all", "wplanes"):
    lam3, lh = rnd(n * P * P), rnd(R * P * P)
    W = torch.empty(R * n * P * P, dtype=torch.float64, device=dev)
    timed("lattice_wplanes", R * n * P * P * 1.0, (R * n * P * P + R * P * P + n * P * P) * 8.0, lambda: hip.lattice_wplanes(R, P, P, n, lam3, lh, W))
    del lam3, lh, W
if which in ("all", "colgemv"):
    # posterior mean: weighted column sums of A K (8448 x 2 N)
    m_, n_ = 8448, 2 * n * n * n
    X, v = rnd(m_, n_), rnd(m_)
    timed("colgemv", 2.0 * m_ * n_, m_ * n_ * 8.0, lambda: hip.colgemv(X, v))
if which in ("all", "fold_inv_mul"):
    # the same rows of L^-1 A with W = Lambda[iz] * lhat_r formed inside the kernel from the two cache-resident factors: reads 8 MB + 33 MB
    # (once each, algorithmically), writes the rows
    lam3, lh = rnd(n * P * P), rnd(R * P * P)
    out = torch.empty((R, n * n * n), dtype=torch.float64, device=dev)
    timed("xz2d_fold_inv_mul", R * n * 0.5 * (P * P * n + n * P * n), (n * P * P + R * P * P + R * n * n * n) * 8.0,
          lambda: hip.xz2d_fold_inv_mul(n, R, n, lam3, P * P, lh, P * P, F, F, out, out.stride(0), n, n * n))
    del lam3, lh, out
