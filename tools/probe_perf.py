"""Per-kernel timing probe (GPU box): prints TFLOP/s of the MFMA kernels and GB/s of k_block. Scratch tool."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geobo_amd import hip

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best

what = sys.argv[1:] or ["peak", "fused", "nt", "potrf", "post", "kblock", "asens"]
if "peak" in what:
    for blocks, iters in ((256, 20000), (512, 20000), (1024, 20000), (2048, 10000)):
        print("mfma f64 peak blocks=%d: %.1f TF/s" % (blocks, hip.mfma_f64_peak(blocks, iters)), flush=True)
n = 32
N, Ms = n ** 3, n * n
g = torch.Generator().manual_seed(0)
if "fused" in what:
    for (n_, ms_) in ((32, 1024), (40, 4096)):
        N, Ms = n_ ** 3, ms_
        A = torch.rand((Ms, N), generator=g, dtype=torch.float64).cuda()
        idx = np.arange(N)
        xyz = tuple(hip.to_dev(v * 100.0) for v in ((idx // n_) % n_ + 1.0, idx // (n_ * n_) + 1.0, idx % n_ + 1.0))
        out = torch.empty((Ms, N), dtype=torch.float64, device="cuda")
        for name, cross in (("exp", False), ("exp", True), ("matern32", False), ("matern32", True), ("d2", None)):
            kid = hip.KERNEL_IDS["d2"] if cross is None else hip.kernel_id(name, cross)
            t = timed(lambda: hip.ak_fused(kid, A, xyz, 0, N, 200.0, 204.0, 0.2, 1.0, out), reps=2)
            print("ak_fused N=%d Ms=%d %s cross=%s: %.3f s  %.1f TF/s" % (N, Ms, name, cross, t, 2.0 * Ms * N * N / t / 1e12), flush=True)
        if n_ >= 16:
            tab = hip.cov_table(hip.kernel_id("matern32", True), n_, n_, n_, 100., 100., 100., 200.0, 204.0, 0.2, 1.0)
            t = timed(lambda: hip.ak_fused_grid(A, n_, n_, n_, tab, 0, N, out), reps=2)
            print("ak_fused_grid N=%d Ms=%d: %.3f s  %.1f TF/s" % (N, Ms, t, 2.0 * Ms * N * N / t / 1e12), flush=True)
        del A, out
if "nt" in what:
    for (m, n2, k) in ((2304, 1024, 32768), (8448, 4096, 65536)):
        X = torch.rand((m, k), generator=g, dtype=torch.float64).cuda(); Y = torch.rand((n2, k), generator=g, dtype=torch.float64).cuda()
        C = torch.empty((m, n2), dtype=torch.float64, device="cuda")
        t = timed(lambda: hip.gemm_nt(X, Y, C))
        print("gemm_nt %dx%dx%d: %.4f s %.1f TF/s" % (m, n2, k, t, 2.0 * m * n2 * k / t / 1e12), flush=True)
        del X, Y, C
if "potrf" in what:
    for m in (2304, 8448):
        B = torch.rand((m, m), generator=g, dtype=torch.float64).cuda()
        S = B @ B.t() / m + torch.eye(m, dtype=torch.float64, device="cuda")
        def run():
            L = S.clone(); hip.potrf_inv(L)
        t0 = timed(lambda: S.clone())
        t = timed(run)
        print("potrf_inv m=%d: %.4f s (clone %.4f) -> %.2f TF/s on 2/3 m^3" % (m, t, t0, (2.0 / 3 * m ** 3) / (t - t0) / 1e12), flush=True)
        del B, S
if "post" in what:
    for (m, nc) in ((2304, 65536), (8448, 131072)):
        Linv = torch.tril(torch.rand((m, m), generator=g, dtype=torch.float64)).cuda()
        AK = torch.rand((m, nc), generator=g, dtype=torch.float64).cuda(); u = torch.rand(m, generator=g, dtype=torch.float64).cuda()
        t = timed(lambda: hip.posterior_reduce(Linv, AK, u, 1.0))
        print("posterior_reduce m=%d nc=%d: %.4f s %.1f TF/s (m^2 nc flops)" % (m, nc, t, 1.0 * m * m * nc / t / 1e12), flush=True)
        del Linv, AK
if "kblock" in what:
    nr, nc = 8192, 262144
    idx = np.arange(nc)
    cxyz = tuple(hip.to_dev(v * 100.0) for v in ((idx // 64) % 64 + 1.0, idx // 4096 + 1.0, idx % 64 + 1.0))
    rxyz = tuple(c[:nr].clone() for c in cxyz)
    out = torch.empty((nr, nc), dtype=torch.float64, device="cuda")
    for name, cross in (("d2", None), ("exp", False), ("exp", True), ("matern32", False), ("matern32", True), ("sparse", False)):
        kid = hip.KERNEL_IDS["d2"] if cross is None else hip.kernel_id(name, cross)
        t = timed(lambda: hip.k_block(kid, rxyz, cxyz, 200.0, 204.0, 0.2, 1.0, out))
        print("k_block %dx%d %s cross=%s: %.4f s  %.0f GB/s written" % (nr, nc, name, cross, t, nr * nc * 8 / t / 1e9), flush=True)
if "asens" in what:
    from geobo_amd.config_loader import Settings
    from geobo_amd.engine import PosteriorEngine
    for n_ in (32, 64):
        s = Settings(dict(xmax=100.0 * n_, ymax=100.0 * n_, zLcube=100.0 * n_, xNcube=n_, yNcube=n_, zNcube=n_))
        eng = PosteriorEngine(s)
        xs = np.linspace(0.5, n_ - 0.5, n_) * 100.0
        X, Y, Z = np.meshgrid(xs, xs, 1.0)
        loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
        for f in ("grav", "magn"):
            torch.cuda.synchronize(); t0 = time.perf_counter(); eng.operator(f, loc); torch.cuda.synchronize()
            print("a_sens %s n=%d: %.4f s" % (f, n_, time.perf_counter() - t0), flush=True)
        del eng
