"""Base rate of the NT GEMM core on AkA-like shapes (padded leading dimensions), full vs lower-only, split-K variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
N = 262144
def mat(r, c, pad=16):
    buf = torch.empty((r, c + pad), dtype=torch.float64, device="cuda")
    g = torch.Generator().manual_seed(r)
    for c0 in range(0, c, 65536):
        buf[:, c0:c0 + 65536] = torch.rand((r, 65536), generator=g, dtype=torch.float64).cuda()
    return buf[:, :c]
def timeit(f, n=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
X = mat(8448, N); A = mat(4096, N)
C = torch.zeros((8448, 4096), dtype=torch.float64, device="cuda")
ws = torch.empty(8 * 8448 * 4096, dtype=torch.float64, device="cuda")
for m, n in ((8192, 4096), (4096, 4096), (2048, 4096), (8192, 2048)):
    t = timeit(lambda: hip.gemm_nt(X[:m], A[:n], C[:m, :n]))
    print("full NT m=%d n=%d k=%d: %.4f s %.1f TF/s (%d tiles)" % (m, n, N, t, 2.0 * m * n * N / t / 1e12, m // 256 * (n // 128)), flush=True)
for sp in (2, 4):
    t = timeit(lambda: hip.gemm_nt_splitk(X[:8192], A, C[:8192], sp, ws))
    print("full NT m=8192 n=4096 splitk=%d: %.4f s %.1f TF/s" % (sp, t, 2.0 * 8192 * 4096 * N / t / 1e12), flush=True)
for rows in (8448, 4352):
    tiles = sum(min(2 * (b + 1), 32) for b in range(rows // 256))
    for sp in (1, 2, 4):
        f = (lambda: hip.gemm_nt(X[:rows], A, C[:rows], lower_only=True)) if sp == 1 else (lambda: hip.gemm_nt_splitk(X[:rows], A, C[:rows], sp, ws, lower_only=True))
        t = timeit(f)
        print("lower NT rows=%d splitk=%d: %.4f s %.1f TF/s (%d tiles)" % (rows, sp, t, 2.0 * 256 * 128 * N * tiles / t / 1e12, tiles), flush=True)
