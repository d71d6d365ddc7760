import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
M, Ms, k = 8448, 4096, 65536
g = torch.Generator().manual_seed(0)
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3
C = torch.zeros((M, Ms), dtype=torch.float64, device="cuda")
for ldx_mult, ldy_mult, padx in ((8, 4, 0), (8, 4, 16), (8, 4, 128), (8, 4, 272), (8, 4, 2064)):
    Xb = torch.rand((M, k), generator=g, dtype=torch.float64).cuda() if ldx_mult == 1 else None
    if ldx_mult > 1:
        Xb = torch.empty((M, k * ldx_mult + padx), dtype=torch.float64, device="cuda"); Xb[:, :k] = torch.rand((M, k), generator=g, dtype=torch.float64).cuda()
    Yb = torch.empty((Ms, k * ldy_mult), dtype=torch.float64, device="cuda"); Yb[:, :k] = torch.rand((Ms, k), generator=g, dtype=torch.float64).cuda()
    X, Y = Xb[:, :k], Yb[:, :k]
    t = timed(lambda: hip.gemm_nt(X, Y, C))
    print("gemm_nt %dx%dx%d  row stride X %.1f MB, Y %.1f MB: %.4f s %.1f TF/s" % (M, Ms, k, X.stride(0) * 8 / 1e6, Y.stride(0) * 8 / 1e6, t, 2.0 * M * Ms * k / t / 1e12), flush=True)
    del Xb, Yb, X, Y
