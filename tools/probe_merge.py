import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
from probe_nn import timeit
m = 8448
L = torch.tril(torch.rand((m, m + 16), dtype=torch.float64, device="cuda")[:, :m])
Li = torch.tril(torch.rand((m, m + 16), dtype=torch.float64, device="cuda")[:, :m])
for r, c in ((4352, 4096), (4096, 4096), (2048, 2048), (1024, 1024)):
    T = torch.empty((r, c), dtype=torch.float64, device="cuda")
    X, Y = L[c:c + r, :c], Li[:c, :c]
    for yl in (False, True):
        t = timeit(lambda: hip.gemm_nn(X, Y, T, y_lower=yl), n=5)
        fl = 2.0 * r * c * c * (0.5 if yl else 1.0)
        print("T = L[hi,lo] Linv[lo,lo] r=%d c=%d y_lower=%s: %.3f ms %.1f TF/s" % (r, c, yl, t * 1e3, fl / t / 1e12), flush=True)
    X2 = Li[c:c + r, c:c + r]
    O = Li[c:c + r, :c]
    for xl in (False, True):
        t = timeit(lambda: hip.gemm_nn(X2, T, O, alpha=-1.0, x_lower=xl), n=5)
        fl = 2.0 * r * r * c * (0.5 if xl else 1.0)
        print("Linv[hi,lo] = -Linv[hi,hi] T r=%d c=%d x_lower=%s: %.3f ms %.1f TF/s" % (r, c, xl, t * 1e3, fl / t / 1e12), flush=True)
