"""NN GEMM rate vs the row stride of the k x n operand (posterior_reduce reads AK with a 4 MiB row stride)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
def timeit(f, n=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
m = k = 8448
g = torch.Generator().manual_seed(1)
X = torch.rand((m, k + 16), generator=g, dtype=torch.float64).cuda()[:, :k]
for n, ld in ((32768, 32768 + 16), (32768, 524288 + 16), (131072, 131072 + 16), (131072, 524288 + 16)):
    Yb = torch.empty((k, ld), dtype=torch.float64, device="cuda")
    Yb.uniform_(-1, 1)
    Y = Yb[:, :n]
    C = torch.empty((m, n + 16), dtype=torch.float64, device="cuda")[:, :n]
    for xl in (False, True):
        t = timeit(lambda: hip.gemm_nn(X, Y, C, x_lower=xl))
        fl = 2.0 * m * n * k * (0.5 * (1 + 256 / m * 1.0) if xl else 1.0)
        print("NN m=k=%d n=%d ldy=%d x_lower=%s: %.4f s %.1f TF/s" % (m, n, ld, xl, t, fl / t / 1e12), flush=True)
    del Yb, Y, C
