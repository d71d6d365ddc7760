"""Per-tile fixed cost of the NN core: time vs contraction length at fixed tile count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
from probe_nn import timeit
m, n = 8192, 131072
X = torch.rand((m, 8448 + 16), dtype=torch.float64, device="cuda")
Yb = torch.empty((8448, 524288 + 16), dtype=torch.float64, device="cuda"); Yb.uniform_(-1, 1)
C = torch.empty((m, n + 16), dtype=torch.float64, device="cuda")[:, :n]
tiles_per_cu = (m // 256) * (n // 128) / 256
for k in (256, 512, 1024, 2048, 4096, 8192):
    t = timeit(lambda: hip.gemm_nn(X[:, :k], Yb[:k, :n], C))
    print("NN m=%d n=%d k=%d: %.5f s %.1f TF/s, %.1f us per tile" % (m, n, k, t, 2.0 * m * n * k / t / 1e12, t / tiles_per_cu * 1e6), flush=True)
