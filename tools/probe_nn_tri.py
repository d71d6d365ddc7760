import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
from probe_nn import timeit
m = k = 8448
n = 131072
X = torch.rand((m, k + 16), dtype=torch.float64, device="cuda")[:, :k]
Yb = torch.empty((k, 524288 + 16), dtype=torch.float64, device="cuda"); Yb.uniform_(-1, 1)
C = torch.empty((m, n + 16), dtype=torch.float64, device="cuda")[:, :n]
for mm in (8448, 8192, 4096, 2048, 1024):
    t = timeit(lambda: hip.gemm_nn(X[:mm, :mm], Yb[:mm, :n], C[:mm], x_lower=True))
    nb = mm // 256
    fl = 2.0 * 256 * 256 * n * nb * (nb + 1) / 2
    ideal = (n // 128) * (nb * 17e-6 + 54.5e-6 * nb * (nb + 1) / 2) / 256
    print("map=%s tri NN m=%d: %.4f s %.1f TF/s (tile-model %.4f s)" % (os.environ.get("GEOBO_TILE_MAP", "auto"), mm, t, fl / t / 1e12, ideal), flush=True)
