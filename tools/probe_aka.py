import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
M, Ms, N = 8448, 4096, 262144
g = torch.Generator().manual_seed(0)
AK = torch.empty((M, 2 * N), dtype=torch.float64, device="cuda")
for c in range(0, 2 * N, 65536):
    AK[:, c:c + 65536] = torch.rand((M, 65536), generator=g, dtype=torch.float64).cuda()
A = torch.rand((Ms, N), generator=g, dtype=torch.float64).cuda()
AkA = torch.zeros((M, M), dtype=torch.float64, device="cuda")
ws = torch.empty(16 * M * Ms, dtype=torch.float64, device="cuda")
def run(splits):
    for s_ in (0, 1):
        r0 = s_ * Ms
        X, C = AK[r0:, s_ * N:(s_ + 1) * N], AkA[r0:, r0:r0 + Ms]
        if splits == 1: hip.gemm_nt(X, A, C, lower_only=True)
        else: hip.gemm_nt_splitk(X, A, C, splits, ws, lower_only=True)
tiles = sum(min(2 * (b + 1), 32) for b in range(33)) + sum(min(2 * (b + 1), 32) for b in range(17))
for splits in (1, 2, 4, 8, 16):
    run(splits); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(splits); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    print("map=%s splits=%2d: %.4f s  %.1f TF/s" % (os.environ.get("GEOBO_TILE_MAP", "default"), splits, t, 2.0 * 256 * 128 * N * tiles / t / 1e12), flush=True)
