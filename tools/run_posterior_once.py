"""One launch of the spectral route's dominant kernel (geobo_posterior_reduce) at the 64^3 headline shape, for rocprofv3
PMC passes: m = 8448 rows, ncols = 2 * 262144 voxel-property columns."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
m, ncols = 8448, 2 * 262144
g = torch.Generator().manual_seed(0)
Linv = torch.tril(torch.rand((m, m), generator=g, dtype=torch.float64)).cuda()
AK = torch.empty((m, ncols), dtype=torch.float64, device="cuda")
for c in range(0, ncols, 65536):
    AK[:, c:c + 65536] = torch.rand((m, 65536), generator=g, dtype=torch.float64).cuda()
u = torch.rand(m, generator=g, dtype=torch.float64).cuda()
ws = torch.empty(hip.posterior_ws_doubles(m, ncols), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.posterior_reduce(Linv, AK, u, 1.0, ws); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
fl = 2.0 * ncols * sum(64.0 * (256 * b + 64 * (g + 1)) for b in range(m // 256) for g in range(4))
print("posterior_reduce m=%d ncols=%d: %.4f s, %.1f TF/s executed, flop %.0f; algorithmic bytes: AK %.3e + Linv(lower) %.3e" % (m, ncols, t, fl / t / 1e12, fl, m * ncols * 8.0, m * m * 4.0))
