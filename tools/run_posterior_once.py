"""One launch of the spectral route's dominant kernel (geobo_posterior_reduce) at the 64^3 headline shape, for rocprofv3
PMC passes: m = 8448 rows (8242 valid), ncols = 2 * 262144 voxel-property columns."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
m, ncols, mv = 8448, 2 * 262144, 8242   # 8242 observation rows in 8448 padded ones, like the bench
g = torch.Generator().manual_seed(0)
Linv = torch.tril(torch.rand((m, m), generator=g, dtype=torch.float64)).cuda()
AK = torch.empty((m, ncols), dtype=torch.float64, device="cuda")
for c in range(0, ncols, 65536):
    AK[:, c:c + 65536] = torch.rand((m, 65536), generator=g, dtype=torch.float64).cuda()
u = torch.rand(m, generator=g, dtype=torch.float64).cuda()
ws = torch.empty(hip.posterior_ws_doubles(m, ncols), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.posterior_reduce(Linv, AK, u, 1.0, ws, m_valid=mv); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
fl = 2.0 * ncols * sum(64.0 * 64 * g + 2560.0 for g in range((mv + 63) // 64))   # executed: see engine.posterior
print("posterior_reduce m=%d (valid %d) ncols=%d: %.4f s, %.1f TF/s executed, flop %.0f; algorithmic bytes: AK %.3e + Linv(lower) %.3e" % (m, mv, ncols, t, fl / t / 1e12, fl, m * ncols * 8.0, m * m * 4.0))
