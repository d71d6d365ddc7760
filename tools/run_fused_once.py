"""One representative launch of the dominant kernel for rocprofv3 PMC passes: 64^3 geometry (Ms=4096, N=262144),
`ncols` output columns (default 32768 = 1/8 of the full launch, 16 rounds of 256 workgroups)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
n = 64
ncols = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N, Ms = n ** 3, n * n
g = torch.Generator().manual_seed(0)
A = torch.rand((Ms, N), generator=g, dtype=torch.float64).cuda()
out = torch.empty((Ms, ncols), dtype=torch.float64, device="cuda")
tab = hip.cov_table(hip.kernel_id("matern32", True), n, n, n, 100., 100., 100., 200.0, 204.0, 0.2, 1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.ak_fused_grid(A, n, n, n, tab, 0, ncols, out); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
print("ak_fused_grid Ms=%d N=%d ncols=%d: %.4f s %.1f TF/s; algorithmic bytes: A %.3e (once) + out %.3e" % (Ms, N, ncols, t, 2.0 * Ms * N * ncols / t / 1e12, Ms * N * 8.0, Ms * ncols * 8.0))
