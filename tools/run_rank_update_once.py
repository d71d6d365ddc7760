"""One rank-128 trailing update of the Cholesky factorisation at m = 8192 (C <- C - P P^T, lower tiles), for rocprofv3 PMC passes:
algorithmic bytes = C read and written once (lower triangle) + the panel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
m, k = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = torch.zeros((m, m), dtype=torch.float64, device="cuda")
P = torch.rand((m, k), dtype=torch.float64, device="cuda")
f = lambda: hip.gemm_nt(P, P, C, alpha=-1.0, beta=1.0, lower_only=True)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
fl, by = 1.0 * m * m * k, m * m * 8.0 + m * k * 8.0
print("rank_update m=%d k=%d: %.6f s, %.1f TF/s executed, flop %.0f; algorithmic bytes %.3e (%.2f TB/s)" % (m, k, t, fl / t / 1e12, fl, by, by / t / 1e12))
