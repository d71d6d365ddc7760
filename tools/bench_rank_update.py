"""Rank-k update  C <- C - P P^T  (lower tiles) at the shapes of the Cholesky trailing updates: time, flop rate and the traffic of C
for k = 128 / 256 / 512, with and without the read of C (beta).  python tools/bench_rank_update.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in (8192, 8320, 4096, 2048):
    C = torch.zeros((m, m), dtype=torch.float64, device="cuda")
    for k in (128, 256, 512):
        P = torch.rand((m, k), dtype=torch.float64, device="cuda")
        for beta in (0.0, 1.0):
            us = t(lambda: hip.gemm_nt(P, P, C, alpha=-1.0, beta=beta, lower_only=True))
            fl = m * m * k * 1.0
            print("m=%d k=%d beta=%g: %.1f us  %.1f TF/s  C traffic %.2f TB/s" % (m, k, beta, us, fl / us / 1e6, m * m * 4 * (1 + (beta != 0)) / us / 1e6), flush=True)
