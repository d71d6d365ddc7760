"""Warm timing of the y-stage kernels at the 64^3 batch shape (256 rows) or, with an argument, at n^3 (n = 128: 32 rows of 64 MB, the
windowed kernel): 40 back-to-back launches each,
HIP events around the lot.
    python tools/time_toeplitz.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R, C = (256 if n <= 64 else 32), 4 * n * n
rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device="cuda") * 2 - 1
src, src2 = rnd(R * n * C), rnd(R * n * C)
tabs = [rnd(n * C) for _ in range(4)]
outs = [torch.empty(R * n * C, dtype=torch.float64, device="cuda") for _ in range(3)]
def timed(name, flop, by, f, reps=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print("%-22s %.4f ms  %.1f TF/s fp64 FMA (%.2f of 78.6)  %.2f TB/s (%.2f of 8)" % (name, t * 1e3, flop / t / 1e12, flop / t / 78.6e12, by / t / 1e12, by / t / 8e12), flush=True)
timed("toeplitz_y 2 blocks", R * C * 2.0 * n * n * 2, R * n * C * 8.0 * 3, lambda: hip.toeplitz_y(n, C, R, src, tabs[:2], outs[:2]))
timed("toeplitz_y 1 block", R * C * 2.0 * n * n, R * n * C * 8.0 * 2, lambda: hip.toeplitz_y(n, C, R, src, tabs[:1], outs[:1]))
if n > 64 and n in hip.SPECTRAL_Y_NY:
    timed("spectral_y 1 block", R * C * 2.0 * (1.0 * n * n + 9 * n), R * n * C * 8.0 * 2, lambda: hip.spectral_y(n, C, R, src, tabs[:1], outs[:1]))
    timed("spectral_y 2 blocks", R * C * 2.0 * (1.5 * n * n + 14 * n), R * n * C * 8.0 * 3, lambda: hip.spectral_y(n, C, R, src, tabs[:2], outs[:2]))
    timed("spectral_y 3 blocks", R * C * 2.0 * (2.0 * n * n + 19 * n), R * n * C * 8.0 * 4, lambda: hip.spectral_y(n, C, R, src, tabs[:3], outs))
    timed("spectral_y 3 blocks, accumulating", R * C * 2.0 * (2.0 * n * n + 19 * n), R * n * C * 8.0 * 7, lambda: hip.spectral_y(n, C, R, src, tabs[:3], outs, accumulate=True))
    timed("spectral_y3t two terms -> 3 blocks", R * C * 2.0 * (2.5 * n * n + 34 * n), R * n * C * 8.0 * 5, lambda: hip.spectral_y3t(n, C, R, src, src2, tabs[:3], tabs[1:4], outs))
    timed("spectral_y 1 block, 16-plane slab", R * C * 2.0 * (1.0 * n * n + 9 * n), R * C * 8.0 * (n + 16), lambda: hip.spectral_y(n, C, R, src, tabs[:1], outs[:1], 16, 32))
if n > 64:
    timed("toeplitz_y 3 blocks", R * C * 2.0 * n * n * 3, R * n * C * 8.0 * 4, lambda: hip.toeplitz_y(n, C, R, src, tabs[:3], outs))
    timed("toeplitz_y 1 block, 16-plane slab", R * C * 2.0 * n * 16, R * C * 8.0 * (n + 16), lambda: hip.toeplitz_y(n, C, R, src, tabs[:1], outs[:1], 16, 32))
    sys.exit(0)
if n <= 64 and n in hip.SPECTRAL_Y_NY:
    # executed multiply-adds per mode: ny^2 / 2 per transform on the matrix pipe (radix 4) + the orbit butterflies on the vector pipe
    timed("spectral_y 2 blocks", R * C * 2.0 * (1.5 * n * n + 14 * n), R * n * C * 8.0 * 3, lambda: hip.spectral_y(n, C, R, src, tabs[:2], outs[:2]))
    timed("spectral_y 1 block", R * C * 2.0 * (1.0 * n * n + 9 * n), R * n * C * 8.0 * 2, lambda: hip.spectral_y(n, C, R, src, tabs[:1], outs[:1]))
    timed("spectral_y2s", R * C * 2.0 * (2.0 * n * n + 24 * n), R * n * C * 8.0 * 4, lambda: hip.spectral_y2s(n, C, R, src, src2, tabs[0], tabs[1], tabs[2], outs[:2]))
timed("toeplitz_y2t", R * C * 2.0 * n * n * 4, R * n * C * 8.0 * 4, lambda: hip.toeplitz_y2t(n, C, R, src, src2, tabs[:2], tabs[2:], outs[:2]))
timed("toeplitz_y2s (3 products)", R * C * 2.0 * n * n * 3, R * n * C * 8.0 * 4, lambda: hip.toeplitz_y2s(n, C, R, src, src2, tabs[0], tabs[1], tabs[2], outs[:2]))
