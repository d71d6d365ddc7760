"""Warm timing of the batched axis passes (radix-2 GEMM passes geobo_gemm_fold; round 6: + the radix-4 axis kernels geobo_spectral_axis) of a grid without fused (x, z) kernels: the four passes of one
covariance product (z and x analysis, x and z synthesis) on R rows of an n^3 grid, HIP events around 10 repetitions of each.
    python tools/time_axis_passes.py [n = 128] [R = 16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
from geobo_amd.spectral import SpectralProduct
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sp = SpectralProduct(n, n, n, "cuda", rows_per_batch=R)
assert not sp.fused_xz
nx = ny = nz = n
Px = Pz = 2 * n
N = n ** 3
src = torch.rand(R * N + 4096, dtype=torch.float64, device="cuda")
t1 = sp.buf("T1", R * ny * nx * Pz)
t2 = sp.buf("T2", R * ny * Px * Pz)
u1 = sp.buf("U1", R * ny * nx * Pz)
out = torch.empty(R * N + 4096, dtype=torch.float64, device="cuda")
fold = sp.fold
passes = {
    "z analysis  (FWD_Z)": (lambda: hip.axis_pass(fold, False, False, hip.pad_n(ny * nx), hip.pad_n(Pz), nz, src, nz, N, sp.G["z"], nz, 0, t1, Pz, ny * nx * Pz, ny * nx, Pz, R),
                            R * ny * nx * Pz * nz * 1.0, 8.0 * R * (N + ny * nx * Pz)),
    "x analysis  (FWD_X)": (lambda: hip.axis_pass(fold, True, False, hip.pad_n(Px), hip.pad_n(Pz), nx, sp.G["x"], nx, 0, t1, Pz, nx * Pz, t2, Pz, Px * Pz, Px, Pz, R * ny),
                            R * ny * Px * Pz * nx * 1.0, 8.0 * R * ny * (nx * Pz + Px * Pz)),
    "x synthesis (INV_X)": (lambda: hip.axis_pass(fold, True, True, hip.pad_n(nx), hip.pad_n(Pz), Px, sp.GT["x"], Px, 0, t2, Pz, Px * Pz, u1, Pz, nx * Pz, nx, Pz, R * ny),
                            R * ny * nx * Pz * Px * 1.0, 8.0 * R * ny * (Px * Pz + nx * Pz)),
    "z synthesis (INV_Z)": (lambda: hip.axis_pass(fold, False, True, hip.pad_n(ny * nx), hip.pad_n(nz), Pz, u1, Pz, ny * nx * Pz, sp.GT["z"], Pz, 0, out, nz, N, ny * nx, nz, R),
                            R * ny * nx * nz * Pz * 1.0, 8.0 * R * (ny * nx * Pz + N)),
}
if sp.x_mfma:
    # round 6: the x passes as radix-4 axis kernels on the half-integer basis (geobo_spectral_axis: n^2 / 2 multiply-adds per item and mode)
    passes["x analysis  (axis kernel)"] = (lambda: hip.spectral_axis(False, nx, Pz, Pz, Pz, nx * Pz, Px * Pz, R * ny, t1, t2),
                                           R * ny * Px * Pz * nx * 0.5, 8.0 * R * ny * (nx * Pz + Px * Pz))
    passes["x synthesis (axis kernel)"] = (lambda: hip.spectral_axis(True, nx, Pz, Pz, Pz, Px * Pz, nx * Pz, R * ny, t2, u1),
                                           R * ny * nx * Pz * Px * 0.5, 8.0 * R * ny * (Px * Pz + nx * Pz))
for name, (f, flop, by) in passes.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 10
    print("n=%d R=%d %-22s %.3f ms  %.1f TF/s executed (%.2f of 78.6)  %.2f TB/s" % (n, R, name, t * 1e3, flop / t / 1e12, flop / t / 78.6e12, by / t / 1e12), flush=True)
