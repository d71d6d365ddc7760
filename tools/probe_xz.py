import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
nx = nz = 64
rows, ppr = 256, 64
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
for inverse in (False, True):
    ix, iz, ox, oz = (2 * nx, 2 * nz, nx, nz) if inverse else (nx, nz, 2 * nx, 2 * nz)
    Mx = torch.rand((ox, ix), dtype=torch.float64, device="cuda"); Mz = torch.rand((oz, iz), dtype=torch.float64, device="cuda")
    src = torch.rand((rows, ppr * ix * iz), dtype=torch.float64, device="cuda")
    out = torch.empty((rows, ppr * ox * oz), dtype=torch.float64, device="cuda")
    t = timeit(lambda: hip.xz2d(inverse, nx, nz, rows, ppr, src, src.stride(0), ix * iz, Mx, Mz, out, out.stride(0), ox * oz))
    fl = rows * ppr * 2.0 * (ix * iz * oz + ox * ix * oz)
    gb = rows * ppr * (ix * iz + ox * oz) * 8 / 1e9
    print("%s xz2d inverse=%s: %.4f ms  %.1f TF/s  %.2f TB/s  %.2f us/plane/CU" % (os.environ.get("GEOBO_HIP_LIB", "default").split("/")[-1], inverse, t * 1e3, fl / t / 1e12, gb / t / 1e3, t * 256 / (rows * ppr) * 1e6), flush=True)
