"""One launch of the materialising covariance-block kernel (geobo_k_block / geobo_k_block_f32: one block of create_cov,
kernels.py:183-195, straight from voxel coordinates) for rocprofv3 PMC passes: 8192 rows x 262144 columns of the 64^3 grid.
    python tools/run_k_block_once.py [family: exp | matern32 | matern32_x | sparse] [f64 | f32] [coords | grid]
grid = the regular-grid form (geobo_k_block_grid: gather from the block's difference-lattice table)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geobo_amd import hip

fam = sys.argv[1] if len(sys.argv) > 1 else "exp"
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
n = 64
ax = (np.arange(n) + 1) * 100.0
X, Y, Z = np.meshgrid(ax, ax, ax)
cols = tuple(hip.to_dev(a.ravel()) for a in (X, Y, Z))
rows = tuple(c[::32].contiguous() for c in cols)                  # 8192 voxels as row points
nr, nc = rows[0].numel(), cols[0].numel()
out = torch.empty((nr, nc), dtype=dt, device="cuda")
kid = hip.KERNEL_IDS[fam]
grid = len(sys.argv) > 3 and sys.argv[3] == "grid"
if grid:
    tab = hip.cov_table(kid, n, n, n, 100.0, 100.0, 100.0, 200.0, 204.0, 0.7, 1.0)
    if dt == torch.float32:
        hip.round_f32_(tab)
    ridx = torch.arange(0, nc, 32, device="cuda", dtype=torch.int64)
    launch = lambda: hip.k_block_grid(tab, n, n, n, ridx, 0, out)
    by = nr * nc * out.element_size() + 8.0 * tab.numel() + 8.0 * nr
else:
    launch = lambda: hip.k_block(kid, rows, cols, 200.0, 204.0, 0.7, 1.0, out)
    by = nr * nc * out.element_size() + 24.0 * (nr + nc)
launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
print("k_block%s %s %s %d x %d: %.5f s, %.2f TB/s written+read (algorithmic bytes %.0f); flop 0" % ("_grid" if grid else "", fam, str(dt).split(".")[1], nr, nc, t, by / t / 1e12, by))
