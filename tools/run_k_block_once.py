"""One launch of the materialising covariance-block kernel (geobo_k_block / geobo_k_block_f32: one block of create_cov,
kernels.py:183-195, straight from voxel coordinates) for rocprofv3 PMC passes: 8192 rows x 262144 columns of the 64^3 grid.
    python tools/run_k_block_once.py [family: exp | matern32 | matern32_x | sparse] [f64 | f32]"""
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
hip.k_block(kid, rows, cols, 200.0, 204.0, 0.7, 1.0, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); hip.k_block(kid, rows, cols, 200.0, 204.0, 0.7, 1.0, out); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3
by = nr * nc * out.element_size() + 24.0 * (nr + nc)
print("k_block %s %s %d x %d: %.5f s, %.2f TB/s written+read (algorithmic bytes %.0f); flop 0" % (fam, str(dt).split(".")[1], nr, nc, t, by / t / 1e12, by))
