"""Which Python garbage collections happen inside the 64^3 step and how long they take (gc.callbacks)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                  zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
inv = Inversion(settings=s, props=(0, 1), device="cuda:0")
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
gl = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
def step():
    inv.engine.clear_operators()
    inv.gp_length = gl.copy()
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
log = []
t_start = [0.0]
def cb(phase, info):
    if phase == "start":
        t_start[0] = time.perf_counter()
    else:
        log.append((info["generation"], (time.perf_counter() - t_start[0]) * 1e3, info["collected"]))
gc.callbacks.append(cb)
for _ in range(3): step()
inv.engine.kernel_events = []
for i in range(12):
    log.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    wall = (time.perf_counter() - t0) * 1e3
    print("step %d: %.1f ms; collections: %s" % (i, wall, ", ".join("gen%d %.1f ms (%d freed)" % c for c in log if c[1] > 0.3) or "none > 0.3 ms"), "| total", len(log), flush=True)
print("tracked objects:", len(gc.get_objects()))
