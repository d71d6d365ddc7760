"""Chronological per-step breakdown of the 64^3 bench step: wall time, device time of every stage (HIP events) and the rest
(host work / idle device).  python tools/step_variation.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                  zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
inv = Inversion(settings=s, props=(0, 1), device="cuda:0")
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
gl = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
def step():
    inv.engine.clear_operators()
    inv.gp_length = gl.copy()
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
step()
for i in range(steps):
    inv.engine.kernel_events = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    wall = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    d = {}
    for name, fl, alg, valu, e0, e1 in inv.engine.kernel_events:
        d[name] = d.get(name, 0.0) + e0.elapsed_time(e1)
    tot = sum(d.values())
    print("step %d: wall %.1f ms, stages %.1f, rest %.1f | %s" % (i, wall, tot, wall - tot, " ".join("%s %.1f" % (k, v) for k, v in d.items())), flush=True)
