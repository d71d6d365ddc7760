import sys, os, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                  zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
inv = Inversion(settings=s, props=(0, 1), device="cuda:0", profile=True)
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
gl = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
def step():
    inv.engine.clear_operators()
    inv.gp_length = gl.copy()
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
step(); step()
inv.engine.timings = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 3
print("step %.1f ms (with stage syncs)" % (tot * 1e3))
print({k: round(v / 3 * 1e3, 2) for k, v in inv.engine.timings.items()}, "sum %.1f" % (sum(inv.engine.timings.values()) / 3 * 1e3))
inv2 = Inversion(settings=s, props=(0, 1), device="cuda:0")
inv2.sensor_locations = loc
def step2():
    inv2.engine.clear_operators()
    inv2.gp_length = gl.copy()
    return inv2.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
step2(); step2()
pr = cProfile.Profile(); torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable(); step2(); step2(); pr.disable(); torch.cuda.synchronize()
print("2 steps wall %.3f s" % (time.perf_counter() - t0))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumtime").print_stats(45); print(st.getvalue()[:9000])
