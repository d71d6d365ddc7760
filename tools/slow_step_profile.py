"""cProfile of every 64^3 bench step on its own; prints the profile of the steps that come out slow (host stalls)."""
import os, sys, time, cProfile, pstats, io
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
for _ in range(3): step()
walls = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable(); step(); pr.disable()
    wall = (time.perf_counter() - t0) * 1e3
    walls.append(wall)
    if i > 3 and wall > np.median(walls) + 12:
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(10)
        print("SLOW step %d: %.1f ms (median %.1f)" % (i, wall, np.median(walls)))
        print("\n".join(l for l in st.getvalue().splitlines()[6:20]), flush=True)
print("walls", [round(w, 1) for w in walls])
