"""Host-side (Python) time of one bench step: cProfile around inv.cubing with the GPU work left asynchronous."""
import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                  zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32",
                  XMAG=0, YMAG=0, ZMAG=1))
inv = Inversion(settings=s, props=(0, 1), device="cuda:0")
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
gl = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
def step():
    inv.engine.clear_operators()
    inv.gp_length = gl.copy()
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
step(); torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); step(); step(); pr.disable()
torch.cuda.synchronize()
print("2 steps wall %.3f s" % (time.perf_counter() - t0))
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
