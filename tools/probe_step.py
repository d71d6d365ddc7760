"""Where does a 64^3 step spend its wall time? (host-side breakdown, scratch tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, cProfile, pstats
import bench
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc="matern32"))
inv = Inversion(settings=s, props=(0, 1), profile=True)
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
def step():
    inv.engine.clear_operators()
    inv.gp_length = np.array([200.0, 202.0, 204.0])
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
step()
inv.engine.timings = {}
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("step wall %.3f s" % (time.perf_counter() - t0))
print({k: round(v, 4) for k, v in inv.engine.timings.items()})
inv.engine.profile = False
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
