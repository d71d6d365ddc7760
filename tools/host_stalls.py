"""Where the host loses time in the odd slow 64^3 step: every blocking call of a step (device->host reads, pageable uploads) is
preceded by an explicit synchronize, so its own duration is separated from waiting for the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geobo_amd import hip
from geobo_amd.config_loader import Settings
from geobo_amd.inversion import Inversion
n = 64
s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                  zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
inv = Inversion(settings=s, props=(0, 1), device="cuda:0")
grav, mag, loc, drill0 = bench.synthetic_inputs(inv, 50)
gl = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
acc = {}
SYNC_FIRST = os.environ.get('SYNC_FIRST', '1') == '1'
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        if SYNC_FIRST: torch.cuda.synchronize()
        t1 = time.perf_counter()
        r = f(*a, **k)
        t2 = time.perf_counter()
        acc["sync"] = acc.get("sync", 0.0) + (t1 - t0) * 1e3
        acc[label] = acc.get(label, 0.0) + (t2 - t1) * 1e3
        return r
    setattr(obj, name, g)
wrap(torch.Tensor, "cpu", "cpu()")
wrap(torch.Tensor, "item", "item()")
wrap(torch.Tensor, "tolist", "tolist()")
wrap(hip, "to_dev", "to_dev")
def step():
    inv.engine.clear_operators()
    inv.gp_length = gl.copy()
    return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
for _ in range(2): step()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    wall = (time.perf_counter() - t0) * 1e3
    other = wall - sum(acc.values())
    print("step %d: %.1f ms | %s | python/other %.1f" % (i, wall, " ".join("%s %.1f" % kv for kv in acc.items()), other), flush=True)
