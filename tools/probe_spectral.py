"""Spectral vs dense AK on the GPU: correctness + timing (scratch tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geobo_amd import hip
from geobo_amd.config_loader import Settings
from geobo_amd.engine import PosteriorEngine, weight_matrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kern = sys.argv[2] if len(sys.argv) > 2 else "matern32"
s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc=kern))
xs = np.linspace(0.5, n - 0.5, n) * 100.0
X, Y, Z = np.meshgrid(xs, xs, 1.0)
loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
lengths = [200.0, 202.0, 204.0]
W = weight_matrix(s.gp_coeff)
res = {}
for method in ("dense", "spectral"):
    eng = PosteriorEngine(s, method=method)
    A_g, A_m = eng.operator("grav", loc), eng.operator("magn", loc)
    sel = torch.as_tensor(np.arange(0, eng.N, eng.N // 40)[:40], device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        AK, M_pad = eng._assemble_AK(A_g, A_m, sel, lengths, W, kern, 1.0, (0, 1))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: AK assembly %.3f s" % (method, dt), flush=True)
    res[method] = AK
    del eng
d = (res["dense"] - res["spectral"]).abs().max().item()
print("max|dense - spectral| = %.3e  (max|AK| = %.3e, rel %.3e)" % (d, res["dense"].abs().max().item(), d / res["dense"].abs().max().item()))
