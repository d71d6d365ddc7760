import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geobo_amd import hip
from geobo_amd.config_loader import Settings
from geobo_amd.engine import PosteriorEngine, weight_matrix
from geobo_amd.spectral import SpectralProduct
n = 64
s = Settings(dict(xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n, kernelfunc="matern32"))
xs = np.linspace(0.5, n - 0.5, n) * 100.0
X, Y, Z = np.meshgrid(xs, xs, 1.0)
loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
eng = PosteriorEngine(s)
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print("%-28s %.4f s" % (label, time.perf_counter() - t0), flush=True); return r
for rep in range(2):
    print("--- rep", rep)
    eng.clear_operators()
    A_g = T("operator grav", lambda: eng.operator("grav", loc))
    A_m = T("operator magn", lambda: eng.operator("magn", loc))
    AK = T("zeros AK", lambda: torch.zeros((8448, 2 * eng.N), dtype=torch.float64, device="cuda"))
    if eng._spectral is None:
        eng._spectral = T("SpectralProduct()", lambda: SpectralProduct(n, n, n, eng.device))
    sp = eng._spectral
    tab = T("cov_table", lambda: hip.cov_table(4, n, n, n, 100., 100., 100., 200., 202., 0.2, 1.0))
    lam = T("eigenvalues", lambda: sp.eigenvalues(tab))
    T("product grav (2 blocks)", lambda: sp.product(A_g, eng.Ms, [lam, lam], [AK[:4096, :eng.N], AK[:4096, eng.N:]]))
    del AK
