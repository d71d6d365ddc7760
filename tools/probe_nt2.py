import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip
from probe_nt import mat, timeit, X, A, C, ws, N
for m in (6144, 6656, 7168):
    for sp in (1, 2, 4, 8):
        f = (lambda: hip.gemm_nt(X[:m], A, C[:m])) if sp == 1 else (lambda: hip.gemm_nt_splitk(X[:m], A, C[:m], sp, ws))
        t = timeit(f)
        print("full NT m=%d (%d tiles) splitk=%d: %.4f s %.1f TF/s" % (m, m // 256 * 32, sp, t, 2.0 * m * 4096 * N / t / 1e12), flush=True)
