"""Phase times of the chain walker of the tile-DAG factorisation (potrf.hip, dag_walk): needs a library built with -DDAG_WALK_STATS
(python tools/ab_variants.py wstats=-DDAG_WALK_STATS; GEOBO_HIP_LIB=geobo_amd/lib/variants/wstats.so python tools/potrf_walk_stats.py 2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geobo_amd import hip, _lib
m = int(sys.argv[1]); nb = m // 128
g = torch.Generator().manual_seed(0)
B = torch.rand((m, 512), generator=g, dtype=torch.float64).cuda()
S = B @ B.t() / 512 + 0.5 * torch.eye(m, dtype=torch.float64, device="cuda")
Linv = torch.empty((m, m), dtype=torch.float64, device="cuda")
nbytes = _lib.load().geobo_potrf_ws_bytes(m)
ws = torch.zeros(nbytes // 8 + 2, dtype=torch.float64, device="cuda")
def tree(n):
    if n <= 1: return 0
    mid = n // 2
    if n > 2 and mid & 1: mid += 1
    return (n - mid) * mid + tree(mid) + tree(n - mid)
for rep in range(3):
    L = S.clone(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.potrf_inv(L, Linv, ws); e1.record(); torch.cuda.synchronize()
raw = ws.cpu().numpy().view(np.int32)
off = tree(nb) * 128 * 128 * 2
c = raw[off:off + 16]
print("m=%d %.3f ms; walker us per column: potf2b+publish %.1f | wait pl %.1f | fin + publish %.1f | wait pd %.1f | update + store %.1f" % ((m, e0.elapsed_time(e1)) + tuple(c[2 + q] / 100.0 / nb for q in range(5))))
