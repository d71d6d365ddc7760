"""Task timeline of the persistent tile-DAG factorisation (geobo_potrf_inv, potrf.hip) from its own trace records.
Needs a library built with -DGEOBO_DAG_TRACE:
    python -c "from geobo_amd.build import build; build(extra_flags=('-DGEOBO_DAG_TRACE',))"
    python tools/potrf_dag_trace.py 8448 [out.txt]
(rebuild the default library afterwards: python -m geobo_amd.build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geobo_amd import hip, _lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8448
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
nb = m // 128
g = torch.Generator().manual_seed(0)
B = torch.rand((m, 512), generator=g, dtype=torch.float64).cuda()
S = B @ B.t() / 512 + 0.5 * torch.eye(m, dtype=torch.float64, device="cuda")
Linv = torch.empty((m, m), dtype=torch.float64, device="cuda")
nbytes = _lib.load().geobo_potrf_ws_bytes(m)
ws = torch.zeros(nbytes // 8 + 2, dtype=torch.float64, device="cuda")
for rep in range(3):
    L = S.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.potrf_inv(L, Linv, ws); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)


def tree(n):
    if n <= 1:
        return 0
    mid = n // 2
    if n > 2 and mid & 1:
        mid += 1
    return (n - mid) * mid + tree(mid) + tree(n - mid)


raw = ws.cpu().numpy().view(np.uint8)
off = tree(nb) * 128 * 128 * 8 + 4 * (16 + 3 * nb + nb * nb)
base = ws.data_ptr() + off
pad = (-base) % 8
tr = raw[off + pad: off + pad + nb * nb * 64].view(np.int64).reshape(nb * nb, 8)
ty, ti, tj = tr[:, 0] >> 40, (tr[:, 0] >> 20) & 0xfffff, tr[:, 0] & 0xfffff
t0 = tr[:, 2].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731
claim, acc_done, dep, pub, poll, nseg = us(tr[:, 2]), us(tr[:, 3]), us(tr[:, 4]), us(tr[:, 5]), tr[:, 6] / 100.0, tr[:, 7]
print("m = %d (nb = %d): %.3f ms by events; trace span %.3f ms" % (m, nb, ms, pub.max() / 1000), file=out)
dur = pub - claim
nsteps = np.where(ty == 0, tj, ti - tj) + np.where((ty == 0) & (ti == tj), 0, 1)
busy = dur - poll
print("tasks %d; sum of (published - claimed) %.1f ms over %d workgroups = %.2f of the span; polling %.1f ms = %.2f of that"
      % (len(tr), dur.sum() / 1000, len(np.unique(tr[:, 1] >> 8)), dur.sum() / (len(np.unique(tr[:, 1] >> 8)) * pub.max()),
         poll.sum() / 1000, poll.sum() / dur.sum()), file=out)
big = nsteps >= 8
print("us per 128-deep contraction step (busy time / steps, tasks with >= 8 steps): L tiles %.2f, X tiles %.2f"
      % ((busy[big & (ty == 0)] / nsteps[big & (ty == 0)]).mean() if (big & (ty == 0)).any() else 0,
         (busy[big & (ty == 1)] / nsteps[big & (ty == 1)]).mean() if (big & (ty == 1)).any() else 0), file=out)
diag = (ty == 0) & (ti == tj)
dpub = np.full(nb, np.nan)
dpub[tj[diag]] = pub[diag]
sub = (ty == 0) & (ti == tj + 1)
spub = np.full(nb, np.nan)
spub[tj[sub]] = pub[sub]
print("\ncolumn: diag published (us), delta to the previous column | diag: last input seen -> contraction done -> S parked -> factorised -> "
      "published | sub-diagonal tile: D_j seen -> published | column's last tile published, X row complete", file=out)
for j in range(nb):
    d = diag & (tj == j)
    s_ = sub & (tj == j)
    col = (ty == 0) & (tj == j)
    xr = (ty == 1) & (ti == j)
    tacc = us(tr[d, 1] >> 16)[0]            # diagonal tasks carry "contraction done" in the upper bits of slot 1
    lastin = spub[j - 1] if j else 0.0       # the diagonal tile's last input is the sub-diagonal tile of the column before
    print("%3d %9.1f %7.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f | %9.1f %9.1f" % (
        j, dpub[j], dpub[j] - dpub[j - 1] if j else dpub[j], tacc - lastin, acc_done[d][0] - tacc, dep[d][0] - acc_done[d][0],
        pub[d][0] - dep[d][0], (pub[s_] - dep[s_])[0] if s_.any() else 0, pub[col].max(), pub[xr].max() if xr.any() else 0), file=out)
