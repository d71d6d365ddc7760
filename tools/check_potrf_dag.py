"""geobo_potrf_inv, persistent tile-DAG form against torch and against the stream schedule (GEOBO_POTRF=streams):
    python tools/check_potrf_dag.py [m ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip

sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 2304, 4224, 8448]
for m in sizes:
    g = torch.Generator().manual_seed(m)
    B = torch.rand((m, 512), generator=g, dtype=torch.float64).cuda()
    S = B @ B.t() / 512 + 0.5 * torch.eye(m, dtype=torch.float64, device="cuda")
    Lref = torch.linalg.cholesky(S)
    eye = torch.eye(m, dtype=torch.float64, device="cuda")
    Linv = torch.empty((m, m), dtype=torch.float64, device="cuda")
    ws = torch.empty(hip.potrf_ws_doubles(m), dtype=torch.float64, device="cuda")
    ctx = hip.PotrfContext()
    for mode in ("dag", "streams"):
        os.environ["GEOBO_POTRF"] = mode
        ts = []
        for rep in range(4):
            L = S.clone()
            Linv.fill_(float("nan"))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, info = hip.potrf_inv(L, Linv, ws, ctx=ctx)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        eL = (torch.tril(L) - Lref).abs().max().item()
        eI = (Linv @ Lref - eye).abs().max().item()
        up = torch.triu(Linv, 1).abs().max().item()
        print("m=%5d %-7s info=%d  |L-Lref|=%.2e  |Linv L - I|=%.2e  upper(Linv)=%.1e  ms=%s" % (
            m, mode, int(info.item()), eL, eI, up, " ".join("%.2f" % t for t in ts)), flush=True)
    # run to run: bit-identical
    os.environ["GEOBO_POTRF"] = "dag"
    L1 = S.clone(); X1 = torch.empty_like(S); hip.potrf_inv(L1, X1, ws)
    L2 = S.clone(); X2 = torch.empty_like(S); hip.potrf_inv(L2, X2, ws)
    torch.cuda.synchronize()
    print("        bit-identical repeat:", torch.equal(torch.tril(L1), torch.tril(L2)) and torch.equal(X1, X2), flush=True)
