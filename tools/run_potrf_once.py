"""geobo_potrf_inv at the 64^3 headline size (M_pad = 8448), with and without the fork context (look-ahead + concurrent L^-1
subtrees); for rocprofv3 --kernel-trace timelines:  python tools/run_potrf_once.py [m] [ctx|noctx]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geobo_amd import hip

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8448
modes = [sys.argv[2]] if len(sys.argv) > 2 else ["noctx", "ctx"]
g = torch.Generator().manual_seed(0)
B = torch.rand((m, 512), generator=g, dtype=torch.float64).cuda()
S = B @ B.t() / 512 + 0.5 * torch.eye(m, dtype=torch.float64, device="cuda")
Linv = torch.empty((m, m), dtype=torch.float64, device="cuda")
ws = torch.empty(hip.potrf_ws_doubles(m), dtype=torch.float64, device="cuda")
for mode in modes:
    ctx = hip.PotrfContext() if mode == "ctx" else None
    for rep in range(3):
        L = S.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); hip.potrf_inv(L, Linv, ws, ctx=ctx); e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("potrf_inv m=%d %-5s: %.2f ms on the stream, %.2f ms of host enqueue time" % (m, mode, ms, 1e3 * t_host))
    # (last line in the form tools/pmc_collect.py parses: seconds, rate, executed flop = m^3/3 for L + m^3/3 for L^-1)
    print("potrf_inv m=%d %s: %.6f s, %.2f TF/s flop %d" % (m, mode, ms * 1e-3, 2.0 * m ** 3 / 3.0 / (ms * 1e-3) / 1e12, int(2.0 * m ** 3 / 3.0)))
