"""What does ONE rank of a G-rank sharded step cost?  Measured on one MI355X, collectives replaced by their local part.

    python tools/emulate_rank.py [--of 2,4,8] [--rank 0] [--steps 5] [--size 64] > gpurun_out/emulated_ranks.json

No multi-GPU node has been available to the builder (SCALE_r01 / r02 are `skipped` records), so DESIGN.md section 7's scaling table
rests on what one device can measure: for every G the engine is built as rank r of G with a `sharding.EmulatedGroup` -- the SAME
code path a real rank runs (row-sharded transforms, row-sharded lattice Gram, row-sharded transposed posterior; with
GEOBO_POSTERIOR=dense the round-2 form: crop per destination, all-to-all, column-sharded posterior), with
  all-to-all of A K block-columns   -> this rank's own send buffer stands in for what the peers would send (right size, finite)
  all-gather of AkA row blocks      -> own block in every slot, then the TRUE AkA (kept from a 1-rank step) is put in its place
                                       so that the replicated factorisation is the real one
  all-reduce / all-gather of slices -> identity / own slice
and HIP-event stage times are recorded.  The step time of a real run is then PREDICTED, not measured:
  compute (this tool) + bytes the collectives move / xGMI link rate (7 point-to-point links per GPU; the rate is a parameter: the
  task statement's ~153 GB/s per link, and a conservative 64 GB/s per direction).
The JSON says so in every table it prints.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--of", default="2,4,8")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--drill", type=int, default=50)
    a = ap.parse_args()
    import bench
    from geobo_amd.config_loader import Settings
    from geobo_amd.inversion import Inversion
    from geobo_amd.sharding import EmulatedGroup
    n = a.size
    s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n,
                      gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
    gp_length = np.array([2.00, 2.02, 2.04]) * s.xvoxsize
    torch.cuda.set_device(0)

    def run(inv, grav, mag, loc, drill0, steps, warmup):
        def step():
            inv.engine.clear_operators()
            inv._axes_of = (None, None)
            inv.gp_length = gp_length.copy()
            return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
        for _ in range(warmup):
            step()
        ev = inv.engine.kernel_events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        inv.engine.kernel_events = None
        stages = {}
        for name, fl, alg, valu, e0, e1 in ev:
            stages[name] = stages.get(name, 0.0) + e0.elapsed_time(e1) / steps
        return dt * 1e3, {k: round(v, 3) for k, v in stages.items()}

    # ---- the 1-rank step: reference time, and the true AkA for the emulated ranks ------------------------------------------------
    inv1 = Inversion(settings=s, props=(0, 1))
    grav, mag, loc, drill0 = bench.synthetic_inputs(inv1, a.drill)
    kept = {}
    inv1.engine.aka_hook = lambda AkA: kept.__setitem__("AkA", AkA.clone())
    ms1, st1 = run(inv1, grav, mag, loc, drill0, a.steps, a.warmup)
    inv1.engine.aka_hook = None
    true_AkA = kept["AkA"]
    N, Ms, Ms_pad = inv1.engine.N, inv1.engine.Ms, inv1.engine.Ms_pad
    out = dict(what="PREDICTED, NOT MEASURED: per-rank compute of the sharded 64^3 step measured on ONE MI355X with the collectives "
                    "replaced by their local part (tools/emulate_rank.py), plus a link-rate model for the bytes the collectives move",
               workload="%d^3, Matern-3/2 (2.00,2.02,2.04)x100 m, %d drill rows, P_out = 2" % (n, a.drill),
               one_rank=dict(ms_per_step=round(ms1, 2), stage_ms=st1), ranks={})
    inv1.engine.kernel_events = None
    del inv1
    gc.collect()                     # (engine <-> closure cycles keep the ~100 GB of workspaces alive otherwise)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for G in [int(v) for v in a.of.split(",") if v]:
        r = min(a.rank, G - 1)
        inv = Inversion(settings=s, props=(0, 1), rank=r, world=G, group=EmulatedGroup(r, G), operators="resident")
        inv.sensor_locations = loc
        eng = inv.engine
        eng.aka_hook = lambda AkA: AkA.copy_(true_AkA)
        ms, st = run(inv, grav, mag, loc, drill0, a.steps, a.warmup)
        P_c, nc = 2, eng.nc
        rows_r = Ms // G
        rowp = bool(eng._rowpath)              # row-sharded posterior: no all-to-all, no slices; one all-reduce of 2 P_c N doubles
        # bytes this rank SENDS per collective (fp64)
        a2a_per_peer = rows_r * P_c * nc * 8.0 if (eng.exchange and not rowp) else 0.0   # one operator, one destination
        a2a_total = 2 * (G - 1) * a2a_per_peer                                       # both operators, all peers
        if rowp or (eng.exchange and eng._row_gram()):
            aka_bytes_in = (G - 1) * rows_r * (3.0 if rowp else 4.0) * Ms_pad * 8.0   # all-gather of row blocks: what a rank receives
            aka_kind = "all_gather of AkA row blocks"
        else:
            aka_bytes_in = 2.0 * (G - 1) / G * true_AkA.numel() * 8.0                # ring all-reduce: 2 (G-1)/G S per rank
            aka_kind = "all_reduce of the partial AkA"
        slices = 0.0 if rowp else (G - 1) * P_c * nc * 8.0 * 2                        # mu and var slices received
        allred = 1.0 * P_c * N * 8.0 if rowp else 0.0                                 # partial sums of squares (every rank forms the mean whole)
        pred = {}
        for label, link in (("153 GB/s per link (task statement)", 153e9), ("64 GB/s per link and direction (conservative)", 64e9)):
            links = min(G - 1, 7)
            t_a2a = 2 * a2a_per_peer / link * 1e3                                    # every peer over its own link, both operators in sequence
            t_aka = aka_bytes_in / (links * link) * 1e3
            t_sl = (slices + 2.0 * (G - 1) / G * allred) / (links * link) * 1e3 + 0.05
            # shared communicator (the default): the all-to-all runs under the second operator's transforms and the row Gram; what does
            # not fit there is exposed in front of the all-gather
            hide = st.get("spectral_product", 0.0) / 2 + st.get("aka_lattice", 0.0) if (eng.exchange and not rowp) else 0.0
            exposed = max(0.0, t_a2a - hide) + t_aka + t_sl
            pred[label] = dict(all_to_all_ms=round(t_a2a, 2), aka_collective_ms=round(t_aka, 2), slices_or_all_reduce_ms=round(t_sl, 2),
                               step_ms_no_overlap=round(ms + t_a2a + t_aka + t_sl, 1), step_ms=round(ms + exposed, 1),
                               speedup_vs_one_rank=round(ms1 / (ms + exposed), 2))
        out["ranks"][str(G)] = dict(rank=r, route=eng.route.describe(), row_exchange=bool(eng.exchange and not rowp), row_posterior=rowp,
                                    row_gram=bool(rowp or eng._row_gram()),
                                    compute_ms_per_step_measured=round(ms, 2), stage_ms_measured=st,
                                    replicated_ms=round(st.get("potrf_inv", 0.0), 2),
                                    bytes_sent_all_to_all=a2a_total, aka_collective=aka_kind, bytes_received_aka=aka_bytes_in,
                                    predicted=pred, memory_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))
        print("G = %d: compute %.1f ms (1 rank: %.1f ms); stages %s" % (G, ms, ms1, st), file=sys.stderr, flush=True)
        eng.aka_hook = None
        del inv, eng
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
