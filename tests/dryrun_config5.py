"""BASELINE config 5 on ONE MI355X: rank r of an 8-rank column-sharded run of the 128^3 x 3-property inversion
(fp32 kernel assembly + fp64 Cholesky), with independent CPU-oracle spot checks of what the rank produced.

    python tests/dryrun_config5.py [--size 128] [--world 8] [--rank 0] [--no-oracle] > gpurun_out/config5_rank0.json

What runs is exactly the engine's per-rank step in the column-sharded form that needs no peer (forward passes of every sensor
row replicated, backward passes cropped to the rank's y-slabs): streamed forward operators (A is 275 GB per type at 128^3 and is
never resident), spectral A K product written as fp32 (the rank's shard: 33024 x 3 x 262144 fp32 = 104 GB), partial AkA by
column panels (operator columns regenerated per panel), Cholesky + L^-1 at M_pad = 33024, posterior mean / variance of the
rank's 3 x 262144 voxel-property columns.  The two collectives are the only thing missing: the all-reduce of AkA is replaced by
the identity, so the matrix that is factorised here is a stand-in with the right size (the partial AkA of one rank is not
positive definite on its own) -- timings of the factorisation and of the posterior sweep do not depend on the values.

Oracle spot checks (oracle/geobo_oracle.py, nothing borrowed from the device): forward-operator rows of a few sensors, the same
rows of A K through the oracle's FFT form of the covariance product on this rank's columns, and entries of the rank's partial AkA.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# one process plays the ranks: the column-sharded form that needs no peer (forward passes replicated, backward passes cropped);
# a real multi-rank run of these modes uses the chunked row exchange instead (engine._exchange_chunked)
os.environ["GEOBO_SPECTRAL_EXCHANGE"] = "0"


def sequential(a):
    """The whole 8-rank job on one GPU: pass 1 = every shard's A K and partial AkA (summed), factorisation, pass 2 = every shard's
    A K again (104 GB per shard: only one fits) and its posterior columns.  Checks: 0 < var <= 1, the posterior mean reproduces
    the survey to the noise level (operator rows streamed once more), checksums."""
    import geobo_amd.engine as E
    from geobo_amd import geometry, hip
    from geobo_amd.config_loader import Settings
    from geobo_amd.sharding import shard_columns
    n, G = a.size, a.world
    s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n,
                      gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
    torch.cuda.set_device(0)
    eng = E.PosteriorEngine(s, rank=0, world=G, assembly="f32", operators="streamed")
    props, lengths, W = (0, 1, 2), [200.0, 202.0, 204.0], E.weight_matrix(s.gp_coeff)
    N, Ms = eng.N, eng.Ms
    # synthetic truth (the reference's cylinders + trend, as bench.py) and survey = A rho / A chi through streamed operator rows
    xc, yc, zc = geometry.centre_axes(s)
    C = geometry.expand(xc, yc, zc)
    x3, y3, z3 = C[0], C[1], C[2]
    rad = s.yLcube / 18.
    rho = x3 * 0. + 0.1
    rho[((y3 - s.yLcube / 4. - rad) ** 2) + ((z3 + s.zLcube / 4 - rad) ** 2) <= rad ** 2] = 1.
    rho[((y3 - s.yLcube / 1.3 - rad) ** 2) + ((z3 + s.zLcube / 4 - rad) ** 2) <= rad ** 2] = 1.
    rho[(x3 < s.xLcube / 5.) | (x3 > s.xLcube * 4. / 5.)] = 0.1
    rho = rho + 0.02 * (x3 / s.xLcube + 2. * y3 / s.yLcube - z3 / s.zLcube)
    chi = s.gp_coeff[1] * rho
    xs = np.linspace(0.5, n - 0.5, n) * s.xvoxsize
    X, Y, Z = np.meshgrid(xs, xs, s.zmax + s.zoff)
    loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
    sel = np.sort(np.random.default_rng(2020).choice(N, a.drill, replace=False))
    sel_t = torch.as_tensor(sel, device="cuda")
    A_g, A_m = eng.operator("grav", loc, B=s.magneticField * 0.), eng.operator("magn", loc, B=s.magneticField)

    def apply(op, v):      # op @ v with rows generated in batches
        vd = hip.to_dev(np.asarray(v).reshape(-1))
        buf = eng._workspace2d("op_rows", 256, eng.N_pad)
        out = torch.empty(Ms, dtype=torch.float64, device="cuda")
        for r0 in range(0, Ms, 256):
            R = min(256, Ms - r0)
            out[r0:r0 + R] = op.rows_into(buf, r0, R)[:, :N] @ vd
        return out.cpu().numpy()
    z = lambda v: (v - v.mean()) / v.std()
    grav = apply(A_g, rho).astype(np.float32).astype(np.float64)
    mag = apply(A_m, chi).astype(np.float32).astype(np.float64)
    drill = rho.reshape(-1)[sel]
    y_g, y_m, y_d = z(grav), z(mag), z(drill)
    state = dict(acc=None, last=False)

    def sum_over_ranks(t, world, group=None):
        state["acc"] = t.clone() if state["acc"] is None else state["acc"].add_(t)
        if state["last"]:
            t.copy_(state["acc"])
        return t
    E.allreduce_sum_ = sum_over_ranks
    t_all = time.perf_counter()
    times = dict(ak=[], aka=[], post=[])

    def shard(r):
        eng.rank = r
        eng.c0, eng.c1 = shard_columns(eng.N_pad, G, r)
        eng.nc = eng.c1 - eng.c0

    def timed(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        times[key].append(time.perf_counter() - t0)
        return out
    for r in range(G):
        shard(r)
        state["last"] = r == G - 1
        AK, M_pad = timed("ak", lambda: eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, props))
        AkA = timed("aka", lambda: eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, props))
        print("pass 1 shard %d: A K %.1f s, partial AkA %.1f s" % (r, times["ak"][-1], times["aka"][-1]), file=sys.stderr, flush=True)
    state["acc"] = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Linv, info = hip.potrf_inv(AkA, eng._workspace("Linv", (M_pad, M_pad)), eng._workspace("potrf_ws", (hip.potrf_ws_doubles(M_pad),)),
                               ctx=hip.PotrfContext())
    assert int(info.item()) == 0, "AkA not positive definite"
    u, stats = hip.trmv_stats(Linv, eng._pad_y(y_g, y_m, y_d, M_pad), AkA)
    torch.cuda.synchronize()
    t_chol = time.perf_counter() - t0
    st = stats.cpu().numpy()
    logl = -0.5 * (st[0] + st[1] + N * np.log(2 * np.pi))
    mu, var = np.full(3 * N, np.nan), np.full(3 * N, np.nan)
    Mv = 2 * eng.Ms_pad + sel.size
    for r in range(G):
        shard(r)
        AK, _ = timed("ak", lambda: eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, props))
        ncols = AK.shape[1]
        pw = max(128, min(ncols, int((3 << 30) // (8 * M_pad)) // 128 * 128))

        def post():
            ws = eng._workspace("post_ws", (hip.posterior_ws_doubles(M_pad, pw),))
            parts = [hip.posterior_reduce(Linv, eng._panel64("post_panel64", AK[:, cs:min(ncols, cs + pw)], cols=pw), u, 1.0, ws, m_valid=Mv)
                     for cs in range(0, ncols, pw)]
            return torch.cat([p[0] for p in parts]).cpu().numpy(), torch.cat([p[1] for p in parts]).cpu().numpy()
        m_, v_ = timed("post", post)
        for jj in range(3):
            mu[jj * N + eng.c0:jj * N + eng.c1] = m_[jj * eng.nc:(jj + 1) * eng.nc]
            var[jj * N + eng.c0:jj * N + eng.c1] = v_[jj * eng.nc:(jj + 1) * eng.nc]
        print("pass 2 shard %d: A K %.1f s, posterior %.1f s" % (r, times["ak"][-1], times["post"][-1]), file=sys.stderr, flush=True)
    total = time.perf_counter() - t_all
    rg = apply(A_g, mu[:N]) - y_g
    rm = apply(A_m, mu[N:2 * N]) - y_m
    out = dict(what="BASELINE config 5 end to end on ONE MI355X: %d^3 voxels x 3 properties, fp32 assembly + fp64 Cholesky, the %d column "
                    "shards run one after the other (partial AkA summed on the device)" % (n, G),
               N_voxels=N, M_rows=2 * Ms + sel.size, total_seconds=total, voxel_properties_per_s=3.0 * N / total,
               per_shard_seconds=dict(ak_pass1_and_2=[round(v, 2) for v in times["ak"]], partial_aka=[round(v, 2) for v in times["aka"]],
                                      posterior=[round(v, 2) for v in times["post"]]),
               cholesky_linv_trmv_seconds=t_chol, logl=float(logl),
               checks=dict(var_min=float(var.min()), var_max=float(var.max()), finite=bool(np.isfinite(mu).all() and np.isfinite(var).all()),
                           rms_residual_grav=float(np.sqrt(np.mean(rg ** 2))), rms_residual_magn=float(np.sqrt(np.mean(rm ** 2))),
                           drill_rms_residual=float(np.sqrt(np.mean((mu[2 * N + sel] - y_d) ** 2)))),
               checksums=dict(sum_abs_mu=[float(np.abs(mu[j * N:(j + 1) * N]).sum()) for j in range(3)],
                              sum_var=[float(var[j * N:(j + 1) * N].sum()) for j in range(3)]),
               max_memory_allocated_GB=torch.cuda.max_memory_allocated() / 1e9)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--drill", type=int, default=50)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="run ALL ranks' shards one after the other on this GPU, summing the partial "
                    "AkA (what the all-reduce does) -- a complete, correct inversion of the cube on one device")
    a = ap.parse_args()
    if a.sequential:
        return sequential(a)
    import geobo_amd.engine as E
    from geobo_amd import hip
    from geobo_amd.config_loader import Settings
    n = a.size
    s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n,
                      gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))
    E.allreduce_sum_ = lambda t, world, group=None: t            # the one collective of this stage: identity in the dry run
    torch.cuda.set_device(0)
    eng = E.PosteriorEngine(s, rank=a.rank, world=a.world, assembly="f32", operators="streamed")
    assert eng.use_spectral and not eng.exchange
    props = (0, 1, 2)
    lengths = [200.0, 202.0, 204.0]
    W = E.weight_matrix(s.gp_coeff)
    xs = np.linspace(0.5, n - 0.5, n) * s.xvoxsize
    X, Y, Z = np.meshgrid(xs, xs, s.zmax + s.zoff)
    loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
    sel = np.sort(np.random.default_rng(2020).choice(eng.N, a.drill, replace=False))
    sel_t = torch.as_tensor(sel, device="cuda")
    # The stages below are what a rank repeats every step with its workspaces in place.  The first touch of the 104 GB A K shard is
    # not part of that: the driver clears recycled VRAM on allocation (4 s on a fresh box, up to 10 s right after another process
    # freed the memory) -- allocate it before the clock starts, as an engine that has done one step already has.
    eng._workspace2d("AK", hip.pad_m(2 * eng.Ms_pad + a.drill), len(props) * eng.nc, dtype=hip.F32).zero_()
    ev = eng.kernel_events = []
    stamps = {}

    def stage(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        stamps[name] = time.perf_counter() - t0
        print("%-22s %8.2f s   (max alloc %.1f GB)" % (name, stamps[name], torch.cuda.max_memory_allocated() / 1e9), file=sys.stderr, flush=True)
        return r

    A_g, A_m = stage("operators (plans, Q)", lambda: (eng.operator("grav", loc, B=s.magneticField * 0.), eng.operator("magn", loc, B=s.magneticField)))
    AK, M_pad = stage("A K (spectral, fp32)", lambda: eng._assemble_AK(A_g, A_m, sel_t, lengths, W, "matern32", 1.0, props))
    assert AK.dtype == torch.float32 and AK.shape == (M_pad, 3 * eng.nc)
    AkA = stage("partial AkA (panels)", lambda: eng._assemble_AkA(AK, M_pad, A_g, A_m, sel_t, lengths, "matern32", 1.0, s.gp_err, props))
    checks = {}
    if not a.no_oracle:
        from oracle import geobo_oracle as O
        G = O.Grid(nx=n, ny=n, nz=n, xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, kernelfunc="matern32")
        edges = G.edges()
        sens = [n + 3, (n // 2) * n + n // 3, n * n - 2]
        Wn = O.weight_matrix(s.gp_coeff)
        ln = np.array(lengths)
        t0 = time.perf_counter()
        Ao = {0: O.a_sens(G, G.B * 0., loc, edges, "grav", rows=sens), 1: O.a_sens(G, G.B, loc, edges, "magn", rows=sens)}
        c0, c1 = eng.c0, eng.c1
        errs_ak, errs_aka, errs_a = [], [], []
        abuf = eng._op_rows_buffer()
        for s_, op in ((0, A_g), (1, A_m)):                                   # forward-operator rows as the streamed operator generates them
            for k, r in enumerate(sens):
                got = op.rows_into(abuf, r, 1)[0, :eng.N].cpu().numpy()
                errs_a.append(float(np.abs(got - Ao[s_][k]).max() / np.abs(Ao[s_][k]).max()))
        AkA_l = torch.tril(AkA)
        for s_ in (0, 1):
            for k, r in enumerate(sens):
                w = {j: O.ak_row_fft(G, Ao[s_][k], "matern32", ln, Wn, s_, j) for j in props}
                for jj, j in enumerate(props):
                    got = AK[s_ * eng.Ms_pad + r, jj * eng.nc:(jj + 1) * eng.nc].double().cpu().numpy()
                    want = w[j][c0:c1].astype(np.float32).astype(np.float64)   # fp32 storage of the exact row
                    errs_ak.append(float(np.abs(got - want).max() / np.abs(want).max()))
                # this rank's contribution to AkA[row, col] for the checked sensors (lower triangle: col <= row)
                row = s_ * eng.Ms_pad + r
                for t_, Aot in ((0, Ao[0]), (1, Ao[1])):
                    for k2, r2 in enumerate(sens):
                        col = t_ * eng.Ms_pad + r2
                        if col > row:
                            continue
                        want = float(w[t_][c0:c1] @ Aot[k2][c0:c1])
                        got = float(AkA_l[row, col].item()) - (0.1 ** 2 if col == row else 0.0)
                        errs_aka.append(abs(got - want) / max(abs(want), 1e-300))
        checks = dict(sensors=sens, a_sens_rows_vs_oracle=max(errs_a), ak_rows_vs_oracle_fp32_rounded=max(errs_ak), partial_aka_entries_rel=max(errs_aka),
                      oracle_seconds=time.perf_counter() - t0,
                      note="A K rows: device fp32 shard vs the oracle's exact row rounded to fp32 (normwise); AkA: this rank's partial sum "
                           "vs oracle operator rows x oracle A K rows on the rank's columns (fp32-storage accuracy expected)")
        print("oracle checks:", checks, file=sys.stderr, flush=True)
    # stand-in SPD matrix of the right size for the factorisation (see module docstring)
    AkA.zero_()
    AkA.diagonal().fill_(1.01)
    ctx = hip.PotrfContext()
    Linv, info = stage("Cholesky + L^-1", lambda: hip.potrf_inv(AkA, eng._workspace("Linv", (M_pad, M_pad)),
                                                                eng._workspace("potrf_ws", (hip.potrf_ws_doubles(M_pad),)), ctx=ctx))
    assert int(info.item()) == 0
    y = torch.randn(M_pad, dtype=torch.float64, device="cuda")
    u, _ = hip.trmv_stats(Linv, y, AkA)
    Mv = 2 * eng.Ms_pad + sel.size
    ncols = AK.shape[1]
    pw = max(128, min(ncols, int((3 << 30) // (8 * M_pad)) // 128 * 128))

    def post():
        ws = eng._workspace("post_ws", (hip.posterior_ws_doubles(M_pad, pw),))
        return [hip.posterior_reduce(Linv, eng._panel64("post_panel64", AK[:, cs:min(ncols, cs + pw)], cols=pw), u, 1.0, ws, m_valid=Mv)
                for cs in range(0, ncols, pw)]
    parts = stage("posterior (panels)", post)
    post_finite = bool(all(torch.isfinite(p[0]).all().item() and torch.isfinite(p[1]).all().item() for p in parts))
    Mu = 2 * eng.Ms + sel.size
    alg = (1.0 * Mu * Mu + 4.0 * Mu) * ncols
    stages = {}
    for name, fl, algf, valu, e0, e1 in ev:
        d = stages.setdefault(name, dict(seconds=0.0, flop=0.0))
        d["seconds"] += e0.elapsed_time(e1) * 1e-3
        d["flop"] += fl
    ws_gb = {k: round(v.numel() * v.element_size() / 1e9, 2) for k, v in eng._ws.items()}
    sp = eng._spectral
    ws_gb.update({"spectral:" + k: round(v.numel() * 8 / 1e9, 2) for k, v in sp._bufs.items()})
    step = sum(v for k, v in stamps.items())
    out = dict(what="BASELINE config 5 dry run: rank %d of %d, %d^3 voxels x 3 properties, fp32 assembly, streamed operators" % (a.rank, a.world, n),
               N_voxels=eng.N, M_rows=Mu, M_pad=M_pad, shard_columns=[eng.c0, eng.c1], ak_shard_shape=list(AK.shape), ak_dtype="float32",
               wall_seconds=stamps, rank_step_seconds=step,
               posterior_tflops_algorithmic=alg / stamps["posterior (panels)"] / 1e12,
               kernel_stage_seconds={k: round(v["seconds"], 3) for k, v in stages.items()},
               kernel_stage_tflops_executed={k: round(v["flop"] / v["seconds"] / 1e12, 2) for k, v in stages.items() if v["flop"] > 0},
               memory_map_GB=dict(sorted(ws_gb.items(), key=lambda kv: -kv[1])), max_memory_allocated_GB=torch.cuda.max_memory_allocated() / 1e9,
               oracle_checks=checks, posterior_finite=post_finite,
               missing="the all-reduce of AkA (identity here: the factorised matrix is a stand-in of the right size) and the all-gather of "
                       "the mu / var slices; with the row exchange (>= 4 ranks, all_to_all of A K block-columns) a rank transforms 1/8 of "
                       "the sensor rows instead of all of them: the 'A K' stage divides by ~8 there")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
