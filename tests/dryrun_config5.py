"""BASELINE config 5 on ONE MI355X: rank r of an 8-rank run of the 128^3 x 3-property inversion (fp32 kernel assembly + fp64
Cholesky) in the ROW form (geobo_amd/rowform.py), with independent CPU-oracle checks of what the rank produced -- the posterior
mean and the rank's partial sums of squares BY VALUE included.

    python tests/dryrun_config5.py [--size 128] [--world 8] [--rank 0] [--no-oracle] > gpurun_out/config5_rank0.json
    python tests/dryrun_config5.py --sequential          # the whole 8-rank job on this one GPU (complete posterior cubes)

What runs is exactly the engine's per-rank step (sharding.EmulatedGroup: every collective replaced by its local part): streamed
forward operators (one is 275 GB at 128^3: rows are generated a transform batch at a time), the rank's 2 x Ms/G sensor rows through
the covariance product for ALL voxels in chunks and straight on through the lattice Gram (A K is never held; covariance tables
rounded through fp32), Cholesky + L^-1 at M_pad = 33024, the transposed posterior on the rank's rows of L^-1.  To factorise the REAL
matrix on one device the other ranks' row blocks of AkA are computed here as well, one rank after the other (what the all-gather
would have delivered); only `--rank`'s stages are timed as "the rank step".

Oracle checks (oracle/geobo_oracle.py):
  * nothing borrowed from the device: forward-operator rows of a few sensors; the same rows of A K through the oracle's FFT form of
    the covariance product; entries of AkA (oracle operator rows x oracle A K rows + sigma^2) against the matrix the device factorised;
  * given the device's factor L, the data vector u = L^-1 y and the device's forward operators (whose rows were just checked) applied
    to ORACLE covariance columns: V[:, q] = L^-1 (A3 K)[:, q] by scipy's triangular solve on the host for a sample of voxel columns q
    and all three property blocks -> posterior mean mu[q] = V[:, q] . u and the rank's partial sum of squares
    sum_(m in the rank's rows) V[m, q]^2, against what the spectral / lattice / row-sharded path produced.  That path shares nothing
    with this check but L: no transform, no stencil table, no Toeplitz stage, no reduction kernel.
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


def settings(n):
    from geobo_amd.config_loader import Settings
    return Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n, zNcube=n,
                         gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc="matern32", XMAG=0, YMAG=0, ZMAG=1))


def survey(eng, s, n, md):
    """Synthetic truth (the reference's cylinders + trend, as bench.py), survey = A rho / A chi through streamed operator rows."""
    from geobo_amd import geometry
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
    sel = np.sort(np.random.default_rng(2020).choice(eng.N, md, replace=False))
    return rho, chi, loc, sel


def apply_rows(eng, op, V):
    """op @ V for a streamed or resident operator, V (N x c) on the device: rows generated 256 at a time."""
    from geobo_amd.engine import StreamedOperator
    if not isinstance(op, StreamedOperator):
        return op[:eng.Ms, :eng.N] @ V
    buf = eng._workspace2d("op_rows_check", 256, eng.N_pad)
    out = torch.empty((eng.Ms, V.shape[1]), dtype=torch.float64, device=V.device)
    for r0 in range(0, eng.Ms, 256):
        R = min(256, eng.Ms - r0)
        out[r0:r0 + R] = op.rows_into(buf, r0, R)[:, :eng.N] @ V
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--drill", type=int, default=50)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--repeat", type=int, default=1, help="passes of the rank's step; the last one is reported, the first as `cold`")
    ap.add_argument("--sequential", action="store_true", help="every rank's posterior share as well: the complete cubes of the 8-rank job "
                    "on this one GPU (0 < var <= 1, data residuals, checksums)")
    a = ap.parse_args()
    import geobo_amd.engine as E
    from geobo_amd import hip
    from geobo_amd.sharding import EmulatedGroup
    n, G = a.size, a.world
    s = settings(n)
    os.environ.setdefault("GEOBO_ROWS", "1")            # (sizes below the planner's threshold, e.g. --size 32 in the test tier)
    torch.cuda.set_device(0)
    eng = E.PosteriorEngine(s, rank=a.rank, world=G, group=EmulatedGroup(a.rank, G), assembly="f32", operators="streamed")
    assert eng.route.family == "rows", eng.route
    props, lengths, W, name = (0, 1, 2), [200.0, 202.0, 204.0], E.weight_matrix(s.gp_coeff), "matern32"
    eng._W = W
    N, Ms, Msp = eng.N, eng.Ms, eng.Ms_pad
    rho, chi, loc, sel = survey(eng, s, n, a.drill)
    sel_t = torch.as_tensor(sel, device="cuda") if sel.size else None
    ev = eng.kernel_events = []
    stamps = {}

    def stage(key, fn, quiet=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        stamps[key] = stamps.get(key, 0.0) + time.perf_counter() - t0
        if not quiet:
            print("%-34s %8.2f s   (max alloc %.1f GB)" % (key, stamps[key], torch.cuda.max_memory_allocated() / 1e9), file=sys.stderr, flush=True)
        return r

    peers, cold = None, None
    for rep in range(a.repeat):
        # (--repeat 2: the first pass pays what a process pays once -- code objects of every kernel, the allocator's first 100 GB, the
        # plans -- and is reported as `cold`; the step a fit loop repeats is the last pass)
        if rep:
            cold = dict(stamps)
            stamps.clear()
            del ev[:]
        A_g, A_m = stage("operators (plans, Q, slabs)", lambda: (eng.operator("grav", loc, B=s.magneticField * 0.), eng.operator("magn", loc, B=s.magneticField)))
        eng._spectral_product()
        assert eng._rows_ok(A_g, A_m)
        eng._rowpath = True
        z = lambda v: (v - v.mean()) / v.std()
        vd = lambda v: hip.to_dev(np.asarray(v).reshape(-1, 1))
        grav = apply_rows(eng, A_g, vd(rho))[:, 0].cpu().numpy().astype(np.float32).astype(np.float64)
        mag = apply_rows(eng, A_m, vd(chi))[:, 0].cpu().numpy().astype(np.float32).astype(np.float64)
        y_g, y_m, y_d = z(grav), z(mag), (z(rho.reshape(-1)[sel]) if sel.size else np.zeros(0))
        M_pad = hip.pad_m(2 * Msp + sel.size)
        # ---- AkA: this rank's row blocks (timed), then the other ranks' (what the all-gather delivers) -------------------------------
        blocks, drill = [None] * G, None
        order = [a.rank] + [r for r in range(G) if r != a.rank]
        for r in order:
            if r != a.rank and peers is not None:
                blocks[r] = peers[r]                         # (a repeat: what the all-gather delivers is the first pass's)
                continue
            eng.rank = r
            key = "A K -> AkA row blocks (this rank)" if r == a.rank else "AkA row blocks of the %d peers" % (G - 1)
            lo, dr = stage(key, lambda: eng._rows_aka_local(props, sel_t, lengths, W, name, 1.0), quiet=r != a.rank)
            blocks[r] = lo.clone() if G > 1 else lo
            drill = dr
        eng.rank = a.rank
        AkA = eng._workspace("AkA", (M_pad, M_pad))
        AkA.zero_()
        eng._rows_aka_place(AkA, blocks, drill, sel_t)
        eng._finish_AkA(AkA, M_pad, sel_t, lengths, name, 1.0, s.gp_err)
        peers = blocks
        del blocks
        AkA_low = None
        if not a.no_oracle:
            AkA_low = {}          # entries the oracle will look at, read before the factorisation overwrites the matrix
        sens = [n + 3, (n // 2) * n + n // 3, n * n - 2]
        if AkA_low is not None:
            for s_ in (0, 1):
                for r in sens:
                    row = s_ * Msp + r
                    for t_ in (0, 1):
                        for r2 in sens:
                            col = t_ * Msp + r2
                            if col <= row:
                                AkA_low[(row, col)] = float(AkA[row, col].item())
        ctx = hip.PotrfContext()
        Linv, info = stage("Cholesky + L^-1", lambda: hip.potrf_inv(AkA, eng._workspace("Linv", (M_pad, M_pad)),
                                                                    eng._workspace("potrf_ws", (hip.potrf_ws_doubles(M_pad),)), ctx=ctx))
        assert int(info.item()) == 0, "AkA not positive definite"
        u, stats = hip.trmv_stats(Linv, eng._pad_y(y_g, y_m, y_d, M_pad), AkA)
        st = stats.cpu().numpy()
        logl = -0.5 * (st[0] + st[1] + N * np.log(2 * np.pi))
        # ---- posterior: this rank's share (timed); --sequential: every rank's ----------------------------------------------------------
        mu_t, var_t = stage("posterior (this rank's rows)", lambda: eng._posterior_rows(Linv, u, sel_t, lengths, W, name, 1.0, props, M_pad))
        mu = mu_t.cpu().numpy().reshape(3, N)
        part = (1.0 - var_t).cpu().numpy().reshape(3, N)          # the rank's partial sums of squares (the all-reduce is the identity here)
        total_ss = part.copy()
        if a.sequential and rep == a.repeat - 1:
            for r in range(G):
                if r != a.rank:
                    eng.rank = r
                    _, v_r = stage("posterior shares of the %d peers" % (G - 1), lambda: eng._posterior_rows(Linv, u, sel_t, lengths, W, name, 1.0, props, M_pad),
                                   quiet=True)
                    total_ss += (1.0 - v_r).cpu().numpy().reshape(3, N)
            eng.rank = a.rank
        rank_step = sum(stamps[k] for k in ("operators (plans, Q, slabs)", "A K -> AkA row blocks (this rank)", "Cholesky + L^-1",
                                             "posterior (this rank's rows)"))
    checks = {}
    if not a.no_oracle:
        from scipy.linalg import solve_triangular
        from oracle import geobo_oracle as O
        Gd = O.Grid(nx=n, ny=n, nz=n, xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, kernelfunc=name)
        edges = Gd.edges()
        Wn, ln = O.weight_matrix(s.gp_coeff), np.array(lengths)
        t0 = time.perf_counter()
        f32 = lambda v: v.astype(np.float32).astype(np.float64)
        # (1) nothing from the device: operator rows, rows of A K (FFT form, exact covariance), entries of AkA
        Ao = {0: O.a_sens(Gd, Gd.B * 0., loc, edges, "grav", rows=sens), 1: O.a_sens(Gd, Gd.B, loc, edges, "magn", rows=sens)}
        errs_a, errs_ak, errs_aka = [], [], []
        abuf = eng._workspace2d("op_rows_check", 256, eng.N_pad)
        for s_, op in ((0, A_g), (1, A_m)):
            for k, r in enumerate(sens):
                got = op.rows_into(abuf, r, 1)[0, :N].cpu().numpy()
                errs_a.append(float(np.abs(got - Ao[s_][k]).max() / np.abs(Ao[s_][k]).max()))
        sp = eng._spectral
        one = torch.zeros((sp.R, eng.N_pad), dtype=torch.float64, device="cuda")
        outs = [torch.empty((sp.R, N), dtype=torch.float64, device="cuda") for _ in props]
        for s_, op in ((0, A_g), (1, A_m)):
            gens = [eng._gens[(s_, j)] for j in props]
            for k, r in enumerate(sens):
                w = {j: O.ak_row_fft(Gd, Ao[s_][k], name, ln, Wn, s_, j) for j in props}
                one[0, :N] = torch.as_tensor(Ao[s_][k], device="cuda")
                sp.product(one, 1, gens, outs)                                   # the device's covariance product on the ORACLE's operator row
                for jj, j in enumerate(props):
                    got = outs[jj][0].cpu().numpy()
                    errs_ak.append(float(np.abs(got - w[j]).max() / np.abs(w[j]).max()))
                row = s_ * Msp + r
                for t_ in (0, 1):
                    for k2, r2 in enumerate(sens):
                        col = t_ * Msp + r2
                        if col <= row:
                            want = float(w[t_] @ Ao[t_][k2]) + (0.1 ** 2 if col == row else 0.0)
                            errs_aka.append(abs(AkA_low[(row, col)] - want) / max(abs(want), 1e-300))
        # (2) given L, u and the device's operators: posterior columns by scipy's triangular solve on oracle covariance columns
        P3 = O.grid_points((n, n, n), (s.xvoxsize, s.yvoxsize, s.zvoxsize))
        plane = n * n
        qs = [3 * plane // 2 + 5, (n // 2) * plane + (n // 3) * n + n // 5, (n - 1) * plane + plane // 2 + 7, 17, N - 9, (n // 4) * plane + 11 * n + n - 1]
        rows_all = np.r_[0:Ms, Msp:Msp + Ms, 2 * Msp:2 * Msp + sel.size]
        L = torch.tril(AkA).cpu().numpy()[np.ix_(rows_all, rows_all)]
        uh = u.cpu().numpy()[rows_all]
        AKc = np.empty((rows_all.size, len(props) * len(qs)))
        for jj, j in enumerate(props):
            D2 = O.sqdist(P3, P3[qs])                                          # (N, len(qs))
            cols = slice(jj * len(qs), (jj + 1) * len(qs))
            AKc[:Ms, cols] = apply_rows(eng, A_g, hip.to_dev(f32(O.k_block(name, D2, ln, Wn, 0, j)))).cpu().numpy()
            AKc[Ms:2 * Ms, cols] = apply_rows(eng, A_m, hip.to_dev(f32(O.k_block(name, D2, ln, Wn, 1, j)))).cpu().numpy()
            if sel.size:
                AKc[2 * Ms:, cols] = f32(O.k_block(name, D2[sel], ln, Wn, 2, j))
        V = solve_triangular(L, AKc, lower=True)
        mu_o = V.T @ uh
        rows_r = Ms // G
        dper = -(-sel.size // G)
        d0 = min(sel.size, a.rank * dper)
        own = np.r_[a.rank * rows_r:(a.rank + 1) * rows_r, Ms + a.rank * rows_r:Ms + (a.rank + 1) * rows_r,
                    2 * Ms + d0:2 * Ms + min(sel.size, d0 + dper)]
        ss_own = np.einsum("mq,mq->q", V[own], V[own])
        ss_all = np.einsum("mq,mq->q", V, V)
        e_mu, e_ss, e_var = [], [], []
        for jj in range(len(props)):
            for i, q in enumerate(qs):
                c = jj * len(qs) + i
                e_mu.append(abs(mu[jj, q] - mu_o[c]))
                e_ss.append(abs(part[jj, q] - ss_own[c]))
                if a.sequential:
                    e_var.append(abs(total_ss[jj, q] - ss_all[c]))
        mu_scale = float(np.abs(mu).max())
        checks = dict(sensors=sens, a_sens_rows_vs_oracle=max(errs_a), ak_rows_vs_oracle=max(errs_ak), aka_entries_rel=max(errs_aka),
                      posterior_columns=qs, posterior_mean_abs_over_max=max(e_mu) / mu_scale,
                      partial_sumsq_abs=max(e_ss), partial_sumsq_values=[float(v) for v in ss_own[:len(qs)]],
                      oracle_seconds=time.perf_counter() - t0,
                      note="A K rows / AkA entries: fp32-rounded covariance tables on the device vs the oracle's exact covariance; posterior: "
                           "V[:, q] = L^-1 (A3 K)[:, q] by scipy on oracle covariance columns (rounded to fp32 like the device's tables), the "
                           "device's L, u and forward operators -> mean and this rank's partial sum of squares at %d voxels x 3 blocks" % len(qs))
        if a.sequential:
            checks["posterior_var_abs"] = max(e_var)
        print("oracle checks:", checks, file=sys.stderr, flush=True)
    stages = {}
    for nm, fl, algf, valu, e0, e1 in ev:
        d = stages.setdefault(nm, dict(seconds=0.0, flop=0.0))
        d["seconds"] += e0.elapsed_time(e1) * 1e-3
        d["flop"] += fl
    ws_gb = {k: round(v.numel() * v.element_size() / 1e9, 2) for k, v in eng._ws.items()}
    ws_gb.update({"spectral:" + k: round(v.numel() * 8 / 1e9, 2) for k, v in eng._spectral._bufs.items()})
    out = dict(what="BASELINE config 5: rank %d of %d in the row form, %d^3 voxels x 3 properties, fp32 covariance tables, streamed operators"
                    % (a.rank, G, n),
               route=eng.route.describe(), N_voxels=N, M_rows=2 * Ms + sel.size, M_pad=M_pad, covariance_tables="fp32-rounded",
               wall_seconds=stamps, rank_step_seconds=rank_step, wall_seconds_cold_pass=cold, logl=float(logl),
               kernel_stage_seconds={k: round(v["seconds"], 3) for k, v in stages.items()},
               kernel_stage_tflops_executed={k: round(v["flop"] / v["seconds"] / 1e12, 2) for k, v in stages.items() if v["flop"] > 0 and v["seconds"] > 0},
               memory_map_GB=dict(sorted(ws_gb.items(), key=lambda kv: -kv[1])[:24]), max_memory_allocated_GB=torch.cuda.max_memory_allocated() / 1e9,
               oracle_checks=checks, posterior_finite=bool(np.isfinite(mu).all() and np.isfinite(part).all()),
               collectives_per_step="one all-gather of the (rows_r x 3 Ms) AkA row blocks (%.2f GB received per rank) + one all-reduce of "
                                    "3 N doubles (%.0f MB)" % ((G - 1) * (Ms // G) * 3 * Msp * 8 / 1e9, 3 * N * 8 / 1e6))
    if a.sequential:
        var = 1.0 - total_ss
        vmu = lambda k: hip.to_dev(mu[k].reshape(-1, 1))
        rg = apply_rows(eng, A_g, vmu(0))[:, 0].cpu().numpy() - y_g
        rm = apply_rows(eng, A_m, vmu(1))[:, 0].cpu().numpy() - y_m
        out["sequential"] = dict(
            total_seconds=sum(stamps.values()), voxel_properties_per_s=3.0 * N / sum(stamps.values()),
            checks=dict(var_min=float(var.min()), var_max=float(var.max()), finite=bool(np.isfinite(mu).all() and np.isfinite(var).all()),
                        rms_residual_grav=float(np.sqrt(np.mean(rg ** 2))), rms_residual_magn=float(np.sqrt(np.mean(rm ** 2))),
                        drill_rms_residual=float(np.sqrt(np.mean((mu[2][sel] - y_d) ** 2))) if sel.size else 0.0),
            checksums=dict(sum_abs_mu=[float(np.abs(mu[j]).sum()) for j in range(3)], sum_var=[float(var[j].sum()) for j in range(3)]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
