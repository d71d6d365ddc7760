#!/usr/bin/env python3
"""Benchmark of the GeoBO GP joint-inversion hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 64] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the hot path over one synthetic survey: everything about the forward operators A_g/A_m that the
route needs built on the device (stencil tables + boundary slabs on a lattice survey -- no operator is materialised there --, incl.
the host analysis of the survey geometry and its uploads), A K by the spectral route (radix-2 real-DFT transforms over x and z,
Toeplitz blocks over y), AkA (lattice Gram), Cholesky + L^-1, posterior mean + variance in the transposed order (round 3:
V = (L^-1 A3) K -- rows of L^-1 A through the same covariance kernels, squared and summed on the way out of the inverse transform;
the mean as three rows -- A_g^T w_g, A_m^T w_m, the drill weights, w = L^-T u -- through the covariance product), D2H of the cubes.
Which algorithm family a shape / rank count / precision runs in is decided by geobo_amd/plan.py and reported as `config.route`.
Workload = BASELINE
config 3/4: 64^3 voxels of 100 m, gravity + magnetics joint inversion (density and magnetic-susceptibility cubes, P_out = 2),
Matern-3/2 kernel with lengths (2.00, 2.02, 2.04) x 100 m, 50 drill-core constraints, M = 4096 + 4096 + 50 observation rows.
With N > 1 the SAME problem is sharded over the ranks (strong scaling) by ROWS: a rank owns Ms / N sensor rows
of each operator (its rows of A K over all voxels a chunk at a time, its row blocks of AkA, its rows of L^-1 A); collectives: one
all-gather of the AkA row blocks and one all-reduce of the partial sums of squares (P N doubles; every rank forms the mean whole).
HIP-event time of each collective per step is reported as `config.collective_ms_per_step` for every rank, next to every rank's stage
table; `--check` re-runs the problem on rank 0 alone inside the same job and compares the cubes.  (Surveys off the lattice and the
dense method keep the round-2 forms: voxel-column shards, all-reduce of the partial AkA or all-to-all of A K block-columns.)

Prints ONE JSON line on rank 0 (see the driver contract); extra objects:
  roofline      dominant kernel = geobo_spectral_y on the default route (the y stage of every covariance product, as an in-kernel spectral
                product on the matrix pipe since round 6; HBM-bound, fp64 pipe co-bound): ALGORITHMIC bytes per launch (spectrum read once +
                one output slab per property block) / mean launch duration measured with HIP events on the launch stream, against 8 TB/s.  (GEOBO_POSTERIOR=dense:
                geobo_posterior_reduce, --method dense: geobo_ak_fused_grid -- fp64 MFMA, algorithmic flop against 78.6 TFLOP/s.)
  roofline_assembly  the HBM-bound regime of SURVEY 8(d): one materialised covariance block (geobo_k_block), bytes written per
                launch / HIP-event duration against 8 TB/s (outside the timed steps)
  roofline_factorisation  the blocked Cholesky + triangular inverse on fp64 MFMA tiles (geobo_potrf_inv, one persistent launch per step):
                executed flop / HIP-event duration inside the timed steps against 78.6 TFLOP/s
  cpu_baseline  the NumPy/OpenBLAS oracle (kind "port") timed on this box's host cores on a bounded column sample of
                the same workload, plus the whole matrix-free oracle step at 16^3 and 32^3 (64^3 extrapolated from 32^3, labelled)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet, FP64 matrix (= FP64 vector); see DESIGN.md for the on-box microbenchmark


def synthetic_inputs(inv, md):
    """Synthetic survey of SURVEY.md section 8(d): reference 'cylinders' truth (+ smooth trend), survey = A rho / A chi
    rounded through float32, drill voxels from default_rng(2020).  Uses the device-resident operators."""
    from geobo_amd import hip
    s = inv.settings
    inv.create_cubegeometry()
    x3, y3, z3 = inv.xxx, inv.yyy, inv.zzz
    rad = s.yLcube / 18.
    rc1 = ((y3 - s.yLcube / 1.3 - rad) ** 2) + ((z3 + s.zLcube / 4 - rad) ** 2)
    rc2 = ((y3 - s.yLcube / 4. - rad) ** 2) + ((z3 + s.zLcube / 4 - rad) ** 2)
    rho = x3 * 0. + 0.1
    rho[rc2 <= rad ** 2] = 1.
    rho[rc1 <= rad ** 2] = 1.
    rho[(x3 < s.xLcube / 5.) | (x3 > s.xLcube * 4. / 5.)] = 0.1
    rho = rho + 0.02 * (x3 / s.xLcube + 2. * y3 / s.yLcube - z3 / s.zLcube)
    chi = s.gp_coeff[1] * rho
    xs = np.linspace(0.5, s.xNcube - 0.5, s.xNcube) * s.xvoxsize
    ys = np.linspace(0.5, s.yNcube - 0.5, s.yNcube) * s.yvoxsize
    X, Y, Z = np.meshgrid(xs, ys, s.zmax + s.zoff)
    loc = np.asarray([X.flatten(), Y.flatten(), Z.flatten()]).T
    inv.sensor_locations = loc
    eng = inv.engine
    # full operators on every rank for the synthetic data (the timed steps build only what each rank needs); in the
    # streamed-operator mode they are generated in row batches here as well (never resident)
    if eng.streamed or eng.auto_ops or eng.world > 1:
        # rows generated in batches (multi-rank runs: no rank holds a whole operator just to synthesise the survey)
        was, eng.streamed = eng.streamed, True
        A_g = eng.operator("grav", loc, B=s.magneticField * 0.)
        A_m = eng.operator("magn", loc, B=s.magneticField)
        grav = eng.apply_operator(A_g, rho).cpu().numpy().astype(np.float32).astype(np.float64)
        mag = eng.apply_operator(A_m, chi).cpu().numpy().astype(np.float32).astype(np.float64)
        eng.streamed = was
        eng.clear_operators()
    else:
        A_g = eng.operator("grav", loc, B=s.magneticField * 0., full=True)
        A_m = eng.operator("magn", loc, B=s.magneticField, full=True)
        grav = eng.apply_operator(A_g, rho).cpu().numpy().astype(np.float32).astype(np.float64)
        mag = eng.apply_operator(A_m, chi).cpu().numpy().astype(np.float32).astype(np.float64)
    drill0 = np.zeros_like(rho)
    if md > 0:
        sel = np.random.default_rng(2020).choice(rho.size, md, replace=False)
        drill0.reshape(-1)[sel] = rho.reshape(-1)[sel]
    return grav, mag, loc, drill0


PMC_FILES = {"ak_fused_grid": "r01_pmc_ak_fused_grid_v2.json", "posterior_reduce": "r03_pmc_posterior_reduce.json",
             "k_block_grid": "r06_pmc_k_block_grid_f64.json", "toeplitz_y": "r05_pmc_toeplitz_y.json", "toeplitz_y2t": "r05_pmc_toeplitz_y2t.json", "toeplitz_y2s": "r05_pmc_toeplitz_y2s.json",
             "spectral_y": "r06_pmc_spectral_y.json", "spectral_y2s": "r06_pmc_spectral_y2s.json"}
PMC_VALU_FILES = {"toeplitz_y": "r05_pmc_toeplitz_y_valu.json", "toeplitz_y2t": "r05_pmc_toeplitz_y2t_valu.json",
                  "toeplitz_y2s": "r05_pmc_toeplitz_y2s_valu.json", "spectral_y": "r06_pmc_spectral_y_valu.json",
                  "spectral_y2s": "r06_pmc_spectral_y2s_valu.json"}
GPU_DENSE_ROUTE = "profiles/r06_bench64_dense.json"   # builder-run bench line of `--method dense` (same algorithm as the CPU sample)


def factorisation_roofline(stage, steps):
    """north_star's second evidence item: the blocked Cholesky + L^-1 of AkA (inversion.py:100,105) on fp64 MFMA tiles, ONE persistent
    tile-DAG launch per step (csrc/potrf.hip), timed with HIP events inside the timed steps; flop = M_pad^3 / 3 (factor) + M_pad^3 / 3
    (inverse of the triangular factor), the arithmetic the kernel executes on its padded tiles; MFMA issue from the committed counter pass."""
    sec, fl = stage["seconds"] / steps, stage["flop"] / steps
    r = {"bound": "mfma", "kernel": "geobo_potrf_inv (potrf_dag_kernel: one launch = L, L^-1, log-determinant)", "achieved": fl / sec / 1e12,
         "peak": 78.6, "unit": "TFLOP/s", "frac": fl / sec / 78.6e12, "flop_per_launch": fl, "mean_launch_s": sec, "traffic": None,
         "bound_note": "dependency-chain bound: for 40 of the 66 tile columns at M_pad = 8448 the trailing update is shorter than the chain "
                       "diagonal tile -> panel -> next diagonal tile (profiles/r06_potrf_dag_trace_8448.txt)"}
    try:
        pf = [f for f in ("r06_pmc_potrf_dag.json", "r05_pmc_potrf_dag.json") if os.path.exists(os.path.join(ROOT, "profiles", f))][0]
        d = json.load(open(os.path.join(ROOT, "profiles", pf)))
        r["issue_counters"] = {"mfma_busy_frac_of_simd_cycles": d["derived"]["mfma_busy_frac_of_simd_cycles"],
                               "source": "profiles/%s (rocprofv3 --pmc passes of ONE launch at M_pad = 8448; not collected in this run)" % pf}
        r["traffic"] = d["derived"]["hbm_bytes_per_launch_corrected"]
    except Exception:
        pass
    return r


def pmc_traffic(kernel, executed_flop_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate --pmc runs of
    tools/run_fused_once.py / tools/run_posterior_once.py: FETCH_SIZE x2 (gfx950 half-count of wide loads,
    MI355X_MICROARCH.md) + WRITE_SIZE, KiB), scaled from the profiled launch to this launch by the executed flop count.
    Returns (bytes or None, source): the figure is a committed measurement of the same kernel and shape, not of this run."""
    for name in (PMC_FILES.get(kernel, ""), PMC_FILES.get(kernel, "").replace("r03_", "r02_"), PMC_FILES.get(kernel, "").replace("r03_", "r01_")):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return d["derived"]["hbm_bytes_per_launch_corrected"] * executed_flop_per_launch / d["flop"], "profiles/" + name
        except Exception:
            continue
    return None, None


def assembly_roofline(inv, lengths, rows=8192, launches=5):
    """SURVEY 8(d) regime (i), materialised kernel assembly: one block of create_cov (kernels.py:183-195) written to HBM -- here
    the (density, magsus) cross block of the headline workload for `rows` row voxels x all N column voxels, on the regular grid
    gathered from the block's difference-lattice table (geobo_k_block_grid).  Algorithmic bytes per launch = the block written
    (8 B per element) + the table and the row indices read once; HIP events on the launch stream; outside the timed steps."""
    from geobo_amd import hip
    from geobo_amd.engine import weight_matrix
    eng, s = inv.engine, inv.settings
    if not eng.use_grid:
        return None
    N = eng.N
    rows = int(min(rows, N))
    W = weight_matrix(s.gp_coeff)
    with torch.cuda.device(eng.device):
        tab = eng._cov_table(hip.kernel_id(s.kernelfunc, True), lengths[1], lengths[0], W[0][1], 1.0)
        ridx = torch.arange(0, N, max(N // rows, 1), device=eng.device, dtype=torch.int64)[:rows].contiguous()
        out = torch.empty((rows, N + 16), dtype=torch.float64, device=eng.device)[:, :N]
        hip.k_block_grid(tab, eng.nx, eng.ny, eng.nz, ridx, 0, out)
        torch.cuda.synchronize()
        evs = []
        for _ in range(launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hip.k_block_grid(tab, eng.nx, eng.ny, eng.nz, ridx, 0, out)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        durs = sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in evs)
        del out
    mean_s = sum(durs) / len(durs)
    by = 8.0 * rows * N + 8.0 * tab.numel() + 8.0 * rows
    traffic, tsrc = None, None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PMC_FILES["k_block_grid"])))
        traffic = d["derived"]["hbm_bytes_per_launch_corrected"] * by / d["derived"]["algorithmic_bytes"]
        tsrc = "profiles/" + PMC_FILES["k_block_grid"]
    except Exception:
        pass
    return {"bound": "hbm", "kernel": "geobo_k_block_grid (k_block_grid_kernel<double>)", "achieved": by / mean_s / 1e9, "peak": 8000.0,
            "unit": "GB/s", "frac": by / mean_s / 8e12, "traffic": traffic,
            "traffic_source": None if tsrc is None else tsrc + " (committed rocprofv3 --pmc passes of the same kernel: FETCH_SIZE x2 + WRITE_SIZE, "
            "scaled by the algorithmic bytes; not collected in this run)",
            "launches_timed": launches, "bytes_per_launch": by, "bytes_per_launch_is": "algorithmic: 8 B per element of the %d x %d block written "
            "+ the 2N-entry lattice table and the row indices read once" % (rows, N), "mean_launch_s": mean_s, "median_launch_s": durs[len(durs) // 2],
            "what": "materialised assembly of one covariance block (block (0,1) of create_cov, %s cross kernel) on the %d x %d x %d grid: SURVEY 8(d) "
                    "regime (i); not part of the timed step (the matrix-free path never materialises K)" % (s.kernelfunc, eng.nx, eng.ny, eng.nz)}


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def cpu_baseline(inv, lengths, target_seconds=20.0):
    """Oracle ("port") on the host cores, bounded sample of THIS workload: the fused A.K product (dense algorithm: the
    covariance block evaluated from squared distances, N-deep contraction) + the V solve / reductions for `b` voxel columns
    x 2 properties, operators and Cholesky factor taken as given (so the CPU rate is an upper bound: A_sens, AkA and the
    factorisation are not charged)."""
    from oracle import geobo_oracle as O
    from scipy.linalg import solve_triangular
    eng = inv.engine
    s = inv.settings
    N, Ms = eng.N, eng.Ms
    # resident copies of the operators (in the streamed-operator mode the timed steps never held them)
    A_g = eng.operator("grav", inv.sensor_locations, B=s.magneticField * 0., full=True)[:Ms, :N].cpu().numpy()
    A_m = eng.operator("magn", inv.sensor_locations, B=s.magneticField, full=True)[:Ms, :N].cpu().numpy()
    L = torch.tril(eng.last["L"]).cpu().numpy()
    rows = np.r_[0:Ms, eng.Ms_pad:eng.Ms_pad + Ms, 2 * eng.Ms_pad:2 * eng.Ms_pad + inv._sel.size]
    L = L[np.ix_(rows, rows)]
    u = eng.last["u"].cpu().numpy()[rows]
    P3 = O.grid_points((s.xNcube, s.yNcube, s.zNcube), (s.xvoxsize, s.yvoxsize, s.zvoxsize))
    W = O.weight_matrix(s.gp_coeff)
    sel = inv._sel
    name = s.kernelfunc
    M = L.shape[0]

    def columns(b):
        """b voxel columns SPREAD over the cube (round-5 review: the sample was one contiguous block at the cube's centre): evenly strided
        flat indices from the first to the last voxel -- every y-slab incl. the two padded ones iy = 0 and ny - 1, x / z faces, the corners
        p = 0 and N - 1 -- plus up to eight drilled voxels."""
        cols = np.unique(np.r_[np.linspace(0, N - 1, b).astype(np.int64), sel[:8]])
        return cols

    def sample(cols):
        t0 = time.perf_counter()
        D2 = O.sqdist(P3, P3[cols])
        out = []
        for j in (0, 1):
            AK = np.empty((M, cols.size))
            AK[:Ms] = A_g @ O.k_block(name, D2, lengths, W, 0, j)
            AK[Ms:2 * Ms] = A_m @ O.k_block(name, D2, lengths, W, 1, j)
            if sel.size:
                AK[2 * Ms:] = O.k_block(name, D2[sel], lengths, W, 2, j)
            V = solve_triangular(L, AK, lower=True)
            out.append((V.T @ u, 1.0 - np.einsum("mq,mq->q", V, V)))
        return time.perf_counter() - t0, out

    t_small, _ = sample(columns(16))
    b = int(max(16, min(2048, 16 * target_seconds / max(t_small, 1e-3))))
    cols = columns(b)
    b = cols.size
    t, out = sample(cols)
    flop = 2.0 * b * (2.0 * (2 * Ms) * N + 1.0 * M * M + 4.0 * M)       # section 8(d) terms for b columns x 2 properties
    return dict(value=2.0 * b / t, unit="voxel-properties/s", cores=host_threads(), kind="port",
                algorithm="dense (covariance evaluated from squared distances, N-deep contraction): SURVEY 8(d) flop model",
                gflops=flop / t / 1e9,
                sample="%d of %d voxel columns x 2 properties of the same workload, spread over the cube (strided from the first to the last "
                       "voxel: every y-slab incl. the padded ones, faces, corners; %d drilled voxels): fused A.K + triangular solve + "
                       "mean/variance reductions in NumPy/OpenBLAS (oracle/geobo_oracle.py); operators and Cholesky factor "
                       "given, so this is an upper bound on the CPU rate; %.1f s" % (b, N, min(8, sel.size), t)), (cols, out)


def cpu_baseline_forms(sizes, dense_sizes, F64, gflops_sample):
    """SURVEY.md section 8(d) CPU forms on this box's host cores: (b) the oracle's full matrix-free inversion (operators, A K,
    AkA, Cholesky, solves, reductions -- the whole step, same synthetic survey formulas) at the given cube edges, (a) the
    reference-shaped form (full K and full posterior covariance, inversion.py:77-122) at `dense_sizes`; the 64^3 time is
    EXTRAPOLATED from the largest measured matrix-free size by the 8(d) flop model and labelled so."""
    from oracle import geobo_oracle as O
    forms = {}

    def flop_model(n, md):
        N, Ms = n ** 3, 2 * n * n
        M = Ms + md
        return 2.0 * Ms * N * N * 2 + 2.0 * M * Ms * N + M ** 3 / 3.0 + 1.0 * M * M * 2 * N + 4.0 * M * 2 * N + M * M

    last = None
    for n, dense in [(n, False) for n in sizes] + [(n, True) for n in dense_sizes]:
        G = O.Grid(nx=n, ny=n, nz=n, xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, kernelfunc="matern32")
        t0 = time.perf_counter()
        sv = O.synthetic_survey(G, 50 if n >= 16 else 5)
        d0 = sv["drilldata0"]
        O.cubing(G, sv["gravfield"], sv["magfield"], d0[d0 != 0], sv["sensor_locations"], d0,
                 gp_length=np.array([2.00, 2.02, 2.04]) * 100.0, dense=dense, props=(0, 1, 2) if dense else (0, 1), A=sv["A"])
        t = time.perf_counter() - t0
        key = ("reference_shaped_%d" if dense else "matrix_free_%d") % n
        forms[key] = dict(seconds=t, voxel_properties_per_s=2.0 * n ** 3 / t, measured=True,
                          what=("full K + full posterior covariance like inversion.py:77-122" if dense else
                                "whole step incl. A_sens, A K, AkA, Cholesky, solves, reductions") + ", oracle/geobo_oracle.py")
        if not dense:
            last = (n, t)
    if last is not None:
        n, t = last
        t64 = t * F64 / flop_model(n, 50)
        forms["matrix_free_64_extrapolated"] = dict(seconds=t64, voxel_properties_per_s=2.0 * 64 ** 3 / t64, measured=False,
                                                    what="EXTRAPOLATED from matrix_free_%d by the section 8(d) flop model "
                                                         "(%.3e / %.3e flop)" % (n, F64, flop_model(n, 50)))
    if gflops_sample:
        t64 = F64 / (gflops_sample * 1e9)
        forms["dense_algorithm_64_from_sample_rate"] = dict(seconds=t64, voxel_properties_per_s=2.0 * 64 ** 3 / t64, measured=False,
                                                            what="section 8(d) flop count of the 64^3 step / the GFLOP/s the bounded sample achieved")
    return forms


def relaunch_with_ranks(a):
    """`python bench.py --gpus N` without a launcher in the environment: start the N ranks ourselves (one process per GPU,
    torch.distributed.run, rendezvous on 127.0.0.1) and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # the task environment states that this image's driver supports dmabuf IPC only and exports HSA_ENABLE_IPC_MODE_LEGACY=0 for RCCL;
    # kept for ranks started from a shell that lost it (no multi-GPU node was available to the builder to verify it is needed)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=64, help="cube edge in voxels (64 = BASELINE headline)")
    ap.add_argument("--kernel", default="matern32")
    ap.add_argument("--drill", type=int, default=50)
    ap.add_argument("--props", type=int, default=2, choices=[2, 3], help="property blocks computed (2 = density + magsus, the headline; "
                    "3 adds the drill property: BASELINE config 5)")
    ap.add_argument("--method", default="auto", choices=["auto", "dense", "spectral"],
                    help="A.K route: dense = fused in-kernel covariance generation; spectral = real-DFT on batched MFMA GEMMs")
    ap.add_argument("--assembly", default="f64", choices=["f64", "f32"], help="f32 = BASELINE config 5's fp32 kernel assembly (A K and the "
                    "covariance tables in fp32, fp64 accumulation and factorisation); NOT the headline configuration")
    ap.add_argument("--operators", default="auto", choices=["auto", "resident", "streamed"],
                    help="auto (the library default): no materialised operators where the lattice forms make them unnecessary")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-forms", default="16,32", help="comma list of cube edges for the full matrix-free CPU oracle step "
                    "(SURVEY 8(d) form (b)); the 64^3 figure is extrapolated from the LARGEST one (32^3: ~1 min on the box's 128 threads)")
    ap.add_argument("--cpu-dense-forms", default="16", help="cube edges for the reference-shaped CPU form (a) of SURVEY 8(d): full K and full "
                    "posterior covariance like inversion.py:77-122; '16' (~30 s, the default) or '16,20' ('' = none)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--oversubscribe", action="store_true", help="dry runs: allow several ranks per device (gloo backend)")
    ap.add_argument("--check", action="store_true", help="N > 1: after the timed steps rank 0 runs the same problem alone (a 1-rank engine on its "
                    "own device, inside this job) and the line carries the largest normwise difference of the cubes and both checksum sets")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_with_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s); refusing to print a line for a different job size" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    if ndev < world and not (a.oversubscribe and a.backend != "nccl"):
        raise SystemExit("bench.py: --gpus %d needs %d visible devices, found %d" % (a.gpus, world, ndev))
    local = local % max(ndev, 1)   # (--oversubscribe dry runs put several ranks on one device)
    torch.cuda.set_device(local)
    dist = None
    ranks_reported = 1
    # one rank: a process group exists only when a launcher started us (WORLD_SIZE=1 under torch.distributed.run) or with --check, which
    # then pushes the row form's collectives through the backend with one rank (RCCL for "nccl": the one-GPU stand-in for a node)
    if world > 1 or a.check or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        dist.init_process_group(backend=a.backend, rank=rank, world_size=world)
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)                      # the rank count the backend itself reports (RCCL for "nccl")
        ranks_reported = int(round(float(ones.item())))
        if ranks_reported != a.gpus or dist.get_world_size() != a.gpus:
            raise SystemExit("bench.py: backend reports %d ranks, --gpus %d" % (ranks_reported, a.gpus))

    from geobo_amd.config_loader import Settings
    from geobo_amd.inversion import Inversion
    n = a.size
    s = Settings(dict(xmin=0, xmax=100.0 * n, ymin=0, ymax=100.0 * n, zmax=0, zoff=1, zLcube=100.0 * n, xNcube=n, yNcube=n,
                      zNcube=n, gp_lengthscale=2, gp_err=[0.1, 0.1, 0.1], gp_coeff=[1.0, 0.2, 0.2], kernelfunc=a.kernel,
                      XMAG=0, YMAG=0, ZMAG=1))
    inv = Inversion(settings=s, props=(0, 1, 2)[:a.props], rank=rank, world=world, device="cuda:%d" % local, method=a.method,
                    assembly=a.assembly, operators=a.operators)
    grav, mag, loc, drill0 = synthetic_inputs(inv, a.drill)
    gp_length = np.array([2.00, 2.02, 2.04]) * s.xvoxsize if a.kernel == "matern32" else None
    # The headline configuration has committed golden VALUES (tests/golden/oracle64_sample_matern32.npz: posterior mean / variance at
    # ~3000 voxels spread over the cube -- padded slabs, faces, corners, every drilled voxel -- from the oracle's own operators, FFT rows
    # of A K, scipy Cholesky; nothing of the device in it).  Its survey is the same synthetic model through the ORACLE's operators
    # (float32-rounded like the GeoTIFF path; the device's own A rho differs from it in the last float32 bit of a few values), so the
    # timed steps run on THAT survey and the line carries the comparison (`parity_vs_oracle_sample`).
    golden = None
    gpath = os.path.join(ROOT, "tests", "golden", "oracle64_sample_matern32.npz")
    if n == 64 and a.kernel == "matern32" and a.drill == 50 and a.assembly == "f64" and os.path.exists(gpath):
        gz = np.load(gpath)
        same = (np.array_equal(gz["sensor_locations"], loc) and np.array_equal(gz["sel"], np.flatnonzero(drill0.reshape(-1) != 0))
                and np.array_equal(gz["drillvalues"], drill0.reshape(-1)[gz["sel"]]) and np.array_equal(gz["gp_length_in"], gp_length))
        if same and np.abs(gz["gravfield"] - grav).max() <= 1e-6 * np.abs(grav).max():
            golden = gz
            grav, mag = gz["gravfield"].copy(), gz["magfield"].copy()

    def step():
        inv.engine.clear_operators()          # operators, stencil tables, survey-geometry plan: rebuilt inside every step
        inv._axes_of = (None, None)           # ... and the node-axis check of the Edges tensor
        if gp_length is not None:
            inv.gp_length = gp_length.copy()
        else:
            inv.gp_length = s.gp_lengthscale * np.asarray([s.xvoxsize] * 3)
        return inv.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()            # the objects that exist now stay out of the collector's generations: a full collection inside a timed step cost
                           # 36 ms once per ~20 steps at 32^3 (three times the step), host time with an idle device
    inv.engine.kernel_events = []
    fence()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(a.steps):
        cubes = step()                         # ends with the D2H of the cubes (synchronous): the marks need no extra sync
        marks.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ev = inv.engine.kernel_events
    inv.engine.kernel_events = None
    # per-stage / per-kernel device time from HIP events recorded on the launch stream (torch's current stream)
    stages = {}
    for name, fl, alg, valu, e0, e1 in ev:
        d = stages.setdefault(name, dict(calls=0, seconds=0.0, flop=0.0, alg=0.0, bytes=0.0, valu=0.0, durs=[]))
        d["calls"] += 1
        sec = e0.elapsed_time(e1) * 1e-3
        d["seconds"] += sec
        d["durs"].append(sec)
        d["flop"] += fl
        # per-kernel events ("kernel:...") carry algorithmic BYTES in the event's third slot, stage events algorithmic FLOP: kept apart
        # so that no sum over stages can mix the two units
        d["bytes" if name.startswith("kernel:") else "alg"] += alg
        d["valu"] += valu
    single = [k for k in stages if k in ("ak_fused_grid", "ak_fused", "posterior_reduce", "posterior_zgemm", "aka_gemm_nt", "kernel:toeplitz_y", "kernel:toeplitz_y2t", "kernel:toeplitz_y2s")]
    dom = max(single, key=lambda k: stages[k]["seconds"]) if single else None
    kernel_names = {"ak_fused_grid": "geobo_ak_fused_grid (gemm_f64_kernel<4,2,TAB>)", "ak_fused": "geobo_ak_fused (gemm_f64_kernel<4,2,GEN>)",
                    "posterior_reduce": "geobo_posterior_reduce (gemm_f64_kernel<4,2,NN,REDUCE>)",
                    "posterior_zgemm": "geobo_gemm_nn, triangular X: Z = L^-1[:, operator columns] A (gemm_f64_kernel<4,2,NN>; two launches per step)",
                    "kernel:toeplitz_y": "geobo_toeplitz_y (toeplitz_y_kernel<64, 1>: the launches with two property blocks per read of the "
                                         "spectrum): y stage of the covariance products (A K and V = (L^-1 A) K)",
                    "kernel:toeplitz_y2t": "geobo_toeplitz_y2t (toeplitz_y2_kernel<64>: the two-term rows of the transposed posterior, V = Z_g K_0j + "
                                           "Z_m K_1j for two property blocks in one pass over both input spectra)",
                    "kernel:toeplitz_y2s": "geobo_toeplitz_y2s (toeplitz_y2s_kernel<64>: the two-term rows of the transposed posterior with the shared "
                                           "cross block K_01 = K_10 -- three products per mode instead of four, both property blocks in one pass)",
                    "aka_gemm_nt": "geobo_gemm_nt (gemm_f64_kernel<4,2,NT>)"}

    coll_names = {"xgmi_all_gather": "all_gather", "xgmi_all_reduce": "all_reduce", "xgmi_all_to_all": "all_to_all"}
    my_table = {k: round(1e3 * v["seconds"] / a.steps, 3) for k, v in stages.items()}
    my_coll = {coll_names[k]: round(1e3 * v["seconds"] / a.steps, 3) for k, v in stages.items() if k in coll_names}
    tables, colls = [my_table], [my_coll]
    if dist is not None:
        tables, colls = [None] * world, [None] * world
        dist.all_gather_object(tables, my_table)
        dist.all_gather_object(colls, my_coll)
    check = None
    operators_in_use = sorted({"streamed" if type(v).__name__ == "StreamedOperator" else "resident" for v in inv.engine._A.values()})
    if a.check and world == 1:
        # the multi-rank form of this step on the one rank there is -- row form forced, its all-gather / all-reduce / agreement issued
        # through the backend (sharding._live) -- against the cubes of the timed steps (a second engine beside the timed one: 2 x 50 GB)
        rows_before = os.environ.get("GEOBO_ROWS")
        os.environ["GEOBO_ROWS"] = "1"
        try:
            solo = Inversion(settings=s, props=(0, 1, 2)[:a.props], rank=0, world=1, device="cuda:%d" % local, method=a.method,
                             assembly=a.assembly, operators=a.operators)
            solo.engine.force_collectives = True
            solo.engine.kernel_events = []
            solo.gp_length = (gp_length.copy() if gp_length is not None else s.gp_lengthscale * np.asarray([s.xvoxsize] * 3))
            ref = solo.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
        finally:
            if rows_before is None:
                os.environ.pop("GEOBO_ROWS", None)
            else:
                os.environ["GEOBO_ROWS"] = rows_before
        idx = (0, 1, 3, 4) if a.props == 2 else range(6)
        evs = [(e[0], e[4].elapsed_time(e[5])) for e in solo.engine.kernel_events if e[0] in coll_names]
        check = {"what": "row form forced on the one rank, collectives through backend '%s' with world size 1" % a.backend,
                 "max_normwise_diff_vs_timed_steps": max(float(np.abs(cubes[i] - ref[i]).max() / np.abs(ref[i]).max()) for i in idx),
                 "cube_checksums_row_form": [float(np.abs(ref[i]).sum()) for i in idx], "route_row_form": solo.engine.route.describe(),
                 "family_of_the_check_step": solo.engine.step_route, "collective_ms": {coll_names[k]: round(v, 3) for k, v in evs},
                 "logl_diff": float(abs(inv.logl - solo.logl))}
        del solo
    if a.check and world > 1:
        inv.engine.release()                   # (dry runs put all ranks on one device: make room for the 1-rank engine)
        dist.barrier()
        if rank == 0:
            solo = Inversion(settings=s, props=(0, 1, 2)[:a.props], rank=0, world=1, device="cuda:%d" % local, method=a.method,
                             assembly=a.assembly, operators=a.operators)
            solo.gp_length = (gp_length.copy() if gp_length is not None else s.gp_lengthscale * np.asarray([s.xvoxsize] * 3))
            ref = solo.cubing(grav, mag, drill0[drill0 != 0], loc, drill0)
            idx = (0, 1, 3, 4) if a.props == 2 else range(6)
            check = {"max_normwise_diff_vs_1_rank": max(float(np.abs(cubes[i] - ref[i]).max() / np.abs(ref[i]).max()) for i in idx),
                     "cube_checksums_1_rank": [float(np.abs(ref[i]).sum()) for i in idx], "route_1_rank": solo.engine.route.describe(),
                     "logl_diff": float(abs(inv.logl - solo.logl))}
            del solo
        dist.barrier()

    if rank == 0:
        eng = inv.engine
        N = eng.N
        p_out = a.props
        value = p_out * N * a.steps / dt
        M = 2 * eng.Ms + int((drill0 != 0).sum())
        Msd = 2 * eng.Ms
        F = 2.0 * Msd * N * N * p_out + 2.0 * M * Msd * N + M ** 3 / 3.0 + 1.0 * M * M * p_out * N + 4.0 * M * p_out * N + M * M
        mfma = {k: v for k, v in stages.items() if v["flop"] > 0 and not k.startswith("kernel:")}   # (per-kernel brackets sit INSIDE stage brackets)
        F_mfma = sum(d["flop"] - d["valu"] for d in mfma.values()) / a.steps   # `flop` = everything executed, `valu` = its fp64-VALU part
        F_valu = sum(d["valu"] for d in mfma.values()) / a.steps
        step_ms = sorted(1e3 * (b - a_) for a_, b in zip(marks[:-1], marks[1:]))
        roof = None
        if dom in ("kernel:toeplitz_y", "kernel:toeplitz_y2t", "kernel:toeplitz_y2s"):
            # The y stage of every covariance product.  Round 6: geobo_spectral_y* (the y axis through its own spectrum inside the kernel, on
            # the matrix pipe) wherever instantiated; the direct vector-pipe kernels geobo_toeplitz_y* otherwise (GEOBO_Y_MFMA=0: the A/B).
            # Both roofs as peers: algorithmic bytes per launch against 8 TB/s, executed flop per launch against the fp64 pipe that
            # v_mfma_f64 and the vector FMAs share (78.6 TFLOP/s, profiles/r01_mfma_coissue.txt); `bound` names the larger fraction.
            y_mfma = bool(getattr(eng._spectral, "y_mfma", False))
            d = stages[dom]
            calls = d["calls"]
            mean_s = d["seconds"] / calls
            by = d["bytes"] / calls
            pflop = d["flop"] / calls if d["flop"] > 0 else (d["valu"] / calls if d["valu"] > 0 else 2.0 * eng.ny * (by / 8.0) * (2.0 / 3.0))
            vflop = d["valu"] / calls
            pmc_key = dom.split(":")[1]
            if y_mfma:
                pmc_key = {"toeplitz_y": "spectral_y", "toeplitz_y2s": "spectral_y2s"}.get(pmc_key, pmc_key)
            traffic, tsrc = None, None
            for pf in (PMC_FILES.get(pmc_key, ""), PMC_FILES.get(pmc_key, "").replace("r05_", "r04_")):
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", pf)))
                    traffic = pj["derived"]["hbm_bytes_per_launch_corrected"] * by / pj["derived"]["algorithmic_bytes"]
                    tsrc = "profiles/" + pf
                    break
                except Exception:
                    pass
            f_hbm, f_pipe = by / mean_s / 8e12, pflop / mean_s / 1e12 / FP64_MATRIX_PEAK_TFLOPS
            ytab = {}
            for k, v in stages.items():
                if k.startswith("kernel:toeplitz"):
                    c_, m_, b_ = v["calls"], v["seconds"] / v["calls"], v["bytes"] / v["calls"]
                    pf_ = v["flop"] / c_ if v["flop"] > 0 else (v["valu"] / c_ if v["valu"] > 0 else 2.0 * eng.ny * (b_ / 8.0) * (2.0 / 3.0 if k == "kernel:toeplitz_y" else 0.5))
                    ytab[k.split(":")[1]] = {"ms_per_step": round(1e3 * v["seconds"] / a.steps, 2), "launches_per_step": c_ / a.steps,
                                             "mean_launch_ms": round(1e3 * m_, 4), "frac_hbm_8TBps": round(b_ / m_ / 8e12, 3),
                                             "frac_fp64_pipe_78.6TF": round(pf_ / m_ / 1e12 / FP64_MATRIX_PEAK_TFLOPS, 3)}
            issue = None
            try:
                pv = json.load(open(os.path.join(ROOT, "profiles", PMC_VALU_FILES[pmc_key])))["derived"]
                issue = {"source": "profiles/" + PMC_VALU_FILES[pmc_key] + " (committed rocprofv3 --pmc passes of a lone launch)",
                         "valu_busy_of_simd_cycles": round(pv["sq_active_inst_valu_x4_over_simd_cycles"], 3),
                         "fma_f64_share_of_valu_instructions": round(pv["fma_f64_share_of_valu_instructions"], 3),
                         "clock_GHz_of_the_profiled_launch": round(pv["clock_GHz_during_profiled_pass"], 2)}
                if "mfma_busy_frac_of_simd_cycles" in pv:
                    issue["mfma_busy_of_simd_cycles"] = round(pv["mfma_busy_frac_of_simd_cycles"], 3)
            except Exception:
                pass
            r_hbm = {"roof": "hbm", "achieved": by / mean_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": f_hbm}
            r_pipe = {"roof": "fp64_pipe (v_mfma_f64 and the vector FMAs share it)", "achieved": pflop / mean_s / 1e12,
                      "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_pipe, "issue_counters": issue,
                      "vector_part_TFLOPs": vflop / mean_s / 1e12}
            bind, other = (r_pipe, r_hbm) if f_pipe >= f_hbm else (r_hbm, r_pipe)
            if y_mfma:
                kname = {"kernel:toeplitz_y": "geobo_spectral_y (spectral_y_kernel<64, 1, 2>: the launches with two property blocks per read of the "
                                              "spectrum): y stage of the covariance products (A K and V = (L^-1 A) K), the y axis through its own "
                                              "spectrum inside the kernel",
                         "kernel:toeplitz_y2s": "geobo_spectral_y2s (spectral_y_kernel<64, 2, 2>: the two-term rows of the transposed posterior with "
                                                "the shared cross block, the terms meeting in the y spectrum)"}.get(dom, kernel_names[dom])
                note = ("the y stage is an in-kernel spectral product (csrc/spectral_y.hip): skew-circulant embedding of the per-mode Toeplitz "
                        "block, radix 4 over quarter-period orbits -- ny^2 / 2 multiply-adds per transform and mode on v_mfma_f64_16x16x4 (the 16 "
                        "MFMA columns are 16 modes: 128-byte segments), orbit butterflies and eigenvalue scaling on the vector pipe; one analysis per "
                        "term + one synthesis per block.  It streams its spectra once (PMC: 1.003 x algorithmic) at the rate a bare 1-read-2-writes "
                        "stream of the same segments reaches on this device (5.16 TB/s, profiles/r05_hbm_copy_runs.txt): HBM binds, the fp64 pipe "
                        "is `co_bound`.  `flop_per_launch` = executed flop (MFMA + vector; additions count 1), `bytes_per_launch` = algorithmic bytes")
            else:
                kname = kernel_names[dom]
                note = ("the y stage runs on the fp64 VECTOR FMAs (one (x, z) mode per lane, ny^2 FMA per mode, block and term: no operand "
                        "is shared between modes, so no MFMA) and streams its spectra once: arithmetic intensity 10.7 flop/B against a ridge of "
                        "9.8 -- HBM and the fp64 pipe limit it together (`co_bound`).  `flop_per_launch` = algorithmic FMA flop, "
                        "`bytes_per_launch` = algorithmic bytes")
            roof = {"bound": "fp64_pipe" if bind is r_pipe else "hbm", "kernel": kname, "achieved": bind["achieved"],
                    "peak": bind["peak"], "unit": bind["unit"], "frac": bind["frac"], "co_bound": other,
                    "issue_counters": issue, "frac_hbm": f_hbm, "frac_fp64_pipe": f_pipe,
                    "bound_note": note,
                    "traffic": traffic,
                    "traffic_source": None if tsrc is None else tsrc + " (committed rocprofv3 --pmc passes of the same kernel: FETCH_SIZE x2 + "
                    "WRITE_SIZE, scaled by the algorithmic bytes; not collected in this run)",
                    "launches_timed": calls, "flop_per_launch": pflop, "bytes_per_launch": by,
                    "bytes_per_launch_is": "algorithmic: the batch's (x, z)-spectrum read once (8 B x rows x ny x 4 nx nz; both terms' spectra for the "
                                           "two-term kernel) + one output slab per property block",
                    "y_stage_kernels": ytab,
                    "mean_launch_s": mean_s, "median_launch_s": sorted(d["durs"])[calls // 2],
                    "share_of_step": d["seconds"] / dt}
        elif dom:
            d = stages[dom]
            calls = d["calls"]
            mean_s = d["seconds"] / calls
            alg, exe = d["alg"] / calls, d["flop"] / calls
            traffic, tsrc = pmc_traffic(dom, exe)
            ach = alg / mean_s / 1e12
            roof = {"bound": "mfma", "kernel": kernel_names.get(dom), "achieved": ach, "peak": FP64_MATRIX_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": None if tsrc is None else tsrc + " (committed rocprofv3 --pmc passes of the same kernel and "
                    "shape: FETCH_SIZE x2 + WRITE_SIZE; not collected in this run)",
                    "launches_timed": calls, "flop_per_launch": alg, "flop_per_launch_is": "algorithmic, SURVEY 8(d): unpadded M, "
                    "triangular L^-1 (M^2 P_c nc + 4 M P_c nc for the posterior; 2 Ms N ncols for the fused product)",
                    "executed_flop_per_launch": exe, "achieved_executed": exe / mean_s / 1e12,
                    "mean_launch_s": mean_s, "median_launch_s": sorted(d["durs"])[calls // 2]}
        out = {
            "metric": "voxels/sec posterior (mean+var) for 64^3 x 2-prop joint inversion; fp64 roofline %",
            "value": value, "unit": "voxel-properties/s", "n_gpus": ranks_reported, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if a.assembly == "f64" else "f32 assembly / f64 accumulate+factorise",
            "data": "synthetic (the reference's cylinders model, simcube.py:83-92, plus a smooth trend 0.02 (x/X + 2 y/Y - z/Z) so that drill "
                    "values are non-degenerate at every size; survey = A rho, A chi rounded through float32)",
            "config": {"workload": "%d^3 voxel cube (100 m), gravity+magnetics joint inversion, %s kernel lengths %s, "
                                   "%d drill constraints, M=%d rows, %s cubes (P_out=%d)" % (
                                       n, a.kernel, "(2.00,2.02,2.04)x100 m" if gp_length is not None else "2 x 100 m (create_cov makes them (2.00,2.04,2.00))",
                                       a.drill, M, "density+magsus" + ("+drill" if p_out == 3 else ""), p_out),
                       "N_voxels": N, "M_rows": M, "props_out": p_out, "parallelism": ("sensor-row shards x%d" if eng._rowpath else "voxel-column shards x%d") % world,
                       "route": eng.route.describe(), "route_family_of_the_steps": eng.step_route, "route_note": eng.route.note or None,
                       "collective_ms_per_step": colls, "stage_ms_per_step_by_rank": tables, "check_vs_1_rank": check,
                       "backend": a.backend if dist is not None else None, "ranks_reported_by_backend": ranks_reported,
                       "method": "spectral" if inv.engine.use_spectral else "dense", "assembly": a.assembly, "operators": a.operators,
                       "operators_in_use": operators_in_use,
                       "row_exchange": bool(inv.engine.exchange and not inv.engine._rowpath), "row_posterior": bool(inv.engine._rowpath),
                       "ms_per_step_median": step_ms[len(step_ms) // 2], "ms_per_step_all_rank0": [round(v, 2) for v in step_ms],
                       "ms_per_step_in_order_rank0": [round(1e3 * (b - a_), 2) for a_, b in zip(marks[:-1], marks[1:])],
                       "cube_checksums": [float(np.abs(cubes[i]).sum()) for i in ((0, 1, 3, 4) if p_out == 2 else range(6))],
                       "dense_algorithmic_flop_per_step": F,
                       "executed_mfma_flop_per_step_rank0": F_mfma, "executed_valu_flop_per_step_rank0": F_valu,
                       "mfma_time_frac_of_step_rank0": sum(d["seconds"] for d in mfma.values()) / dt,
                       "stage_ms_per_step_rank0": {k: round(1e3 * v["seconds"] / a.steps, 3) for k, v in stages.items()},
                       "kernel_tflops_executed_rank0": {k: round(v["flop"] / v["seconds"] / 1e12, 2) for k, v in stages.items() if v["flop"] > 0}},
            "roofline": roof,
        }
        if golden is not None:
            q, N_ = golden["voxels"], eng.N
            var_all = inv.cov_rec.diagonal()
            nrm = lambda got, want: float(np.abs(got - want).max() / np.abs(want).max())
            out["parity_vs_oracle_sample"] = {
                "fixture": "tests/golden/oracle64_sample_matern32.npz (tests/golden/make_oracle64_sample.py: oracle operators, FFT rows of A K, "
                           "scipy Cholesky -- nothing from the device)",
                "voxels": int(q.size), "includes": "every 131st voxel, the 50 drilled voxels, 256 voxels of each padded slab iy = 0 / 63, 128 of "
                                                   "each x / z face, the 8 corners",
                "mean_normwise": [nrm(inv.mu_rec[j * N_ + q], golden["mu"][j]) for j in (0, 1)],
                "variance_max_abs": [float(np.abs(var_all[j * N_ + q] - golden["var"][j]).max()) for j in (0, 1)],
                "logl_rel": float(abs(inv.logl - float(golden["logl"])) / abs(float(golden["logl"]))), "tolerance": "1e-8 (north_star)"}
        if world == 1 and a.assembly == "f64":
            out["roofline_assembly"] = assembly_roofline(inv, [float(v) for v in inv.gp_length])
        if "potrf_inv" in stages and stages["potrf_inv"]["flop"] > 0:
            out["roofline_factorisation"] = factorisation_roofline(stages["potrf_inv"], a.steps)
        if not a.no_cpu and world == 1:   # CPU baseline: rank 0 at N = 1 only
            lengths = inv.gp_length
            cb, (cols, smp) = cpu_baseline(inv, [float(v) for v in lengths])
            cb["sample_max_abs_diff_vs_gpu_mu"] = float(np.abs(smp[0][0] - inv.mu_rec[cols]).max())
            cb["sample_max_abs_diff_vs_gpu_var"] = float(np.abs(smp[0][1] - inv.cov_rec.diagonal()[cols]).max())
            sizes = [int(v) for v in a.cpu_forms.split(",") if v]
            dsizes = [int(v) for v in a.cpu_dense_forms.split(",") if v]
            F64 = 2.0 * 8192 * 262144.0 ** 2 * 2 + 2.0 * 8242 * 8192 * 262144 + 8242 ** 3 / 3.0 + 8242.0 ** 2 * 2 * 262144 + 4.0 * 8242 * 2 * 262144 + 8242 ** 2
            cb["forms"] = cpu_baseline_forms(sizes, dsizes, F64, cb["gflops"])
            out["cpu_baseline"] = cb
            # GPU / CPU ratios, like for like first (round-4 review).  (i) the SAME matrix-free algorithm on both sides at the largest size
            # the CPU form was measured at; (ii) the same DENSE algorithm on both sides at 64^3 (committed `--method dense` line against
            # this run's dense-algorithm sample); (iii) last, and NOT like for like: this run's structured GPU step against that dense
            # CPU sample -- the ratio the r01-r04 lines called speedup_vs_cpu_baseline.  None of them says anything about kernel quality.
            ratios = {}
            mf = sorted((int(k.split("_")[-1]), v) for k, v in cb["forms"].items() if k.startswith("matrix_free_") and v.get("measured"))
            if mf:
                nmf, fmf = mf[-1]
                try:
                    g = json.load(open(os.path.join(ROOT, "profiles", "r06_bench%d_config2.json" % nmf)))
                    ratios["same_matrix_free_algorithm_at_%d_cubed" % nmf] = {
                        "gpu_voxel_properties_per_s": g["value"], "gpu_source": "profiles/r06_bench%d_config2.json (committed line, exp kernel, no drill rows)" % nmf,
                        "cpu_voxel_properties_per_s_measured_here": fmf["voxel_properties_per_s"], "ratio": g["value"] / fmf["voxel_properties_per_s"]}
                except Exception:
                    pass
            try:   # the same (dense) algorithm on the GPU: committed bench line of `--method dense`
                dense = json.load(open(os.path.join(ROOT, GPU_DENSE_ROUTE)))
                ratios["same_dense_algorithm_at_64_cubed"] = {
                    "gpu_dense_route_voxel_properties_per_s": dense["value"], "gpu_source": GPU_DENSE_ROUTE,
                    "cpu_sample_voxel_properties_per_s": cb["value"], "ratio": dense["value"] / cb["value"]}
            except Exception:
                pass
            ratios["structured_gpu_step_vs_dense_cpu_sample_NOT_like_for_like"] = value / cb["value"]
            out["config"]["gpu_over_cpu"] = ratios
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
