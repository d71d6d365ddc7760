#!/usr/bin/env python3
"""Golden posterior VALUES of the headline configuration itself -- 64^3, Matern-3/2 (2.00, 2.02, 2.04) x 100 m, 50 drill rows, property
blocks 0 and 1 (BASELINE.json config 3) -- at a SPREAD sample of voxels, computed with nothing from the device: the oracle's own
operators (`a_sens`, sensormodel.py:29-93), rows of A K by FFT convolution (`ak_rows_fft`, pinned to the direct contraction and to the
reference's cubes in tests/test_oracle_golden.py), AkA, scipy Cholesky, V = L^-1 (A K) on the sampled columns (inversion.py:92-117).

The whole-cube form (`cubing(fft=True)`, make_oracle64.py) holds A K (35 GB) next to the operators (17 GB): too much for this
container at 64^3.  Here the rows of A K are formed batch by batch, contracted into AkA at once and only the sampled columns kept --
the same arithmetic, 20 GB.  The sample: every 131st voxel, all 50 drilled voxels, 256 voxels of each padded slab iy = 0 and
iy = ny - 1 (the +-1e6 m quirk of A_sens), 128 of each x / z face, the 8 corners.  ~25 minutes on 8 cores.
Usage: python tests/golden/make_oracle64_sample.py"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np
from scipy.linalg import cholesky, solve_triangular

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import geobo_oracle as O  # noqa: E402

N1, MD = 64, 50
GL = [200.0, 202.0, 204.0]
G = O.Grid(nx=N1, ny=N1, nz=N1, xmax=100.0 * N1, ymax=100.0 * N1, zLcube=100.0 * N1, kernelfunc="matern32")
LOC = G.sensor_locations()
EDGES = G.edges()


def _rows(job):
    func, r0, r1 = job
    return O.a_sens(G, G.B * (0. if func == "grav" else 1.), LOC, EDGES, func, rows=range(r0, r1))


def operator(func, workers):
    ms = G.nx * G.ny
    jobs = [(func, r, min(ms, r + 64)) for r in range(0, ms, 64)]
    with ProcessPoolExecutor(workers) as ex:
        return np.vstack(list(ex.map(_rows, jobs)))


def sample_voxels(sel):
    ny, nx, nz = G.ny, G.nx, G.nz
    N = G.N
    rng = np.random.default_rng(64)
    iy, ix, iz = np.unravel_index(np.arange(N), (ny, nx, nz))
    pick = [np.arange(0, N, 131), np.asarray(sel)]
    for mask, k in ((iy == 0, 256), (iy == ny - 1, 256), (ix == 0, 128), (ix == nx - 1, 128), (iz == 0, 128), (iz == nz - 1, 128)):
        pick.append(rng.choice(np.flatnonzero(mask), k, replace=False))
    pick.append(np.array([(y * nx + x) * nz + z for y in (0, ny - 1) for x in (0, nx - 1) for z in (0, nz - 1)]))
    return np.unique(np.concatenate(pick))


if __name__ == "__main__":
    workers = os.cpu_count() or 1
    t0 = time.time()
    say = lambda *a: print("[%5.0f s]" % (time.time() - t0), *a, flush=True)  # noqa: E731
    cache = os.environ.get("GEOBO_ORACLE_CACHE")         # optional directory: the two operators (17 GB) survive a restart of this script
    A = []
    for func in ("grav", "magn"):
        f = os.path.join(cache, "oracle64_A_%s.npy" % func) if cache else None
        if f and os.path.exists(f):
            A.append(np.load(f))
        else:
            A.append(operator(func, workers))
            if f:
                np.save(f, A[-1])
    A = tuple(A)
    say("operators", A[0].shape)
    sv = O.synthetic_survey(G, MD, A=A)
    d0 = sv["drilldata0"]
    sel = O.drill_selection(d0)
    gravfield, magfield, drillfield = sv["gravfield"], sv["magfield"], d0[d0 != 0]
    # inversion.py:209-214: the data are z-scored with the population std (cubing() of the oracle, same lines)
    gs, ms, ds = gravfield.std(), magfield.std(), drillfield.std()
    y = np.hstack([(gravfield - gravfield.mean()) / gs, (magfield - magfield.mean()) / ms, (drillfield - drillfield.mean()) / ds])
    lengths = O.mutate_lengths(np.array(GL))
    W = O.weight_matrix(G.gp_coeff)
    P3 = O.grid_points((G.nx, G.ny, G.nz), (G.sx, G.sy, G.sz))
    name = G.kernelfunc
    N, mg, mm, md = G.N, A[0].shape[0], A[1].shape[0], len(sel)
    M = mg + mm + md
    q = sample_voxels(sel)
    say("sample of %d voxels" % q.size)
    AkA = np.zeros((M, M))
    S = {j: np.empty((M, q.size)) for j in (0, 1)}           # the sampled columns of A K, property blocks 0 and 1
    B = 128
    ck = os.path.join(cache, "oracle64_sample_state.npz") if cache else None
    done = 0                                             # sensor rows (of both operators, in order) already contracted
    if ck and os.path.exists(ck):
        st = np.load(ck)
        if np.array_equal(st["q"], q):
            AkA, S[0], S[1], done = st["AkA"], st["S0"], st["S1"], int(st["done"])
            say("resumed at sensor row %d" % done)
    for s, As, r_off in ((0, A[0], 0), (1, A[1], mg)):
        for r0 in range(0, As.shape[0], B):
            r1 = min(As.shape[0], r0 + B)
            if r_off + r1 <= done:
                continue
            if ck and r0 % 1024 == 0 and r_off + r0 > done:
                np.savez(ck, AkA=AkA, S0=S[0], S1=S[1], done=r_off + r0, q=q)
            w = O.ak_rows_fft(G, As[r0:r1], name, lengths, W, s, (0, 1), workers=workers)
            AkA[r_off + r0:r_off + r1, :mg] = w[0] @ A[0].T
            AkA[r_off + r0:r_off + r1, mg:mg + mm] = w[1] @ A[1].T
            for j in (0, 1):
                S[j][r_off + r0:r_off + r1] = w[j][:, q]
            if r0 % 1024 == 0:
                say("A K rows of operator %d: %d" % (s, r1))
    D2s = O.sqdist(P3[sel], P3)
    for j in (0, 1):
        kd = O.k_block(name, D2s, lengths, W, 2, j)
        AkA[mg + mm:, (0, mg)[j]:(mg, mg + mm)[j]] = kd @ A[j].T
        S[j][mg + mm:] = kd[:, q]
    AkA[:mg + mm, mg + mm:] = AkA[mg + mm:, :mg + mm].T
    AkA[mg + mm:, mg + mm:] = O.k_block(name, O.sqdist(P3[sel]), lengths, W, 2, 2)
    AkA = AkA + np.diag(O._noise(G.gp_err, mg, mm, md) ** 2)
    say("AkA")
    L = cholesky(AkA, lower=True)
    u = solve_triangular(L, y, lower=True)
    logl = -0.5 * (u @ u + np.log(np.diag(L) ** 2).sum() + N * np.log(2 * np.pi))
    mu, var = np.empty((2, q.size)), np.empty((2, q.size))
    for j in (0, 1):
        V = solve_triangular(L, S[j], lower=True)
        mu[j] = V.T @ u
        var[j] = 1.0 - np.einsum("mq,mq->q", V, V)
    say("posterior at the sample")
    rows_kept = np.array([0, 37, mg - 1, mg, mg + 2077, mg + mm - 1, mg + mm, M - 1])
    np.savez_compressed(os.path.join(HERE, "oracle64_sample_matern32.npz"), dims=np.array([G.nx, G.ny, G.nz]), gravfield=gravfield,
                        magfield=magfield, sensor_locations=sv["sensor_locations"], sel=sel, drillvalues=d0.reshape(-1)[sel],
                        voxels=q, mu=mu, var=var, data_std=np.array([gs, ms, ds]), logl=logl, gp_length_out=lengths,
                        gp_length_in=np.array(GL), L_diag=np.diag(L).copy(), AkA_rows=rows_kept, AkA_values=AkA[rows_kept],
                        A_g_rowsum=A[0].sum(axis=1), A_m_rowsum=A[1].sum(axis=1))
    say("wrote oracle64_sample_matern32.npz")
