#!/usr/bin/env python3
"""Whole-cube golden vectors on 64 x 48 x 64 -- the smallest grid on which the HIP path's radix-2 (x, z) transforms, fused lattice
Gram, fused sum of squares and lattice form of Z = L^-1 A all run (the kernel family of the 64^3 headline) -- Matern-3/2, 20 drill
rows, three property blocks.  Neither the reference (its kcov alone would be 2.8 TB) nor the oracle's column-blocked form reaches
this size; these come from `oracle.geobo_oracle.cubing(fft=True)`: sensor rows of A K by FFT convolution (`ak_rows_fft`, pinned to
the direct contraction and, through `cubing(fft=True)`, to the reference's cubes in tests/test_oracle_golden.py), AkA, scipy
Cholesky, V = L^-1 (A K) column-blocked.  ~15 minutes and ~45 GB on 8 cores.
Usage: python tests/golden/make_oracle64.py"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import geobo_oracle as O  # noqa: E402

NX, NY, NZ, MD = 64, 48, 64, 20
GL = [200.0, 202.0, 204.0]
G = O.Grid(nx=NX, ny=NY, nz=NZ, xmax=100.0 * NX, ymax=100.0 * NY, zLcube=100.0 * NZ, kernelfunc="matern32")
LOC = G.sensor_locations()
EDGES = G.edges()


def _rows(job):
    func, r0, r1 = job
    return O.a_sens(G, G.B * (0. if func == "grav" else 1.), LOC, EDGES, func, rows=range(r0, r1))


def operator(func, workers):
    ms = NX * NY
    jobs = [(func, r, min(ms, r + 64)) for r in range(0, ms, 64)]
    with ProcessPoolExecutor(workers) as ex:
        return np.vstack(list(ex.map(_rows, jobs)))


if __name__ == "__main__":
    workers = os.cpu_count() or 1
    t0 = time.time()
    say = lambda *a: print("[%5.0f s]" % (time.time() - t0), *a, flush=True)  # noqa: E731
    A = (operator("grav", workers), operator("magn", workers))
    say("operators", A[0].shape)
    sv = O.synthetic_survey(G, MD, A=A)
    d0 = sv["drilldata0"]
    r = O.cubing(G, sv["gravfield"], sv["magfield"], d0[d0 != 0], sv["sensor_locations"], d0, gp_length=np.array(GL), A=A,
                 fft=True, workers=workers)
    say("cubing")
    sel = np.flatnonzero(d0.reshape(-1) != 0)
    np.savez_compressed(os.path.join(HERE, "oracle64x48_matern32.npz"), dims=np.array([NX, NY, NZ]), gravfield=sv["gravfield"],
                        magfield=sv["magfield"], sensor_locations=sv["sensor_locations"], sel=sel,
                        drillvalues=d0.reshape(-1)[sel], cubes=r["cubes"].astype(np.float64), logl=r["logl"],
                        gp_length_out=r["gp_length"], gp_length_in=np.array(GL), L_diag=np.diag(r["L"]).copy(),
                        A_g_rowsum=A[0].sum(axis=1), A_m_rowsum=A[1].sum(axis=1))
    say("wrote oracle64x48_matern32.npz")
