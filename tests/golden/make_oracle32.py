#!/usr/bin/env python3
"""Golden vectors for BASELINE config 2 (32^3, gravity+magnetics joint, sq-exp cross-kernel) and a 32^3 Matern-3/2 case
with 50 drill rows.  The reference cannot run 32^3 (its kcov alone is 77 GB), so these are produced by the ORACLE
(oracle/geobo_oracle.py, matrix-free form), which tests/test_oracle_golden.py pins to the reference at the sizes the
reference can run.  Takes ~10 minutes of CPU.   Usage: python tests/golden/make_oracle32.py"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import geobo_oracle as O  # noqa: E402

n = 32
for name, kern, md, gl in (("oracle32_exp", "exp", 0, None), ("oracle32_matern32", "matern32", 50, [200.0, 202.0, 204.0])):
    t0 = time.time()
    G = O.Grid(nx=n, ny=n, nz=n, xmax=100.0 * n, ymax=100.0 * n, zLcube=100.0 * n, kernelfunc=kern)
    sv = O.synthetic_survey(G, md)
    d0 = sv["drilldata0"]
    r = O.cubing(G, sv["gravfield"], sv["magfield"], d0[d0 != 0], sv["sensor_locations"], d0,
                 gp_length=None if gl is None else np.array(gl), A=sv["A"], block=2048)
    sel = np.flatnonzero(d0.reshape(-1) != 0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), gravfield=sv["gravfield"], magfield=sv["magfield"],
                        sensor_locations=sv["sensor_locations"], sel=sel, drillvalues=d0.reshape(-1)[sel],
                        cubes=r["cubes"].astype(np.float64), logl=r["logl"], gp_length_out=r["gp_length"],
                        gp_length_in=np.array(gl) if gl is not None else G.default_gp_length(),
                        A_g_rowsum=r["A_g"].sum(axis=1), A_m_rowsum=r["A_m"].sum(axis=1))
    print("wrote", name, "%.0f s" % (time.time() - t0), flush=True)
