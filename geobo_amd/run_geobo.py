"""YAML-driven workflow around the MI355X inversion -- the reference's `run_geobo.py:380-469` main flow as a function.

    python -m geobo_amd.run_geobo settings.yaml

read survey + drill data -> Inversion.cubing (GPU) -> six VTK cubes -> optional BO proposals.  Plotting (matplotlib /
pyvista) and the synthetic-model generator (`gen_simulation`) of the reference are not part of this package.
"""
import os
import sys

import numpy as np

from . import config_loader, dataio
from .acquisition import Acquisition
from .inversion import Inversion


def run(settings, method="auto", bayesopt=True, write=True):
    s = config_loader.load(settings, create_outpath=write)
    if getattr(s, "gen_simulation", False):
        print("gen_simulation is not supported here: using the existing input files")
    inv = Inversion(settings=s, method=method)
    voxelpos = inv.create_cubegeometry()
    # run_geobo.py:400-403 -- the reference re-shapes the centre arrays to (xN, yN, zN); flat order is unchanged
    for name in ("xxx", "yyy", "zzz"):
        setattr(inv, name, getattr(inv, name).reshape(s.xNcube, s.yNcube, s.zNcube))
    gravfield, magfield, sensor_locations = dataio.read_surveydata(s)
    drilldata, drillcoord, drillminmax = dataio.read_drilldata(s, s.drill_features, voxelpos)
    drilldata0 = drilldata[s.ifeature]
    drillfield = drilldata0[drilldata0 != 0]
    cubes = inv.cubing(gravfield, magfield, drillfield, sensor_locations, drilldata0)
    names = ["cube_density", "cube_magsus", "cube_drill", "cube_density_variance", "cube_magsus_variance", "cube_drill_variance"]
    if write:
        origin = (voxelpos[0].min(), voxelpos[1].min(), voxelpos[2].min())
        voxelsize = (s.xvoxsize, s.yvoxsize, s.zvoxsize)
        for c, n in zip(cubes, names):
            dataio.create_vtkcube(c, origin, voxelsize, fname=os.path.join(s.outpath, n + ".vtk"))
    out = dict(zip(names, cubes), inversion=inv, drillcoord=drillcoord, inputs=(gravfield, magfield, drillfield, sensor_locations, drilldata0))
    if bayesopt:
        acq = Acquisition(s, cubes[2], cubes[5])
        if getattr(s, "bayesopt_vertical", False):
            out["proposals_vertical"] = acq.bayesopt_vert(write=write)
        if getattr(s, "bayesopt_nonvertical", False):
            out["proposals_nonvertical"] = acq.bayesopt_nonvert(write=write)
    return out


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit("usage: python -m geobo_amd.run_geobo settings.yaml")
    run(sys.argv[1])
