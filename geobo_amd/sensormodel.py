"""Forward models -- the reference's `geobo/sensormodel.py` API on the MI355X path.

`A_sens`, `grav_func`, `magn_func`, `A_drill` keep the reference signatures (NumPy in / NumPy out) and run
in the gfx950 kernels of libgeobo_hip.so.  `Inversion.cubing` does not call these host-returning wrappers:
it keeps the operators resident on the device (engine.PosteriorEngine.operator).
"""
import numpy as np
import torch

from . import config_loader, hip


def _edge_axes(Edges, nx, ny, nz):
    """Recover the 1-D node axes from the reference's meshgrid-shaped Edges (3, ny+1, nx+1, nz+1)
    (inversion.py:58-66) and check that Edges really is such a tensor-product grid."""
    E = np.asarray(Edges, dtype=np.float64)
    if E.shape != (3, ny + 1, nx + 1, nz + 1):
        raise ValueError("Edges must have shape (3, yNcube+1, xNcube+1, zNcube+1)")
    xe, ye, ze = E[0][0, :, 0].copy(), E[1][:, 0, 0].copy(), E[2][0, 0, :].copy()
    if not (np.array_equal(E[0], np.broadcast_to(xe[None, :, None], E[0].shape))
            and np.array_equal(E[1], np.broadcast_to(ye[:, None, None], E[1].shape))
            and np.array_equal(E[2], np.broadcast_to(ze[None, None, :], E[2].shape))):
        raise ValueError("Edges is not a tensor-product (meshgrid) node grid")
    return xe, ye, ze


def A_sens(magneticField, locations, Edges, func, settings=None):
    """sensormodel.py:29-93 -- gravity ('grav') or magnetic ('magn') sensitivity matrix.

    Returns (sens, result_ez) like the reference; `result_ez` (the per-sensor node potentials, unused by
    every caller in the reference: inversion.py:223-224, simcube.py:147-148) is returned as None."""
    s = settings or config_loader.active()
    if func not in ("grav", "magn"):
        print('function not supported')  # sensormodel.py:75-76
        raise ValueError(func)
    nx, ny, nz = int(s.xNcube), int(s.yNcube), int(s.zNcube)
    xe, ye, ze = _edge_axes(Edges, nx, ny, nz)
    loc = np.asarray(locations, dtype=np.float64)
    if loc.ndim != 2 or loc.shape[0] < nx * ny or loc.shape[1] < 3:
        # the reference loops over exactly xNcube*yNcube sensors (sensormodel.py:54,58) and runs off the end of a shorter list
        raise IndexError("A_sens needs xNcube*yNcube = %d sensor locations (x, y, z), got an array of shape %s"
                         % (nx * ny, loc.shape))
    loc = np.ascontiguousarray(loc[:nx * ny, :3])
    N = nx * ny * nz
    A = torch.empty((nx * ny, N), dtype=hip.F64, device="cuda")
    if func == "grav":
        mul, div = s.c_MILLIGALS_UNITS, s.fcor_grav
    else:
        mul, div = 1.0, s.fcor_mag
    hip.a_sens(func, np.asarray(magneticField, dtype=float), hip.to_dev(loc), nx, ny, nz, hip.to_dev(xe), hip.to_dev(ye),
               hip.to_dev(ze), mul, div, A)
    return A.cpu().numpy(), None


def grav_func(x, y, z):
    """sensormodel.py:96-110 -- vertical gravity potential term of a prism corner."""
    x, y, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(y, dtype=float), np.asarray(z, dtype=float))
    out = hip.potential("grav", (0., 0., 0.), hip.to_dev(x.reshape(-1)), hip.to_dev(y.reshape(-1)), hip.to_dev(z.reshape(-1)))
    return out.cpu().numpy().reshape(x.shape)


def magn_func(x, y, z, bx, by, bz):
    """sensormodel.py:113-133 -- magnetic potential term of a prism corner."""
    x, y, z = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(y, dtype=float), np.asarray(z, dtype=float))
    out = hip.potential("magn", (bx, by, bz), hip.to_dev(x.reshape(-1)), hip.to_dev(y.reshape(-1)), hip.to_dev(z.reshape(-1)))
    return out.cpu().numpy().reshape(x.shape)


def drill_index(loc, voxelpos):
    """Flat voxel index of every drill coordinate (exact float match, sensormodel.py:147-152); -1 if none."""
    loc = np.asarray(loc, dtype=np.float64)
    x, y, z = (np.asarray(v).flatten() for v in voxelpos)
    idx = np.full(loc.shape[1], -1, dtype=np.int64)
    for i in range(loc.shape[1]):
        hit = np.flatnonzero((x == loc[0, i]) & (y == loc[1, i]) & (z == loc[2, i]))
        if hit.size:
            idx[i] = hit[0]
    return idx


def A_drill(loc, voxelpos, settings=None):
    """sensormodel.py:136-153 -- 0/1 selection matrix (Ndrill, Nvoxel) of the drilled voxels."""
    idx = drill_index(loc, voxelpos)
    n = np.asarray(voxelpos[0]).size
    sens = np.zeros((idx.size, n))
    ok = idx >= 0
    sens[np.flatnonzero(ok), idx[ok]] = 1
    return sens
