"""Cube geometry shared by the host API and the device engine (no GPU needed).

The reference builds everything from four 1-D axes (inversion.py:58-74): node ("edge") coordinates per axis and
voxel-centre coordinates per axis, then expands them with meshgrid.  Here the 1-D axes are the primary objects -- the HIP
kernels take them as they are (geobo_a_sens: xe[nx+1], ye[ny+1], ze[nz+1]) -- and the reference-shaped arrays
(`Edges`, `xxx/yyy/zzz`, `voxelpos`) are broadcast views filled on demand.
"""
import numpy as np


def node_axes(s):
    """Node coordinates along x, y and z (the last already negated: z is positive down inside A_sens, inversion.py:61-66)."""
    xe = np.linspace(0, s.xNcube, s.xNcube + 1) * s.xvoxsize
    ye = np.linspace(0, s.yNcube, s.yNcube + 1) * s.yvoxsize
    ze = np.linspace(0, -s.zNcube, s.zNcube + 1) * s.zvoxsize + s.zmax
    return xe, ye, -ze


def centre_axes(s):
    """Voxel-centre coordinates along x, y and z (z runs downwards from zmax), the half-voxel ladders of inversion.py:67-69."""
    half = lambda size, length: np.arange(size / 2., length + size / 2., size)
    return half(s.xvoxsize, s.xLcube), half(s.yvoxsize, s.yLcube), s.zmax - half(s.zvoxsize, s.zLcube)


def expand(ax_x, ax_y, ax_z):
    """Three 1-D axes -> (3, len_y, len_x, len_z) array in the reference's meshgrid('xy') layout: [0] varies along axis 1,
    [1] along axis 0, [2] along axis 2."""
    out = np.empty((3, len(ax_y), len(ax_x), len(ax_z)), dtype=np.result_type(ax_x, ax_y, ax_z))
    out[0] = ax_x[None, :, None]
    out[1] = ax_y[:, None, None]
    out[2] = ax_z[None, None, :]
    return out
