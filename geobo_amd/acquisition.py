"""Drill-site proposals from the posterior cubes (the reference's BO block, run_geobo.py:175-362).

The utility of a candidate drill path P through the cube is the upper-confidence-bound score

    U(P) = sum_{v in P} mean[v]  +  kappa * sqrt( sum_{v in P} var[v] )  -  beta * sum_{v in P} cost[v]

over the voxels v the path visits (repeats count, as in the reference, which samples the path at half-voxel steps).  Two path
families are searched with SciPy's SHGO, as the reference does: vertical holes (a whole z-column of the cube, two integer
coordinates) and dipping holes (collar position, azimuth, dip).

Organisation here differs from the reference's four module-level functions over global cubes:
  * the vertical family has only (axis0 - 2) x (axis1 - 2) members, so its utility is ONE vectorised reduction of the cubes
    along z (`column_utility`), and the SHGO objective is a table lookup;
  * a dipping path is turned into voxel indices by `path_voxels` (direction cosines x a fixed ladder of ranges that is built
    once), and scored by `score`;
  * `Acquisition.futility_vertical / futility_drill / bayesopt_vert / bayesopt_nonvert` keep the reference's names, argument
    meaning, sign convention (they return MINUS the utility, for a minimiser) and failure values (inf outside the inner
    columns, 0 for a path that leaves the cube).
"""
import os

import numpy as np


def spherical2cartes(x0, y0, z0, phi, theta, r):
    """Point at range r from (x0, y0, z0) along azimuth phi / polar angle theta (radians) -- utils.py:21-36."""
    st = np.sin(theta)
    return x0 + r * st * np.cos(phi), y0 + r * st * np.sin(phi), z0 + r * np.cos(theta)


class Acquisition:
    def __init__(self, settings, drill_rec, drill_var, costs=None):
        self.s = s = settings
        self.mean = np.asarray(drill_rec)
        self.var = np.asarray(drill_var)
        self.cost = np.zeros_like(self.mean) if costs is None else np.asarray(costs)
        self.drill_rec, self.drill_var, self.costs = self.mean, self.var, self.cost      # the reference's names
        self._columns = None
        # range ladder of a dipping hole: length zLcube, two samples per smallest voxel edge (run_geobo.py:223-224)
        self.hole_length = s.zLcube
        self._ranges = np.linspace(0, self.hole_length, int(2 * self.hole_length / min(s.xvoxsize, s.yvoxsize, s.zvoxsize)))

    # ---- scoring ----------------------------------------------------------------------------------------------------------
    def score(self, index):
        """UCB utility of the voxels selected by `index` (any NumPy index into the cubes)."""
        s = self.s
        return np.sum(self.mean[index]) + s.kappa * np.sqrt(np.sum(self.var[index])) - s.beta * np.sum(self.cost[index])

    def column_utility(self):
        """Utility of every vertical hole at once: (n0, n1) table over the first two cube axes."""
        if self._columns is None:
            s = self.s
            # (C-contiguous z-columns: the row-wise reduction then adds in the same pairwise order as summing one column)
            zsum = lambda a: np.ascontiguousarray(a).sum(axis=2)
            self._columns = zsum(self.mean) + s.kappa * np.sqrt(zsum(self.var)) - s.beta * zsum(self.cost)
        return self._columns

    def path_voxels(self, x0, y0, azimuth, dip):
        """Voxel index triplets along a hole collared at (x0, y0, zmax) [m] with the given azimuth / dip [degrees]."""
        s = self.s
        x, y, z = spherical2cartes(x0, y0, s.zmax, np.deg2rad(azimuth), np.deg2rad(180. - dip), self._ranges)
        return (x / s.xvoxsize).astype(int), (y / s.yvoxsize).astype(int), (-z / s.zvoxsize).astype(int)

    # ---- objectives with the reference's names and conventions (run_geobo.py:175-235) --------------------------------------
    def futility_vertical(self, params):
        """-(utility) of the vertical hole at cube indices round(params); +inf on the rim of the cube or for non-finite input."""
        p = np.asarray(params, dtype=float)
        if not np.isfinite(p).all():
            return np.inf
        i0, i1 = int(np.round(p[0])), int(np.round(p[1]))
        table = self.column_utility()
        inner = 0 < i0 < table.shape[0] - 1 and 0 < i1 < table.shape[1] - 1
        return -table[i0, i1] if inner else np.inf

    def futility_drill(self, params):
        """-(utility) of the dipping hole params = [x0, y0, azimuth, dip]; a path that leaves the cube scores 0."""
        x0, y0, azimuth, dip = params
        try:
            return -self.score(self.path_voxels(x0, y0, azimuth, dip))
        except IndexError:
            return -0.

    # ---- searches (run_geobo.py:246-284, :306-341; the figures of the reference are not produced) ---------------------------
    def _proposals(self, found, columns, place, fname, write):
        import pandas as pd
        table = pd.DataFrame(np.round(found.xl, 2), columns=columns)
        place(table)
        table['BO_GAIN'] = -np.round(found.funl, 4)
        if write:
            table.to_csv(os.path.join(self.s.outpath, fname), index=False)
        return table

    def bayesopt_vert(self, write=True):
        """Ranked list of vertical-hole proposals (NORTHING, EASTING at voxel centres, BO_GAIN)."""
        from scipy.optimize import shgo
        s = self.s
        found = shgo(self.futility_vertical, bounds=((1, s.yNcube - 1), (1, s.xNcube - 1)), n=20, iters=20, sampling_method='sobol')
        if not found.success:
            print('WARNING: ' + found.message)

        def place(t):   # rounded voxel index -> voxel-centre coordinate
            t['EASTING'] = np.round(t['EASTING']) * s.xvoxsize + s.xmin + 0.5 * s.xvoxsize
            t['NORTHING'] = np.round(t['NORTHING']) * s.yvoxsize + s.ymin + 0.5 * s.yvoxsize
        return self._proposals(found, ['NORTHING', 'EASTING'], place, 'newdrill_proposals_vertical.csv', write)

    def bayesopt_nonvert(self, write=True, iters=500):
        """Ranked list of dipping-hole proposals (NORTHING, EASTING, AZIMUTH, DIP, BO_GAIN)."""
        from scipy.optimize import shgo
        s = self.s
        box = ((s.yvoxsize, s.yLcube - s.yvoxsize), (s.xvoxsize, s.xLcube - s.xvoxsize), (0, 360), (30, 90))
        found = shgo(self.futility_drill, box, n=10, iters=iters, sampling_method='sobol')

        def place(t):
            t['EASTING'] = np.round(t['EASTING'] + s.xmin, 1)
            t['NORTHING'] = np.round(t['NORTHING'] + s.ymin, 1)
        return self._proposals(found, ['NORTHING', 'EASTING', 'AZIMUTH', 'DIP'], place, 'newdrill_proposals_non-vertical.csv', write)
