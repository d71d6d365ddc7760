"""Bayesian-optimisation acquisition over the posterior cubes.

Mirrors `run_geobo.futility_vertical` (run_geobo.py:175-202), `futility_drill` (:205-235), `bayesopt_vert` (:246-303,
without the plot) and `bayesopt_nonvert` (:306-362) with the cubes and settings passed explicitly instead of module
globals.  Cheap host arithmetic on the (12 MB) cubes the GPU path returns; SciPy's SHGO does the search as in the reference.
"""
import os

import numpy as np


def spherical2cartes(x0, y0, z0, phi, theta, r):
    """utils.py:21-36."""
    return x0 + r * np.sin(theta) * np.cos(phi), y0 + r * np.sin(theta) * np.sin(phi), z0 + r * np.cos(theta)


class Acquisition:
    """UCB-style utility  sum(mean) + kappa*sqrt(sum(var)) - beta*sum(cost)  along a proposed drill path."""

    def __init__(self, settings, drill_rec, drill_var, costs=None):
        self.s = settings
        self.drill_rec = np.asarray(drill_rec)
        self.drill_var = np.asarray(drill_var)
        self.costs = self.drill_rec * 0. if costs is None else np.asarray(costs)

    def futility_vertical(self, params):
        """run_geobo.py:175-202 -- params = (index along axis 0, index along axis 1) of the cubes; returns -utility."""
        s = self.s
        params = np.asarray(params)
        xmaxvox = self.drill_rec.shape[0] - 1
        ymaxvox = self.drill_rec.shape[1] - 1
        if np.isfinite(params).all():
            xd, yd = int(np.round(params[0])), int(np.round(params[1]))
            if (xd > 0) & (xd < xmaxvox) & (yd > 0) & (yd < ymaxvox):
                func = (np.sum(self.drill_rec[xd, yd, :]) + s.kappa * np.sqrt(np.sum(self.drill_var[xd, yd, :]))
                        - s.beta * np.sum(self.costs[xd, yd, :]))
            else:
                func = -np.inf
        else:
            func = -np.inf
        return -func

    def futility_drill(self, params):
        """run_geobo.py:205-235 -- params = [x0, y0, azimuth, dip] (metres, degrees); returns -utility (0 on any failure)."""
        s = self.s
        length_newdrill = s.zLcube
        x0, y0, azimuth, dip = params
        nstep = int(2 * length_newdrill / np.min([s.xvoxsize, s.yvoxsize, s.zvoxsize]))
        rladder = np.linspace(0, length_newdrill, nstep)
        x0 = rladder * 0 + x0
        y0 = rladder * 0 + y0
        z0 = rladder * 0 + s.zmax
        azimuth = rladder * 0 + azimuth
        dip = rladder * 0 + dip
        try:
            xn, yn, zn = spherical2cartes(x0, y0, z0, azimuth * np.pi / 180., (180 - dip) * np.pi / 180., rladder)
            xnew = (xn / s.xvoxsize).astype(int)
            ynew = (yn / s.yvoxsize).astype(int)
            znew = (-zn / s.zvoxsize).astype(int)
            funct = (np.sum(self.drill_rec[xnew, ynew, znew]) + s.kappa * np.sqrt(np.sum(self.drill_var[xnew, ynew, znew]))
                     - s.beta * np.sum(self.costs[xnew, ynew, znew]))
        except Exception:
            funct = 0.
        return -funct

    def bayesopt_vert(self, write=True):
        """run_geobo.py:246-284 -- SHGO over the vertical utility; returns the proposals DataFrame
        (NORTHING, EASTING, BO_GAIN) and writes newdrill_proposals_vertical.csv like the reference."""
        import pandas as pd
        from scipy.optimize import shgo
        s = self.s
        res = shgo(self.futility_vertical, bounds=((1, s.yNcube - 1), (1, s.xNcube - 1)), n=20, iters=20, sampling_method='sobol')
        if not res.success:
            print('WARNING: ' + res.message)
        df = pd.DataFrame(np.round(res.xl, 2), columns=['NORTHING', 'EASTING'])
        df['EASTING'] = np.round(df['EASTING']) * s.xvoxsize + s.xmin + 0.5 * s.xvoxsize
        df['NORTHING'] = np.round(df['NORTHING']) * s.yvoxsize + s.ymin + 0.5 * s.yvoxsize
        df['BO_GAIN'] = -np.round(res.funl, 4)
        if write:
            df.to_csv(os.path.join(s.outpath, 'newdrill_proposals_vertical.csv'), index=False)
        return df

    def bayesopt_nonvert(self, write=True, iters=500):
        """run_geobo.py:306-341 -- SHGO over (y0, x0, azimuth, dip)."""
        import pandas as pd
        from scipy.optimize import shgo
        s = self.s
        bnds = ((s.yvoxsize, s.yLcube - s.yvoxsize), (s.xvoxsize, s.xLcube - s.xvoxsize), (0, 360), (30, 90))
        res = shgo(self.futility_drill, bnds, n=10, iters=iters, sampling_method='sobol')
        df = pd.DataFrame(np.round(res.xl, 2), columns=['NORTHING', 'EASTING', 'AZIMUTH', 'DIP'])
        df['EASTING'] = np.round(df['EASTING'] + s.xmin, 1)
        df['NORTHING'] = np.round(df['NORTHING'] + s.ymin, 1)
        df['BO_GAIN'] = -np.round(res.funl, 4)
        if write:
            df.to_csv(os.path.join(s.outpath, 'newdrill_proposals_non-vertical.csv'), index=False)
        return df
