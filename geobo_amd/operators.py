"""Forward operators that are never resident (engine.operator builds them; the row form, the spectral product's lattice feed and
the streamed column form read them).  Reference: sensormodel.py:29-93 (A_sens materialises the Ms x N matrix)."""
import torch

from . import hip


class StreamedOperator:
    """A forward operator that is never resident: its rows (all voxels of a batch of sensors) or its column slabs (one y-range
    of every sensor) are generated on demand straight from the survey geometry -- on a lattice survey by one contiguous copy
    per (sensor, y-slab) out of the stencil table Q (geobo_a_sens_lattice), otherwise by the direct kernel.  What BASELINE
    config 5 needs: at 128^3 one operator is 275 GB (SURVEY.md section 8 size table)."""

    def __init__(self, eng, func, Bv, mul, div, locd, axes_dev, plan, lws):
        self.eng, self.func, self.Bv, self.mul, self.div = eng, func, Bv, mul, div
        self.locd, self.axes_dev, self.plan, self.lws = locd, axes_dev, plan, lws
        self.lattice = None     # spectral.LatticeRows: the transform reads the rows as windows of the stencil table (keep_stencil)

    def keep_stencil(self, name):
        """Lattice survey: keep this operator's stencil table Q (63 MB at 64^3; the lattice workspace is shared between the
        operators) and its two 1e6-padded boundary slabs for every sensor, and describe the rows as windows of Q -- the forward
        transform then reads the table, which stays in cache, and no operator row is ever written or read."""
        from .spectral import LatticeRows
        e = self.eng
        nx, ny, nz, plane = e.nx, e.ny, e.nz, e.nx * e.nz
        nqx = 2 * nx - 1
        Q = e._workspace("lattice_Q_" + name, (2 * ny - 3, nqx, nz))
        Q.copy_(hip.a_sens_lattice_stencil(self.lws, nx, ny, nz))
        row_off = (((ny - 2 - self.plan["jy"].to(torch.int64)) * nqx + (nx - 1 - self.plan["jx"].to(torch.int64))) * nz).contiguous()
        E2 = e._workspace2d("Aedge_" + name, e.Ms_pad, 2 * plane)
        if e.Ms_pad > e.Ms:
            E2[e.Ms:].zero_()
        xed, yed, zed = self.axes_dev
        for k, iy in enumerate((0, ny - 1)):
            hip.a_sens(self.func, self.Bv, self.locd, nx, ny, nz, xed, yed, zed, self.mul, self.div, E2[:, k * plane:(k + 1) * plane], iy, iy + 1,
                       plan=self.plan, ws=self.lws, col_origin=iy * plane)
        self.edge = E2
        self.lattice = LatticeRows(Q.view(-1), row_off, nqx * nz, E2)

    def rows_into(self, buf, r0, R):
        """buf[:R, :N_pad] <- operator rows r0 .. r0+R-1 (voxel padding columns zero)."""
        e = self.eng
        out = buf[:R, :e.N_pad]
        if e.N_pad > e.N:
            out[:, e.N:].zero_()
        xed, yed, zed = self.axes_dev
        hip.a_sens(self.func, self.Bv, self.locd[r0:r0 + R].contiguous(), e.nx, e.ny, e.nz, xed, yed, zed, self.mul, self.div, out,
                   plan=self.plan, rows=slice(r0, r0 + R), ws=self.lws)
        return out

    def slab_into(self, buf, iy0, iy1):
        """buf[:Ms_pad, :(iy1-iy0)*nx*nz] <- columns of the y-slabs iy0 .. iy1-1 for every sensor (rows >= Ms zero)."""
        e = self.eng
        w = (iy1 - iy0) * e.nx * e.nz
        out = buf[:e.Ms_pad, :w]
        if e.Ms_pad > e.Ms:
            out[e.Ms:].zero_()
        xed, yed, zed = self.axes_dev
        hip.a_sens(self.func, self.Bv, self.locd, e.nx, e.ny, e.nz, xed, yed, zed, self.mul, self.div, out, iy0, iy1,
                   plan=self.plan, ws=self.lws, col_origin=iy0 * e.nx * e.nz)
        return out
