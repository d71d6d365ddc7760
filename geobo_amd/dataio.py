"""Survey / drill-core ingestion and cube export around the inversion.

Mirrors the reference's `run_geobo.read_surveydata` (run_geobo.py:31-82), `read_drilldata` (:85-128), `align_drill`
(:131-158) and `cubeshow.create_vtkcube` (cubeshow.py:175-189) without rasterio / pyvista: a minimal reader for the
uncompressed striped GeoTIFFs GeoBO ships and a legacy-binary VTK writer that reproduces pyvista's file byte for byte.
Host-side data handling (NumPy / pandas / scipy.ndimage.zoom, exactly the reference's own tools); no inversion arithmetic.
"""
import os
import struct

import numpy as np


# ---- GeoTIFF (uncompressed, striped, one band of float32 / float64) ------------------------------------------------
def read_tiff(path):
    """First band of an uncompressed striped TIFF as a NumPy array in the file's own float dtype
    (what `rasterio.open(path).read(1)` returns for the reference's inputs, run_geobo.py:45-52)."""
    b = open(path, "rb").read()
    e = "<" if b[:2] == b"II" else ">"
    if struct.unpack(e + "H", b[2:4])[0] != 42:
        raise ValueError("not a classic TIFF: %s" % path)
    off = struct.unpack(e + "I", b[4:8])[0]
    n = struct.unpack(e + "H", b[off:off + 2])[0]
    tags = {}
    for i in range(n):
        t, ty, cnt, val = struct.unpack(e + "HHI4s", b[off + 2 + 12 * i: off + 14 + 12 * i])
        size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 11: 4, 12: 8, 16: 8}.get(ty)
        if size is None or ty not in (3, 4):
            continue
        data = val[: size * cnt] if size * cnt <= 4 else b[struct.unpack(e + "I", val)[0]:][: size * cnt]
        tags[t] = list(struct.unpack(e + ("H" if ty == 3 else "I") * cnt, data))
    w, h, bits = tags[256][0], tags[257][0], tags[258][0]
    if tags.get(259, [1])[0] != 1:
        raise ValueError("compressed TIFFs are not supported")
    if tags.get(339, [1])[0] != 3 or bits not in (32, 64) or tags.get(277, [1])[0] != 1:
        raise ValueError("expected one band of IEEE float samples")
    raw = b"".join(b[o: o + c] for o, c in zip(tags[273], tags[279]))
    return np.frombuffer(raw, dtype=np.dtype(e + "f" + str(bits // 8))).reshape(h, w).astype("f" + str(bits // 8))


def read_surveydata(s, plot=False):
    """run_geobo.py:31-82 -- gravity / magnetic grids cropped-and-zoomed to (yNcube, xNcube), flattened, plus the sensor
    locations above the voxel centres.  Returns (grav, mag, locations); the arrays keep the TIFF's dtype."""
    from scipy.ndimage import zoom
    grav = read_tiff(os.path.join(s.inpath, s.FNAME_gravsurvey)) if s.FNAME_gravsurvey is not None else None
    mag = read_tiff(os.path.join(s.inpath, s.FNAME_magsurvey)) if s.FNAME_magsurvey is not None else None
    grav2 = zoom(grav, s.xNcube * 1. / grav.shape[1])
    assert grav2.shape == (s.yNcube, s.xNcube)
    mag2 = zoom(mag, s.xNcube * 1. / mag.shape[1])
    assert mag2.shape == (s.yNcube, s.xNcube)
    x_s = np.linspace(0.5, s.xNcube - 0.5, s.xNcube) * s.xvoxsize
    y_s = np.linspace(0.5, s.yNcube - 0.5, s.yNcube) * s.yvoxsize
    z_s = s.zmax + s.zoff
    xs, ys, zs = np.meshgrid(x_s, y_s, z_s)
    locations = np.asarray([xs.flatten(), ys.flatten(), zs.flatten()]).T
    return grav2.flatten(), mag2.flatten(), locations


def align_drill(coord, data, voxelpos, s):
    """run_geobo.py:131-158 -- mean of the drill samples inside the (2 voxel wide) window around every voxel centre:
    (c - d) <= sample < (c + d) on each axis; voxels without samples stay 0.  `voxelpos` = Inversion.create_cubegeometry().
    Returns the cube in the reference's (xNcube, yNcube, zNcube) shape (flat order = voxel order).  Vectorised per axis:
    the voxel centres take only nx / ny / nz distinct values, so membership is three small boolean tables."""
    coord = np.asarray(coord, dtype=np.float64)
    data = np.asarray(data, dtype=np.float64)
    x, y, z = (np.asarray(v, dtype=np.float64) for v in voxelpos)
    good = np.isfinite(data)

    def axis_table(centres, d, samples):
        u, inv = np.unique(centres, return_inverse=True)
        return ((u[:, None] - d) <= samples[None, :]) & (samples[None, :] < (u[:, None] + d)), inv

    bx, ix = axis_table(x, s.xvoxsize, coord[:, 0])
    by, iy = axis_table(y, s.yvoxsize, coord[:, 1])
    bz, iz = axis_table(z, s.zvoxsize, coord[:, 2])
    w = good.astype(np.float64)
    dv = np.where(good, data, 0.0)
    # sums over samples of Bx[ix,s] By[iy,s] Bz[iz,s] {1, data}: contract z last to keep the intermediate small
    bxy_w = np.einsum("as,bs->abs", bx.astype(np.float64), by.astype(np.float64))
    cnt = np.einsum("abs,cs->abc", bxy_w * w[None, None, :], bz.astype(np.float64))
    tot = np.einsum("abs,cs->abc", bxy_w * dv[None, None, :], bz.astype(np.float64))
    any_sel = np.einsum("abs,cs->abc", bxy_w, bz.astype(np.float64)) > 0
    with np.errstate(all="ignore"):
        mean = tot / cnt
    res_u = np.where(any_sel & (cnt > 0) & np.isfinite(mean), mean, 0.0)
    res = res_u[ix, iy, iz]
    return res.reshape(s.xNcube, s.yNcube, s.zNcube)


def read_drilldata(s, features, voxelpos):
    """run_geobo.py:85-128 -- drill-core CSV -> per-feature voxel cubes, local coordinates, per-site first/last points."""
    import pandas as pd
    drill = pd.read_csv(os.path.join(s.inpath, s.FNAME_drilldata))
    drill = drill[(drill.x >= s.xmin) & (drill.x <= s.xmax) & (drill.y >= s.ymin) & (drill.y <= s.ymax)
                  & (drill.z <= s.zmax) & (drill.z >= s.zmin)].copy()
    drill['x'] = drill['x'] - s.xmin
    drill['y'] = drill['y'] - s.ymin
    xd, yd, zd = drill['x'].values, drill['y'].values, drill['z'].values
    try:
        first, last = drill.groupby('SiteID').first(), drill.groupby('SiteID').last()
        minmax = tuple(np.asarray([first[c].values, last[c].values]).T for c in ("x", "y", "z"))
    except Exception:
        minmax = (0., 0., 0.)
    coord = np.vstack([xd, yd, zd]).T
    cubes = [align_drill(coord, drill[f], voxelpos, s) for f in features]
    return np.asarray(cubes), coord, minmax


# ---- VTK export ----------------------------------------------------------------------------------------------------------
def _g(v):
    return ("%.15g" % float(v))


def create_vtkcube(density, origin, voxelsize, fname):
    """cubeshow.py:175-189 -- legacy binary VTK (STRUCTURED_POINTS, CELL_DATA `values`, big-endian doubles, Fortran order),
    byte-identical to what pyvista's UniformGrid.save writes for the reference."""
    density = np.asarray(density, dtype=np.float64)
    dims = np.array(density.shape) + 1
    vals = density.flatten(order="F")
    head = ("# vtk DataFile Version 4.2\nvtk output\nBINARY\nDATASET STRUCTURED_POINTS\n"
            "DIMENSIONS %d %d %d\nSPACING %s %s %s\nORIGIN %s %s %s\nCELL_DATA %d\nSCALARS values double\nLOOKUP_TABLE default\n"
            % (dims[0], dims[1], dims[2], _g(voxelsize[0]), _g(voxelsize[1]), _g(voxelsize[2]), _g(origin[0]), _g(origin[1]),
               _g(origin[2]), vals.size))
    with open(fname, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(vals.astype(">f8").tobytes())
        f.write(b"\n")


def read_vtkcube(fname):
    """Inverse of create_vtkcube: (cube, origin, spacing)."""
    b = open(fname, "rb").read()
    hdr = b[: b.index(b"LOOKUP_TABLE default\n")].decode("ascii").split("\n")
    get = lambda key: [x for x in hdr if x.startswith(key)][0].split()[1:]
    dims = [int(v) - 1 for v in get("DIMENSIONS")]
    k = b.index(b"LOOKUP_TABLE default\n") + len(b"LOOKUP_TABLE default\n")
    n = dims[0] * dims[1] * dims[2]
    cube = np.frombuffer(b[k:k + 8 * n], dtype=">f8").astype(np.float64).reshape(dims, order="F")
    return cube, [float(v) for v in get("ORIGIN")], [float(v) for v in get("SPACING")]
