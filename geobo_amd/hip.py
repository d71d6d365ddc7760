"""Tensor-level wrappers over the C ABI (include/geobo_hip.h).

PyTorch is plumbing here: device memory (torch.float64 CUDA tensors), the current HIP stream and
torch.distributed; every numerical operation of the hot path is a hand-written gfx950 kernel in
libgeobo_hip.so.  All functions are asynchronous on torch's current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

F64 = torch.float64

KERNEL_IDS = {"d2": 0, "exp": 1, "exp_x": 2, "matern32": 3, "matern32_x": 4, "sparse": 5, "sparse_x": 6}
FUNC_IDS = {"grav": 0, "magn": 1}
# kernel-instance tables and padding units: ONE definition (plan.py, which decides routes from them on the CPU); re-exported here
from .plan import PAD_M, PAD_N, SPECTRAL_AXIS_N, SPECTRAL_Y_NY, TOEPLITZ_NY, XZ2D_FOLD_N, XZ2D_SHAPES  # noqa: E402,F401


def kernel_id(name, cross):
    """(kernelfunc, is-cross-block) -> family id of include/geobo_hip.h (kernels.py:183-195)."""
    if name not in ("exp", "matern32", "sparse"):
        raise ValueError("unknown kernelfunc %r (expected 'sparse', 'exp' or 'matern32')" % (name,))
    return KERNEL_IDS[name + ("_x" if cross else "")]


def pad_m(m):
    return (int(m) + PAD_M - 1) // PAD_M * PAD_M


def pad_n(n):
    return (int(n) + PAD_N - 1) // PAD_N * PAD_N


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.GeoboHipUnavailable("no MI355X / ROCm device visible: the GeoBO hot path has no CPU fallback")
    return _lib.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _chk(t, name, dtype=F64):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype):
        raise TypeError("%s must be a CUDA %s tensor" % (name, dtype))
    return t


def _rowmajor(t, name):
    _chk(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("%s must be 2-D with unit column stride" % name)
    return t.stride(0)


def version():
    return _lib.load().geobo_version()


F32 = torch.float32


def k_block(kid, rows_xyz, cols_xyz, l1, l2, w, amp, out):
    """out[r,c] = w*amp*k(|rows_r - cols_c|^2).  rows_xyz / cols_xyz: tuples of three 1-D tensors; out fp64 or fp32."""
    lib = require_gpu()
    rx, ry, rz = (_chk(t, "rows") for t in rows_xyz)
    cx, cy, cz = (_chk(t, "cols") for t in cols_xyz)
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype in (F64, F32) and out.dim() == 2 and out.stride(1) == 1):
        raise TypeError("out must be a 2-D CUDA float64 / float32 tensor with unit column stride")
    ld = out.stride(0)
    nr, nc = rx.numel(), cx.numel()
    assert out.shape[0] >= nr and out.shape[1] >= nc
    fn, name = (lib.geobo_k_block, "geobo_k_block") if out.dtype == F64 else (lib.geobo_k_block_f32, "geobo_k_block_f32")
    _lib.check(fn(kid, _p(rx), _p(ry), _p(rz), nr, _p(cx), _p(cy), _p(cz), nc, float(l1), float(l2), float(w), float(amp),
                  _p(out), ld, _stream()), name)
    return out


def k_block_grid(table, nx, ny, nz, rows, col0, out, row0=0, nr=None):
    """Covariance block on the regular grid gathered from the lattice table of cov_table: out[r, c] for row voxels `rows` (int64
    device tensor of flat voxel indices, or None for the range row0 .. row0 + nr) and column voxels col0 .. col0 + out.shape[1]."""
    lib = require_gpu()
    if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype in (F64, F32) and out.dim() == 2 and out.stride(1) == 1):
        raise TypeError("out must be a 2-D CUDA float64 / float32 tensor with unit column stride")
    if rows is not None:
        assert rows.dtype == torch.int64 and rows.is_cuda and rows.is_contiguous()
        nr = rows.numel()
    nr = out.shape[0] if nr is None else int(nr)
    assert out.shape[0] >= nr and _chk(table, "table").numel() >= 2 * nx * ny * nz
    _lib.check(lib.geobo_k_block_grid(int(nx), int(ny), int(nz), _p(table), C.c_void_p(rows.data_ptr()) if rows is not None else None,
                                      int(row0), nr, int(col0), out.shape[1], 1 if out.dtype == F32 else 0, _p(out), out.stride(0),
                                      _stream()), "geobo_k_block_grid")
    return out


def lattice_wbuild(rows, Py, Px, nz, lamW, lhat, W):
    """W[r][kx][iz][ky] = lamW[kx][iz][ky] * lhat[r][ky][kx]  (geobo_lattice_wbuild)."""
    lib = require_gpu()
    done = 0
    while done < rows:
        nb = min(rows - done, 65535)
        _lib.check(lib.geobo_lattice_wbuild(nb, int(Py), int(Px), int(nz), _p(_chk(lamW, "lamW")),
                                            C.c_void_p(_chk(lhat, "lhat").data_ptr() + done * Py * Px * 8),
                                            C.c_void_p(_chk(W, "W").data_ptr() + done * Px * nz * Py * 8), _stream()), "geobo_lattice_wbuild")
        done += nb


def lattice_wplanes(rows, Py, Px, nz, lam3, lhat, W):
    """W[r][iz][ky][kx] = lam3[iz][ky][kx] * lhat[r][ky][kx]  (geobo_lattice_wplanes)."""
    lib = require_gpu()
    done = 0
    while done < rows:
        nb = min(rows - done, 65535)
        _lib.check(lib.geobo_lattice_wplanes(nb, int(Py), int(Px), int(nz), _p(_chk(lam3, "lam3")),
                                             C.c_void_p(_chk(lhat, "lhat").data_ptr() + done * Py * Px * 8),
                                             C.c_void_p(_chk(W, "W").data_ptr() + done * nz * Py * Px * 8), _stream()), "geobo_lattice_wplanes")
        done += nb


def xz2d_fold_inv_strided(n, rows, ppr, src, in_row, in_plane, Fx, Fz, out, out_row, out_plane, out_rowstride):
    """Inverse radix-2 transform with strided output rows (geobo_xz2d_fold_inv_strided)."""
    lib = require_gpu()
    _lib.check(lib.geobo_xz2d_fold_inv_strided(int(n), int(rows), int(ppr), _p(_chk(src, "src")), int(in_row), int(in_plane), _p(_chk(Fx, "Fx")),
                                               _p(_chk(Fz, "Fz")), _p(_chk(out, "out")), int(out_row), int(out_plane), int(out_rowstride),
                                               _stream()), "geobo_xz2d_fold_inv_strided")


def xz2d_fold_inv_mul(n, rows, ppr, a, a_plane, b, b_row, Fx, Fz, out, out_row, out_plane, out_rowstride):
    """Inverse radix-2 transform of the planes a[p] * b[r] (elementwise), strided output rows (geobo_xz2d_fold_inv_mul)."""
    lib = require_gpu()
    _lib.check(lib.geobo_xz2d_fold_inv_mul(int(n), int(rows), int(ppr), _p(_chk(a, "a")), int(a_plane), _p(_chk(b, "b")), int(b_row),
                                           _p(_chk(Fx, "Fx")), _p(_chk(Fz, "Fz")), _p(_chk(out, "out")), int(out_row), int(out_plane),
                                           int(out_rowstride), _stream()), "geobo_xz2d_fold_inv_mul")


def colgemv(X, v, out=None, ws=None):
    """out[c] = sum_r X[r, c] v[r]  (X: 2-D row-major CUDA float64, unit column stride, even width)."""
    lib = require_gpu()
    ld = _rowmajor(X, "X")
    m, n = X.shape
    assert _chk(v, "v").numel() >= m
    if out is None:
        out = torch.empty(n, dtype=F64, device=X.device)
    nbytes = lib.geobo_colgemv_ws_bytes(m, n)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty(max(nbytes // 8, 1), dtype=F64, device=X.device)
    _lib.check(lib.geobo_colgemv(m, n, _p(X), ld, _p(v), _p(_chk(out, "out")), _p(ws), ws.numel() * 8, _stream()), "geobo_colgemv")
    return out


def rowgemv(X, v, out=None):
    """out[r] = sum_c X[r, c] * v[c] (geobo_rowgemv): a forward operator applied to a model, data = A rho."""
    lib = require_gpu()
    ld = _rowmajor(X, "X")
    m, n = X.shape
    assert v.numel() >= n and v.dtype == F64 and v.is_cuda
    if out is None:
        out = torch.empty(m, dtype=F64, device=X.device)
    _lib.check(lib.geobo_rowgemv(int(m), int(n), _p(X), int(ld), _p(v), _p(out), _stream()), "geobo_rowgemv")
    return out


def sumsq_accum(a, b, rows, ss):
    """ss[slot][c] += sum over the rows r = slot (mod slots) of (a[r, c] + b[r, c])^2  (geobo_sumsq_accum); b may be None.
    a, b: 2-D row-major views (>= rows x n); ss: (slots, n) view with unit column stride."""
    lib = require_gpu()
    lda = _rowmajor(a, "a")
    n = ss.shape[1]
    assert a.shape[1] >= n and a.shape[0] >= rows and _rowmajor(ss, "ss") >= n
    _lib.check(lib.geobo_sumsq_accum(int(rows), int(n), _p(a), lda, _p(b) if b is not None else None,
                                     _rowmajor(b, "b") if b is not None else 0, ss.shape[0], _p(ss), ss.stride(0), _stream()),
               "geobo_sumsq_accum")
    return ss


def lamdot_z(batch, planes, px, nz, D, lam, out):
    """out[b][o] = sum_z D[b][o][z] * lam[b % planes][o][z]  (geobo_lamdot_z)."""
    lib = require_gpu()
    _lib.check(lib.geobo_lamdot_z(int(batch), int(planes), int(px), int(nz), _p(_chk(D, "D")), _p(_chk(lam, "lam")), _p(_chk(out, "out")),
                                  _stream()), "geobo_lamdot_z")
    return out


def convert(src, dst):
    """dst[r, c] = src[r, c] across fp64 <-> fp32 (2-D views, unit column stride, even widths and leading dimensions)."""
    lib = require_gpu()
    assert src.shape == dst.shape and src.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    assert {src.dtype, dst.dtype} == {F64, F32}, "one side float64, the other float32"
    _lib.check(lib.geobo_convert(1 if dst.dtype == F32 else 0, _p(src), src.stride(0), _p(dst), dst.stride(0), src.shape[0],
                                 src.shape[1], _stream()), "geobo_convert")
    return dst


def round_f32_(x):
    """In place x <- (double)(float)x."""
    lib = require_gpu()
    assert _chk(x, "x").is_contiguous()
    _lib.check(lib.geobo_round_f32(_p(x), x.numel(), _stream()), "geobo_round_f32")
    return x


def k_eval(kid, d2, l1, l2, w=1.0, amp=1.0):
    lib = require_gpu()
    d2 = _chk(d2, "d2").contiguous()
    out = torch.empty_like(d2)
    _lib.check(lib.geobo_k_eval(kid, _p(d2), d2.numel(), float(l1), float(l2), float(w), float(amp), _p(out), _stream()),
               "geobo_k_eval")
    return out


def lattice_plan(loc, xe, ye, ze, nx, ny, nz, device="cuda"):
    """Host-side check for the lattice form of A_sens (geobo_a_sens_lattice): the sensors occupy every column of an nx x ny
    lattice at one height AND every node offset xe[j] - sx, ye[i] - sy (planes 1 .. ny-1) is bit-identical for all pairs with
    the same index difference -- then the lattice kernels reproduce geobo_a_sens exactly.  Returns None otherwise (irregular
    survey, or spacings whose differences round differently from pair to pair: the direct kernel is used)."""
    loc = np.asarray(loc, dtype=np.float64)
    xe, ye, ze = (np.asarray(v, dtype=np.float64) for v in (xe, ye, ze))
    if ny < 3 or nz % 2 or loc.shape != (nx * ny, 3) or np.unique(loc[:, 2]).size != 1:
        return None
    ux, uy = np.unique(loc[:, 0]), np.unique(loc[:, 1])
    if ux.size != nx or uy.size != ny:
        return None
    jx, jy = np.searchsorted(ux, loc[:, 0]), np.searchsorted(uy, loc[:, 1])
    if np.unique(jy * nx + jx).size != nx * ny:
        return None
    X = xe[:, None] - ux[None, :]                 # (nx+1, nx): offset of node j from lattice column t
    Y = ye[1:ny, None] - uy[None, :]              # (ny-1, ny): planes 1 .. ny-1 (planes 0 and ny carry the 1e6 padding)
    # every offset must depend on the index difference only, bit for bit: scatter by difference, then compare the whole matrix
    dix = np.arange(nx + 1)[:, None] - np.arange(nx)[None, :] + nx - 1            # d = j - t        -> 0 .. 2nx-1
    diy = np.arange(1, ny)[:, None] - np.arange(ny)[None, :] + ny - 2             # d = i - t, i>=1  -> 0 .. 2ny-3
    dxv, dyv = np.empty(2 * nx), np.empty(2 * ny - 2)
    dxv[dix.ravel()] = X.ravel()
    dyv[diy.ravel()] = Y.ravel()
    if not ((dxv[dix] == X).all() and (dyv[diy] == Y).all()):
        return None
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(device)
    return dict(dxv=dev(dxv, np.float64), dyv=dev(dyv, np.float64), dzv=dev(ze - loc[0, 2], np.float64),
                jx=dev(jx, np.int32), jy=dev(jy, np.int32), rowmajor=bool((jy * nx + jx == np.arange(nx * ny)).all()))


def a_sens(func, B, loc, nx, ny, nz, xe, ye, ze, scale_mul, scale_div, out, iy0=0, iy1=None, plan=None, rows=None, ws=None,
           col_origin=0):
    """Forward operator rows for the sensors in `loc`; optionally only the voxel slab iy0 <= iy < iy1.
    plan (lattice_plan) + rows (slice of the plan's sensors that `loc` holds): interior slabs by the lattice kernels, the
    two 1e6-padded boundary slabs by the direct kernel.
    col_origin: voxel column that out[:, 0] stands for: 0 (full-width operator) or the first column of a compact buffer that holds
    only a slab; the library checks that the requested columns lie inside a buffer row and applies the offset itself."""
    lib = require_gpu()
    ld = _rowmajor(out, "A")
    col_origin = int(col_origin)
    base = _p(out)
    loc = _chk(loc, "loc").contiguous()
    Bh = (C.c_double * 3)(*[float(b) for b in B])
    iy1 = int(ny if iy1 is None else iy1)

    def direct(a, b):
        _lib.check(lib.geobo_a_sens_slab(FUNC_IDS[func], Bh, _p(loc), loc.shape[0], int(nx), int(ny), int(nz), _p(_chk(xe, "xe")),
                                         _p(_chk(ye, "ye")), _p(_chk(ze, "ze")), float(scale_mul), float(scale_div), int(a),
                                         int(b), base, ld, col_origin, _stream()), "geobo_a_sens_slab")
    if plan is None:
        direct(iy0, iy1)
        return out
    rows = slice(0, loc.shape[0]) if rows is None else rows
    jx, jy = plan["jx"][rows].contiguous(), plan["jy"][rows].contiguous()
    assert jx.numel() == loc.shape[0]
    nbytes = lib.geobo_a_sens_lattice_ws_bytes(int(nx), int(ny), int(nz))
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty(nbytes // 8, dtype=F64, device=out.device)
    _lib.check(lib.geobo_a_sens_lattice(FUNC_IDS[func], Bh, loc.shape[0], int(nx), int(ny), int(nz), _p(plan["dxv"]), _p(plan["dyv"]),
                                        _p(plan["dzv"]), C.c_void_p(jx.data_ptr()), C.c_void_p(jy.data_ptr()), float(scale_mul),
                                        float(scale_div), int(iy0), iy1, base, ld, col_origin, _p(ws), nbytes, _stream()),
               "geobo_a_sens_lattice")
    if iy0 == 0:
        direct(0, 1)
    if iy1 == ny:
        direct(ny - 1, ny)
    return out


def potential(func, B, x, y, z):
    lib = require_gpu()
    x, y, z = (_chk(t, "xyz").contiguous() for t in (x, y, z))
    out = torch.empty_like(x)
    Bh = (C.c_double * 3)(*[float(b) for b in B])
    _lib.check(lib.geobo_potential(FUNC_IDS[func], Bh, _p(x), _p(y), _p(z), x.numel(), _p(out), _stream()), "geobo_potential")
    return out


def ak_fused(kid, A, xyz, col0, ncols, l1, l2, w, amp, out):
    """out[:, :ncols] = A @ K[:, col0:col0+ncols] with K generated on the fly (never stored)."""
    lib = require_gpu()
    lda = _rowmajor(A, "A")
    ldo = _rowmajor(out, "AK")
    x, y, z = (_chk(t, "xyz") for t in xyz)
    Ms_pad, N_pad = A.shape
    assert x.numel() >= N_pad and out.shape[0] >= Ms_pad and out.shape[1] >= ncols
    _lib.check(lib.geobo_ak_fused(kid, _p(A), Ms_pad, N_pad, lda, _p(x), _p(y), _p(z), int(col0), int(ncols), float(l1),
                                  float(l2), float(w), float(amp), _p(out), ldo, _stream()), "geobo_ak_fused")
    return out


def cov_table(kid, nx, ny, nz, sx, sy, sz, l1, l2, w, amp, device="cuda"):
    """Covariance on the grid's difference lattice, z mirrored: (ny*nx*2nz,) tensor, index (diy*nx + dix)*2nz + (dz + nz-1)."""
    lib = require_gpu()
    tab = torch.empty(2 * int(nx) * int(ny) * int(nz), dtype=F64, device=device)
    _lib.check(lib.geobo_cov_table(kid, int(nx), int(ny), int(nz), float(sx), float(sy), float(sz), float(l1), float(l2),
                                   float(w), float(amp), _p(tab), _stream()), "geobo_cov_table")
    return tab


def ak_fused_grid(A, nx, ny, nz, table, col0, ncols, out):
    """out[:, :ncols] = A @ K[:, col0:col0+ncols], K gathered from the lattice table (regular grids, nz >= 16)."""
    lib = require_gpu()
    lda = _rowmajor(A, "A")
    ldo = _rowmajor(out, "AK")
    Ms_pad, N_pad = A.shape
    assert out.shape[0] >= Ms_pad and out.shape[1] >= ncols and _chk(table, "table").numel() >= 2 * nx * ny * nz
    _lib.check(lib.geobo_ak_fused_grid(_p(A), Ms_pad, N_pad, lda, int(nx), int(ny), int(nz), _p(table), int(col0), int(ncols),
                                       _p(out), ldo, _stream()), "geobo_ak_fused_grid")
    return out


def gemm_nt(X, Y, C_, alpha=1.0, beta=0.0, lower_only=False, m_valid=0, small_tiles=False):
    """C = alpha X Y^T + beta C.  m_valid > 0: rows >= m_valid of X are zero padding (not contracted, not stored).
    small_tiles: 128-row workgroup tiles even when 256-row tiles would fit."""
    lib = require_gpu()
    ldx, ldy, ldc = _rowmajor(X, "X"), _rowmajor(Y, "Y"), _rowmajor(C_, "C")
    m, k = X.shape
    n = Y.shape[0]
    assert Y.shape[1] == k and C_.shape[0] >= m and C_.shape[1] >= n
    _lib.check(lib.geobo_gemm_nt(m, n, k, float(alpha), _p(X), ldx, _p(Y), ldy, float(beta), _p(C_), ldc,
                                 (1 if lower_only else 0) | (2 if small_tiles else 0), int(m_valid), _stream()), "geobo_gemm_nt")
    return C_


def gemm_nt_splitk(X, Y, C_, splits, ws, lower_only=False, m_valid=0):
    """C = X Y^T with the contraction split into `splits` concurrent slices (ws: >= splits*m*n doubles)."""
    lib = require_gpu()
    ldx, ldy, ldc = _rowmajor(X, "X"), _rowmajor(Y, "Y"), _rowmajor(C_, "C")
    m, k = X.shape
    n = Y.shape[0]
    assert Y.shape[1] == k and C_.shape[0] >= m and C_.shape[1] >= n and ws.numel() >= splits * m * n
    _lib.check(lib.geobo_gemm_nt_splitk(m, n, k, int(splits), _p(X), ldx, _p(Y), ldy, _p(C_), ldc, 1 if lower_only else 0,
                                        int(m_valid), _p(ws), ws.numel() * 8, _stream()), "geobo_gemm_nt_splitk")
    return C_


def gemm_nn(X, Y, C_, alpha=1.0, beta=0.0, x_lower=False, y_lower=False):
    """C = alpha X Y + beta C."""
    lib = require_gpu()
    ldx, ldy, ldc = _rowmajor(X, "X"), _rowmajor(Y, "Y"), _rowmajor(C_, "C")
    m, k = X.shape
    n = Y.shape[1]
    assert Y.shape[0] == k and C_.shape[0] >= m and C_.shape[1] >= n
    _lib.check(lib.geobo_gemm_nn(m, n, k, float(alpha), _p(X), ldx, _p(Y), ldy, float(beta), _p(C_), ldc,
                                 1 if x_lower else 0, 1 if y_lower else 0, _stream()), "geobo_gemm_nn")
    return C_


def _check_extents(fn, ops, batch):
    """Operand tiles may overhang the valid rows -- by contract into slack of the same allocation: checked before the launch."""
    for T, lead, stride, rows, cols, what in ops:
        last = (batch - 1) * int(stride) + (int(rows) - 1) * int(lead) + int(cols)
        room = T.untyped_storage().nbytes() // T.element_size() - T.storage_offset()
        if last > room:
            raise ValueError("%s: operand %s (%d x %d, ld %d, batch %d x stride %d) reads %d elements past its allocation"
                             % (fn, what, rows, cols, lead, batch, stride, last - room))


def axis_pass(fold, y_is_kn, inverse, *args):
    """One batched axis pass of the spectral route against the basis G (analysis) or G^T (synthesis): the radix-2 kernel when `fold`
    (the matrix operand has the pair structure and the strides allow it), else the plain batched GEMM on the same operands."""
    even = (4, 5, 7, 8) + ((10, 11) if not (y_is_kn or inverse) else ())     # operand strides; pair stores of the contiguous analysis
    if fold and all(int(args[i]) % 2 == 0 for i in even):
        return gemm_fold(y_is_kn, inverse, *args)
    return gemm_batched(y_is_kn, *args)


def gemm_fold_lamdot(px, nz, k, G, ldg, Y, ldy, strideY, lam, planes, out, batch):
    """Gram x step in one launch (geobo_gemm_fold_lamdot): out[b][o] = sum_z (G Y_b)[o][z] * lam[b % planes][o][z]."""
    lib = require_gpu()
    _check_extents("geobo_gemm_fold_lamdot", ((G, ldg, 0, pad_n(px), k, "G"), (Y, ldy, strideY, k, pad_n(nz), "Y")), batch)
    done = 0
    while done < batch:
        nb = min(batch - done, 65535)
        _lib.check(lib.geobo_gemm_fold_lamdot(int(px), int(nz), int(k), _p(G), int(ldg), C.c_void_p(Y.data_ptr() + done * strideY * 8),
                                              int(ldy), int(strideY), _p(_chk(lam, "lam")), int(planes), int(done),
                                              C.c_void_p(out.data_ptr() + done * px * 8), int(nb), _stream()), "geobo_gemm_fold_lamdot")
        done += nb
    return out


def gemm_fold(y_is_kn, inverse, m, n, k, X, ldx, strideX, Y, ldy, strideY, C_, ldc, strideC, m_valid, n_valid, batch):
    """Radix-2 form of a gemm_batched axis pass against the pair-interleaved basis G / G^T (geobo_gemm_fold): same arguments as the
    gemm_batched call it replaces (alpha = 1, beta = 0), half the multiply-adds."""
    lib = require_gpu()
    _check_extents("geobo_gemm_fold", ((X, ldx, strideX, m, k, "X"), (Y, ldy, strideY, k if y_is_kn else n, n if y_is_kn else k, "Y")), batch)
    done = 0
    while done < batch:
        nb = min(batch - done, 65535)
        _lib.check(lib.geobo_gemm_fold(1 if y_is_kn else 0, int(inverse), int(m), int(n), int(k),
                                       C.c_void_p(X.data_ptr() + done * strideX * 8), int(ldx), int(strideX),
                                       C.c_void_p(Y.data_ptr() + done * strideY * 8), int(ldy), int(strideY),
                                       C.c_void_p(C_.data_ptr() + done * strideC * 8), int(ldc), int(strideC), int(m_valid), int(n_valid),
                                       int(nb), _stream()), "geobo_gemm_fold")
        done += nb
    return C_


def gemm_batched(y_is_kn, m, n, k, X, ldx, strideX, Y, ldy, strideY, C_, ldc, strideC, m_valid, n_valid, batch, alpha=1.0,
                 beta=0.0):
    """Raw batched GEMM (see include/geobo_hip.h); X/Y/C are tensors whose data_ptr is the batch-0 origin.
    m, n are COMPUTE extents (multiples of 128: the MFMA tiles run without edge predication on their loads; only the stores look at
    m_valid / n_valid), so operand tiles may overhang the valid rows -- by contract into slack of the same allocation.  That contract
    is checked here (round 5: a per-slot operand of the boundary-slab convolution overhung its tensor by 64 KB at nz = 16, which only
    aborted once an unmapped page happened to follow it)."""
    lib = require_gpu()
    done = 0
    esz = 8
    _check_extents("geobo_gemm_batched", ((X, ldx, strideX, m, k, "X"), (Y, ldy, strideY, k if y_is_kn else n, n if y_is_kn else k, "Y")), batch)
    while done < batch:                      # gridDim.y limit
        nb = min(batch - done, 65535)
        _lib.check(lib.geobo_gemm_batched(1 if y_is_kn else 0, int(m), int(n), int(k), float(alpha),
                                          C.c_void_p(X.data_ptr() + done * strideX * esz), int(ldx), int(strideX),
                                          C.c_void_p(Y.data_ptr() + done * strideY * esz), int(ldy), int(strideY), float(beta),
                                          C.c_void_p(C_.data_ptr() + done * strideC * esz), int(ldc), int(strideC),
                                          int(m_valid), int(n_valid), int(nb), _stream()), "geobo_gemm_batched")
        done += nb
    return C_


def scale_broadcast(a, b, out):
    lib = require_gpu()
    _lib.check(lib.geobo_scale_broadcast(_p(_chk(a, "a")), _p(_chk(b, "b")), a.numel(), b.numel(), _p(out), _stream()),
               "geobo_scale_broadcast")
    return out


def scale_broadcast2(a, b0, b1, out0, out1):
    lib = require_gpu()
    _lib.check(lib.geobo_scale_broadcast2(_p(_chk(a, "a")), _p(_chk(b0, "b0")), _p(_chk(b1, "b1")), a.numel(), b0.numel(),
                                          _p(out0), _p(out1), _stream()), "geobo_scale_broadcast2")


# XZ2D_SHAPES (plan.py): (nx, nz) the fused (x, z) transform kernel is instantiated for


def xz2d(inverse, nx, nz, rows, ppr, src, in_row, in_plane, Mx, Mz, out, out_row, out_plane):
    """Fused two-axis transform of rows*ppr planes (geobo_xz2d): X -> Mx X Mz^T."""
    lib = require_gpu()
    _lib.check(lib.geobo_xz2d(1 if inverse else 0, int(nx), int(nz), int(rows), int(ppr), _p(_chk(src, "src")), int(in_row),
                              int(in_plane), _p(_chk(Mx, "Mx")), int(Mx.stride(0)), _p(_chk(Mz, "Mz")), int(Mz.stride(0)),
                              _p(_chk(out, "out")), int(out_row), int(out_plane), _stream()), "geobo_xz2d")


# XZ2D_FOLD_N (plan.py): extents the radix-2 kernels are instantiated for (both axes equal)


def xz2d_fold(inverse, n, rows, ppr, src, in_row, in_plane, Fx, Fz, out, out_row, out_plane):
    """Radix-2 form of xz2d on the pair-interleaved basis (geobo_xz2d_fold); Fx / Fz: (n, n/2, 2) folded matrices."""
    lib = require_gpu()
    _lib.check(lib.geobo_xz2d_fold(1 if inverse else 0, int(n), int(rows), int(ppr), _p(_chk(src, "src")), int(in_row), int(in_plane),
                                   _p(_chk(Fx, "Fx")), _p(_chk(Fz, "Fz")), _p(_chk(out, "out")), int(out_row), int(out_plane), _stream()),
               "geobo_xz2d_fold")


def xz2d_fold_quad(inverse, n, rows, groups, src, in_row, in_plane, Fx, Fz, out, out_row, out_plane):
    """geobo_xz2d_fold for n = 32 planes, four consecutive planes of a row per kernel plane (geobo_xz2d_fold_quad); Fx / Fz: folded
    matrices of diag(G_32, G_32), shape (64, 32, 2)."""
    lib = require_gpu()
    _lib.check(lib.geobo_xz2d_fold_quad(1 if inverse else 0, int(n), int(rows), int(groups), _p(_chk(src, "src")), int(in_row), int(in_plane),
                                        _p(_chk(Fx, "Fx")), _p(_chk(Fz, "Fz")), _p(_chk(out, "out")), int(out_row), int(out_plane), _stream()),
               "geobo_xz2d_fold_quad")


def xz2d_fold_inv_ss_slots(n, rows, ppr):
    return _lib.load().geobo_xz2d_fold_inv_ss_slots(int(n), int(rows), int(ppr))


def xz2d_fold_inv_ss(n, rows, ppr, src, in_row, in_plane, Fx, Fz, ss, src2=None, in2_row=0, r2_first=0):
    """Inverse radix-2 transform fused with the sum of squares over the rows (geobo_xz2d_fold_inv_ss): ss[slot][y][n*n] +=
    sum_r (inverse(src[r][y] (+ src2[r - r2_first][y] for r >= r2_first)))^2."""
    lib = require_gpu()
    assert _chk(ss, "ss").numel() >= xz2d_fold_inv_ss_slots(n, rows, ppr) * ppr * n * n
    _lib.check(lib.geobo_xz2d_fold_inv_ss(int(n), int(rows), int(ppr), _p(_chk(src, "src")), int(in_row), int(in_plane),
                                          _p(_chk(src2, "src2")) if src2 is not None else None, int(in2_row), int(r2_first),
                                          _p(_chk(Fx, "Fx")), _p(_chk(Fz, "Fz")), _p(ss), _stream()), "geobo_xz2d_fold_inv_ss")


def tile_rows(m, m_valid, tile=256, group=64):
    """Rows of each `tile`-row tile that take part in the contraction when rows >= m_valid are padding: whole `group`-row
    wavefront groups (flop accounting of the m_valid argument of geobo_gemm_nt / geobo_posterior_reduce)."""
    m_valid = m if not m_valid or m_valid > m else m_valid
    out = []
    for r0 in range(0, m, tile):
        v = min(max(m_valid - r0, 0), tile)
        out.append((v + group - 1) // group * group)
    return out


def xz2d_fold_lattice(n, rows, ppr, Q, row_off, q_plane, edge, edge_row, Fx, Fz, out, out_row, out_plane):
    """Forward radix-2 transform of operator rows read as windows of the lattice stencil table (geobo_xz2d_fold_lattice)."""
    lib = require_gpu()
    assert row_off.dtype == torch.int64 and row_off.is_contiguous() and row_off.numel() >= rows
    _lib.check(lib.geobo_xz2d_fold_lattice(int(n), int(rows), int(ppr), _p(_chk(Q, "Q")), row_off.data_ptr(), int(q_plane),
                                           _p(_chk(edge, "edge")), int(edge_row), _p(_chk(Fx, "Fx")), _p(_chk(Fz, "Fz")),
                                           _p(_chk(out, "out")), int(out_row), int(out_plane), _stream()), "geobo_xz2d_fold_lattice")


YMUL_SHAPES = ((128, 64),)      # (m, k) geobo_ymul is instantiated for


def ymul(m, k, C, rows, G, src, in_row, out, out_row, fold=False):
    """out[r] (m x C) = G (m x k) . src[r] (k x C) for every row (geobo_ymul); fold: G is pair-interleaved (row 2b+1 = (-1)^j row 2b) and
    the product runs in radix 2 (geobo_ymul_fold)."""
    lib = require_gpu()
    fn, name = (lib.geobo_ymul_fold, "geobo_ymul_fold") if fold else (lib.geobo_ymul, "geobo_ymul")
    _lib.check(fn(int(m), int(k), int(C), int(rows), _p(_chk(G, "G")), int(G.stride(0)), _p(_chk(src, "src")), int(in_row),
                  _p(_chk(out, "out")), int(out_row), _stream()), name)


def xcorr_reduce(nx, nz, rows, planes, src, in_row, in_plane, Mx, lam, out, out_row, out_plane):
    """x step + eigenvalue scaling + channel sum of the lattice Gram (geobo_xcorr_reduce)."""
    lib = require_gpu()
    _lib.check(lib.geobo_xcorr_reduce(int(nx), int(nz), int(rows), int(planes), _p(_chk(src, "src")), int(in_row), int(in_plane),
                                      _p(_chk(Mx, "Mx")), int(Mx.stride(0)), _p(_chk(lam, "lam")), _p(_chk(out, "out")),
                                      int(out_row), int(out_plane), _stream()), "geobo_xcorr_reduce")


def xcorr_reduce_fold(n, rows, planes, src, in_row, in_plane, F, lam, out, out_row, out_plane):
    """Radix-2 form of xcorr_reduce on the pair-interleaved basis (geobo_xcorr_reduce_fold); F: (n, n/2, 2) folded x matrices."""
    lib = require_gpu()
    _lib.check(lib.geobo_xcorr_reduce_fold(int(n), int(rows), int(planes), _p(_chk(src, "src")), int(in_row), int(in_plane),
                                           _p(_chk(F, "F")), _p(_chk(lam, "lam")), _p(_chk(out, "out")), int(out_row), int(out_plane),
                                           _stream()), "geobo_xcorr_reduce_fold")


def a_sens_lattice_stencil(ws, nx, ny, nz):
    """View of the stencil table Q[(2ny-3)][(2nx-1)][nz] that geobo_a_sens_lattice left in its workspace."""
    np_ = (2 * ny - 2) * (2 * nx) * (nz + 1)
    return ws[np_:np_ + (2 * ny - 3) * (2 * nx - 1) * nz].view(2 * ny - 3, 2 * nx - 1, nz)


# TOEPLITZ_NY (plan.py): y extents geobo_toeplitz_y / _y3 are instantiated for


TOEPLITZ_ADD_NY = (80, 96, 112, 128)      # y extents of the accumulating form (geobo_toeplitz_y3_add)


def toeplitz_y(ny, C, R, src, tabs, outs, y0=0, y1=None, plane=None, accumulate=False):
    """outs[j][r, y - y0, c] = sum_y' tabs[j][|y - y'|, c] * src[r, y', c]  (geobo_toeplitz_y3); 1 to 3 property blocks per sweep.
    plane: stride in doubles between the y-planes of src and outs (default C: dense).  accumulate: outs[j] += (geobo_toeplitz_y3_add,
    ny in TOEPLITZ_ADD_NY)."""
    lib = require_gpu()
    y1 = ny if y1 is None else y1
    n = len(tabs)
    assert 1 <= n <= 3 and len(outs) == n
    import ctypes                    # (the argument C -- modes per plane -- shadows the module alias here)
    tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in (_chk(t, "tab") for t in tabs)])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in (_chk(o, "out") for o in outs)])
    fn, name = (lib.geobo_toeplitz_y3_add, "geobo_toeplitz_y3_add") if accumulate else (lib.geobo_toeplitz_y3, "geobo_toeplitz_y3")
    _lib.check(fn(int(ny), int(C), int(C if plane is None else plane), int(R), n, _p(_chk(src, "src")), tp, op, int(y0), int(y1), _stream()), name)


TOEPLITZ_Y2T_NY = (32, 48, 64)      # y extents of the two-term kernel


def toeplitz_y2t(ny, C, R, src_g, src_m, tabs_g, tabs_m, outs, plane=None):
    """outs[j][r, y, c] = sum_y' tabs_g[j][|y - y'|, c] src_g[r, y', c] + tabs_m[j][|y - y'|, c] src_m[r, y', c],  j = 0, 1
    (geobo_toeplitz_y2t: the two-term rows of the transposed posterior in one pass)."""
    lib = require_gpu()
    assert len(tabs_g) == 2 and len(tabs_m) == 2 and len(outs) == 2
    _lib.check(lib.geobo_toeplitz_y2t(int(ny), int(C), int(C if plane is None else plane), int(R), _p(_chk(src_g, "src_g")), _p(_chk(src_m, "src_m")),
                                      _p(_chk(tabs_g[0], "tab")), _p(_chk(tabs_g[1], "tab")), _p(_chk(tabs_m[0], "tab")), _p(_chk(tabs_m[1], "tab")),
                                      _p(_chk(outs[0], "out")), _p(_chk(outs[1], "out")), _stream()), "geobo_toeplitz_y2t")


def toeplitz_y2s(ny, C, R, src_g, src_m, tab_d0, tab_x, tab_d1, outs, plane=None):
    """Two-term rows with a shared cross block (K_10 = K_01) as three products per mode instead of four:
    outs[0] = T(tab_d0) src_g + T(tab_x)(src_g + src_m),  outs[1] = T(tab_d1) src_m + T(tab_x)(src_g + src_m)   (geobo_toeplitz_y2s)."""
    lib = require_gpu()
    assert len(outs) == 2
    _lib.check(lib.geobo_toeplitz_y2s(int(ny), int(C), int(C if plane is None else plane), int(R), _p(_chk(src_g, "src_g")), _p(_chk(src_m, "src_m")),
                                      _p(_chk(tab_d0, "tab")), _p(_chk(tab_x, "tab")), _p(_chk(tab_d1, "tab")),
                                      _p(_chk(outs[0], "out")), _p(_chk(outs[1], "out")), _stream()), "geobo_toeplitz_y2s")


# SPECTRAL_Y_NY (plan.py): y extents of the in-kernel spectral y stage (geobo_spectral_y / _y2s)
_spectral_y_basis = {}              # (ny, device index) -> fragment blob (constant of ny: filled once per device)


def spectral_y_basis(ny, device=None):
    """Transform fragments of geobo_spectral_y for one y extent (a constant of ny: built once per device, kept for the process)."""
    lib = require_gpu()
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    b = _spectral_y_basis.get((ny, dev))
    if b is None:
        n = int(lib.geobo_spectral_y_basis_doubles(int(ny)))
        if n <= 0:
            raise RuntimeError("geobo_spectral_y: ny = %d is not instantiated" % ny)
        b = torch.empty(n, dtype=F64, device="cuda:%d" % dev)
        _lib.check(lib.geobo_spectral_y_basis(int(ny), _p(b), _stream()), "geobo_spectral_y_basis")
        _spectral_y_basis[(ny, dev)] = b
    return b


def spectral_y(ny, C, R, src, tabs, outs, y0=0, y1=None, plane=None, accumulate=False):
    """The sums of toeplitz_y (one to three property blocks; accumulate: outs[j] += ..., ny > 64) through the y axis's own spectrum on the
    matrix pipe (geobo_spectral_y3: ny <= 64 one wave per 16 modes, two blocks per sweep; ny > 64 four waves per 16 modes)."""
    lib = require_gpu()
    y1 = ny if y1 is None else y1
    n = len(tabs)
    assert 1 <= n <= 3 and len(outs) == n
    import ctypes
    tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in (_chk(t, "tab") for t in tabs)])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in (_chk(o, "out") for o in outs)])
    _lib.check(lib.geobo_spectral_y3(int(ny), int(C), int(C if plane is None else plane), int(R), n, _p(_chk(src, "src")), tp, op, int(y0), int(y1),
                                     1 if accumulate else 0, _p(spectral_y_basis(ny, src.device)), _stream()), "geobo_spectral_y3")


SPECTRAL_Y3T_NY = (80, 96, 112, 128)      # y extents of the two-term long-axis form (geobo_spectral_y3t)


def spectral_y3t(ny, C, R, src_g, src_m, tabs_g, tabs_m, outs, plane=None):
    """outs[j] = T(tabs_g[j]) src_g + T(tabs_m[j]) src_m for up to three property blocks, the terms meeting in the y spectrum
    (geobo_spectral_y3t, ny in SPECTRAL_Y3T_NY)."""
    lib = require_gpu()
    n = len(outs)
    assert 1 <= n <= 3 and len(tabs_g) == n and len(tabs_m) == n
    import ctypes
    arr = lambda ts, what: (ctypes.c_void_p * n)(*[t.data_ptr() for t in (_chk(t, what) for t in ts)])
    _lib.check(lib.geobo_spectral_y3t(int(ny), int(C), int(C if plane is None else plane), int(R), n, _p(_chk(src_g, "src_g")), _p(_chk(src_m, "src_m")),
                                      arr(tabs_g, "tab"), arr(tabs_m, "tab"), arr(outs, "out"), _p(spectral_y_basis(ny, src_g.device)), _stream()),
               "geobo_spectral_y3t")


# SPECTRAL_AXIS_N (plan.py): extents of the radix-4 axis passes (geobo_spectral_axis); half-integer basis only


def spectral_axis(inverse, n, C, plane_in, plane_out, item_in, item_out, items, src, dst, mask_ends=False):
    """One axis pass along a strided axis against the half-integer basis G of size n (geobo_spectral_axis): analysis n -> 2n planes or
    synthesis 2n -> n planes of C contiguous modes per item.  mask_ends (analysis): input planes 0 and n - 1 count as zero."""
    lib = require_gpu()
    _lib.check(lib.geobo_spectral_axis(1 if inverse else 0, int(n), int(C), int(plane_in), int(plane_out), int(item_in), int(item_out), int(items),
                                       _p(_chk(src, "src")), _p(_chk(dst, "dst")), _p(spectral_y_basis(n, src.device)), 1 if mask_ends else 0, _stream()),
               "geobo_spectral_axis")
    return dst


def spectral_y2s(ny, C, R, src_g, src_m, tab_d0, tab_x, tab_d1, outs, plane=None):
    """The sums of toeplitz_y2s (two-term rows, shared cross block) with the terms meeting in the y spectrum (geobo_spectral_y2s)."""
    lib = require_gpu()
    assert len(outs) == 2
    _lib.check(lib.geobo_spectral_y2s(int(ny), int(C), int(C if plane is None else plane), int(R), _p(_chk(src_g, "src_g")), _p(_chk(src_m, "src_m")),
                                      _p(_chk(tab_d0, "tab")), _p(_chk(tab_x, "tab")), _p(_chk(tab_d1, "tab")),
                                      _p(_chk(outs[0], "out")), _p(_chk(outs[1], "out")), _p(spectral_y_basis(ny, src_g.device)), _stream()),
               "geobo_spectral_y2s")


class PotrfContext:
    """Fork streams / events of geobo_potrf_inv on the device that is current at construction (owned by the caller: one per
    engine; never shared between concurrent factorisations)."""

    def __init__(self):
        lib = require_gpu()
        self._h = C.c_void_p()
        _lib.check(lib.geobo_potrf_ctx_create(C.byref(self._h)), "geobo_potrf_ctx_create")

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().geobo_potrf_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def potrf_inv(A, Linv=None, ws=None, ctx=None):
    """In-place lower Cholesky of A (m x m, m % 128 == 0); returns (Linv, info_tensor).  Linv / ws may be caller-owned;
    ctx: PotrfContext (concurrent L^-1 subtrees) or None (serial on the current stream)."""
    lib = require_gpu()
    ld = _rowmajor(A, "A")
    m = A.shape[0]
    if Linv is None:
        Linv = torch.empty((m, m), dtype=F64, device=A.device)
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    nbytes = lib.geobo_potrf_ws_bytes(m)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty(max(nbytes // 8, 1), dtype=F64, device=A.device)
    _lib.check(lib.geobo_potrf_inv(m, _p(A), ld, _p(Linv), Linv.stride(0), _p(info), _p(ws), nbytes,
                                   ctx.handle if ctx is not None else None, _stream()), "geobo_potrf_inv")
    return Linv, info


def trmv_stats(Linv, y, L):
    """u = Linv y; stats = [u.u, sum log(L_ii^2)]."""
    lib = require_gpu()
    m = Linv.shape[0]
    u = torch.empty(m, dtype=F64, device=Linv.device)
    stats = torch.empty(2, dtype=F64, device=Linv.device)
    _lib.check(lib.geobo_trmv_stats(m, _p(Linv), _rowmajor(Linv, "Linv"), _p(_chk(y, "y")), _p(L), _rowmajor(L, "L"),
                                    _p(u), _p(stats), _stream()), "geobo_trmv_stats")
    return u, stats


def a_sens_lattice_ws_doubles(nx, ny, nz):
    return max(_lib.load().geobo_a_sens_lattice_ws_bytes(int(nx), int(ny), int(nz)) // 8, 1)


def potrf_ws_doubles(m):
    return max(_lib.load().geobo_potrf_ws_bytes(int(m)) // 8, 1)


def colgemv_ws_doubles(m, n):
    return max(_lib.load().geobo_colgemv_ws_bytes(int(m), int(n)) // 8, 1)


def posterior_ws_doubles(m, ncols):
    return max(_lib.load().geobo_posterior_ws_bytes(int(m), int(ncols)) // 8, 1)


def posterior_reduce(Linv, AK, u, prior_var, ws=None, m_valid=0):
    """mu[c] = sum_m (Linv AK)[m,c] u[m],  var[c] = prior_var - sum_m (Linv AK)[m,c]^2 (V never stored)."""
    lib = require_gpu()
    m, ncols = AK.shape
    mu = torch.empty(ncols, dtype=F64, device=AK.device)
    var = torch.empty(ncols, dtype=F64, device=AK.device)
    nbytes = lib.geobo_posterior_ws_bytes(m, ncols)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty(max(nbytes // 8, 1), dtype=F64, device=AK.device)
    _lib.check(lib.geobo_posterior_reduce(m, ncols, _p(Linv), _rowmajor(Linv, "Linv"), _p(AK), _rowmajor(AK, "AK"),
                                          _p(_chk(u, "u")), float(prior_var), _p(mu), _p(var), int(m_valid), _p(ws), nbytes,
                                          _stream()),
               "geobo_posterior_reduce")
    return mu, var


def mfma_f64_peak(blocks=1024, iters=20000):
    """Time the v_mfma_f64_16x16x4_f64 issue rate; returns TFLOP/s (4 waves/WG x 16 MFMA x 2048 flop per iter)."""
    lib = require_gpu()
    out = torch.empty(blocks * 256, dtype=F64, device="cuda")
    _lib.check(lib.geobo_mfma_f64_peak(blocks, 100, _p(out), _stream()), "geobo_mfma_f64_peak")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.geobo_mfma_f64_peak(blocks, iters, _p(out), _stream()), "geobo_mfma_f64_peak")
    e1.record()
    torch.cuda.synchronize()
    secs = e0.elapsed_time(e1) * 1e-3
    return blocks * 4 * iters * 16 * 2048.0 / secs / 1e12


def to_dev(a, device="cuda"):
    require_gpu()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), dtype=F64).to(device)
