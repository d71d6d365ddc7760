"""Kernel library -- the reference's `geobo/kernels.py` API on the MI355X path.

Same names, argument meaning and return values as the reference (NumPy float64 in, NumPy float64 out);
the arithmetic runs in the hand-written gfx950 kernels of libgeobo_hip.so (k_eval / k_block).  There is
no CPU fallback: without the HIP extension and a GPU these functions raise.

The production path (Inversion.cubing) does NOT go through these materialising functions -- it uses the
matrix-free fused kernel (engine.py) -- they exist so that code written against the reference API
(`create_cov(D2, ...)`, `gpkernel(D2, gamma)` ...) keeps working for small problems.
"""
import numpy as np
import torch

from . import hip
from .engine import create_cov_lengths, weight_matrix


def calcGridPoints3D(Lpix, pixscale):
    """kernels.py:27-42 -- grid points (N,3); row p = (iy*nx + ix)*nz + iz, coordinates (i+1)*scale."""
    Lpix = np.asarray(Lpix)
    pixscale = np.asarray(pixscale)
    xr = np.arange(1, Lpix[0] + 1) * pixscale[0]
    yr = np.arange(1, Lpix[1] + 1) * pixscale[1]
    zr = np.arange(1, Lpix[2] + 1) * pixscale[2]
    # the reference's meshgrid(xr, yr, zr) / ravel / stack / transpose (kernels.py:38-42), written by broadcasting into the
    # same (3, N) buffer: same values, dtype and strides, a fifth of the host time
    arr = np.empty((3, len(yr), len(xr), len(zr)), dtype=np.result_type(xr, yr, zr))
    arr[0] = xr[None, :, None]
    arr[1] = yr[:, None, None]
    arr[2] = zr[None, None, :]
    return arr.reshape(3, -1).T


def _xyz_dev(points):
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("expected an (N,3) array of points")
    return tuple(hip.to_dev(pts[:, d]) for d in range(3))


def calcDistanceMatrix(nDimPoints, distFunc=None):
    """kernels.py:45-61 -- (N,N) matrix of squared distances D2[p,q] = |x_q - x_p|^2.

    3-D points with the default metric go through the HIP kernel (geobo_k_block, family D2).  A caller-supplied `distFunc`
    (the reference hands it the list of per-dimension difference matrices delta[d][p,q] = x_q[d] - x_p[d]) or points of
    another dimension cannot run inside a compiled kernel: the difference matrices are then formed as device tensors and
    the callable is applied to them (any function built from arithmetic / sum works on tensors as on arrays)."""
    pts = np.asarray(nDimPoints, dtype=np.float64)
    if distFunc is not None or pts.ndim != 2 or pts.shape[1] != 3:
        if pts.ndim != 2:
            raise ValueError("expected an (N,dim) array of points")
        dev = hip.to_dev(pts)
        delta = [dev[None, :, d] - dev[:, None, d] for d in range(pts.shape[1])]
        f = distFunc if distFunc is not None else (lambda dl: sum(v ** 2 for v in dl))
        out = f(delta)
        return out.cpu().numpy() if isinstance(out, torch.Tensor) else np.asarray(out)
    xyz = _xyz_dev(pts)
    n = xyz[0].numel()
    out = torch.empty((n, n), dtype=hip.F64, device=xyz[0].device)
    hip.k_block(hip.KERNEL_IDS["d2"], xyz, xyz, 1.0, 1.0, 1.0, 1.0, out)
    return out.cpu().numpy()


def _eval(kid, D2, l1, l2):
    d2 = np.asarray(D2, dtype=np.float64)
    res = hip.k_eval(kid, hip.to_dev(d2.reshape(-1)), l1, l2).cpu().numpy().reshape(d2.shape)
    return res if d2.ndim else float(res)


def gpkernel(D2, gamma):
    """kernels.py:81-88 -- squared exponential exp(-0.5 D2/gamma^2)."""
    return _eval(hip.KERNEL_IDS["exp"], D2, gamma, gamma)


def gpkernel2(D2, gammas):
    """kernels.py:90-99 -- sq-exp x sq-exp cross covariance."""
    return _eval(hip.KERNEL_IDS["exp_x"], D2, gammas[0], gammas[1])


def gpkernel_sparse(D2, gamma):
    """kernels.py:101-114 -- sparse (compact support) kernel of Melkumyan & Ramos."""
    return _eval(hip.KERNEL_IDS["sparse"], D2, gamma, gamma)


def gpkernel_sparse2(D2, gammas):
    """kernels.py:116-138 -- sparse x sparse cross covariance (incl. the l2 += 1e-3*l2 offset at equal lengths)."""
    return _eval(hip.KERNEL_IDS["sparse_x"], D2, gammas[0], gammas[1])


def gpkernel_matern32(D2, gamma):
    """kernels.py:140-146 -- Matern 3/2."""
    return _eval(hip.KERNEL_IDS["matern32"], D2, gamma, gamma)


def gpkernel_matern32_2(D2, gammas):
    """kernels.py:148-156 -- Matern 3/2 x Matern 3/2 cross covariance (singular at equal lengths, like the reference)."""
    return _eval(hip.KERNEL_IDS["matern32_x"], D2, gammas[0], gammas[1])


def create_cov(D2, gplength, crossweights=[1, 1, 1], fkernel='sparse'):
    """kernels.py:158-195 -- (3N,3N) cross-covariance matrix from a squared-distance matrix.

    Reproduces the reference's side effect: `gplength` (if an ndarray) is edited in place so that no two
    lengths are equal ([l,l,l] -> [l, 1.02 l, l])."""
    params = create_cov_lengths(gplength)
    W = weight_matrix(crossweights)
    d2 = np.asarray(D2, dtype=np.float64)
    n0, n1 = d2.shape
    dev = hip.to_dev(d2.reshape(-1))
    out = torch.empty((3 * n0, 3 * n1), dtype=hip.F64, device=dev.device)
    for i in range(3):
        for j in range(3):
            kid = hip.kernel_id(fkernel, i != j)
            blk = hip.k_eval(kid, dev, params[i], params[j], W[i][j], 1.0)
            # reference layout: kcov_i = vstack over j (rows), result = hstack over i (columns); symmetric blocks
            out[j * n0:(j + 1) * n0, i * n1:(i + 1) * n1] = blk.view(n0, n1)
    return out.cpu().numpy()
