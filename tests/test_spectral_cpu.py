"""Host-side identities behind the spectral route and its radix-2 kernels (geobo_amd/spectral.py) -- CPU only.

The covariance blocks of create_cov (kernels.py:158-195) on the grid of calcGridPoints3D (kernels.py:27-42) are symmetric
Toeplitz per axis; embedded in a symmetric circulant of size 2n they are diagonalised by a real basis.  The basis is stored
pair-interleaved (row 2b = base row, row 2b+1 = (-1)^i times it): what the folded transform kernels rely on."""
import numpy as np
import pytest

from geobo_amd.spectral import base_modes, eigen_matrix, folded_matrices, forward_matrix


@pytest.mark.parametrize("n", [16, 32, 48, 64, 128])
def test_basis_diagonalises_symmetric_toeplitz(n):
    k = np.exp(-0.3 * np.arange(n)) * (1 + 0.1 * np.arange(n))
    G, Em = forward_matrix(n), eigen_matrix(n)
    T = G.T @ np.diag(Em @ k / (2 * n)) @ G
    ref = k[np.abs(np.arange(n)[:, None] - np.arange(n)[None, :])]
    assert np.abs(T - ref).max() < 1e-12
    assert np.abs(G.T @ G / (2 * n) - np.eye(n)).max() < 1e-13          # columns orthogonal, norm^2 = P


@pytest.mark.parametrize("n", [16, 64])
def test_pair_interleaved_layout(n):
    G, Em = forward_matrix(n), eigen_matrix(n)
    alt = 1.0 - 2.0 * (np.arange(n) % 2)
    assert np.array_equal(G[1::2], G[0::2] * alt)                       # row 2b+1 = (-1)^i row 2b, exactly
    modes = base_modes(n)
    assert len(modes) == n and modes[0] == ("cos", 0) and modes[n // 2] == ("mid", n // 2) and modes[-1] == ("sin", n // 2 - 1)
    # the mirror row carries the eigenvalue of the mirrored frequency n - omega; both rows of the middle pair the same one
    d = np.arange(n)
    for b, (kind, om) in enumerate(modes):
        for pos, w in ((2 * b, om), (2 * b + 1, n - om)):
            want = np.cos(2 * np.pi * ((w * d) % (2 * n)) / (2 * n)) * np.where(d == 0, 1.0, 2.0)
            assert np.abs(Em[pos] - want).max() < 1e-14
    assert np.array_equal(Em[n], Em[n + 1])


@pytest.mark.parametrize("n", [16, 64])
def test_folded_forward_and_inverse_equal_the_plain_products(n):
    """out[2b] = E + O, out[2b+1] = E - O and x[2j] = Fe^T (s_even + s_odd), x[2j+1] = Fo^T (s_even - s_odd): what
    geobo_xz2d_fold / geobo_xcorr_reduce_fold compute per axis with half the multiply-adds."""
    G = forward_matrix(n)
    Fe, Fo = folded_matrices(n)
    assert Fe.shape == Fo.shape == (n, n // 2)
    x = np.random.default_rng(1).standard_normal((n, 5))
    E, O = Fe @ x[0::2], Fo @ x[1::2]
    out = np.empty((2 * n, 5))
    out[0::2], out[1::2] = E + O, E - O
    assert np.abs(out - G @ x).max() < 1e-12
    s = np.random.default_rng(2).standard_normal((2 * n, 5))
    xi = np.empty((n, 5))
    xi[0::2], xi[1::2] = Fe.T @ (s[0::2] + s[1::2]), Fo.T @ (s[0::2] - s[1::2])
    assert np.abs(xi - G.T @ s).max() < 1e-12


def test_two_axis_product_reproduces_a_covariance_block():
    """crop[(Gy x Gx)^T diag(Lambda) (Gy x Gx)] applied to a plane = the two-level Toeplitz block applied directly."""
    ny, nx = 16, 32
    rng = np.random.default_rng(3)
    k = np.exp(-0.02 * (np.arange(ny)[:, None] ** 2 * 1.3 + np.arange(nx)[None, :] ** 2))     # k(|dy|, |dx|)
    X = rng.standard_normal((ny, nx))
    iy, ix = np.arange(ny), np.arange(nx)
    K = k[np.abs(iy[:, None, None, None] - iy[None, None, :, None]), np.abs(ix[None, :, None, None] - ix[None, None, None, :])]
    ref = np.einsum("abcd,cd->ab", K, X)
    Gy, Gx, Ey, Ex = forward_matrix(ny), forward_matrix(nx), eigen_matrix(ny), eigen_matrix(nx)
    lam = Ey @ k @ Ex.T / (4 * ny * nx)
    got = Gy.T @ (lam * (Gy @ X @ Gx.T)) @ Gx
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
