"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/geobo_hip.h declares.
No compute calls here (CPU-only container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from geobo_amd.build import build
    path = build()
    assert os.path.exists(path)
    return ctypes.CDLL(path)


def declared_functions():
    src = open(os.path.join(ROOT, "include", "geobo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(geobo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libgeobo_hip.so does not export %s" % n


def test_ctypes_table_matches_header():
    from geobo_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()


def test_host_side_helpers(lib):
    lib.geobo_pad_m.restype = ctypes.c_int64
    lib.geobo_pad_m.argtypes = [ctypes.c_int64]
    lib.geobo_pad_n.restype = ctypes.c_int64
    lib.geobo_pad_n.argtypes = [ctypes.c_int64]
    lib.geobo_potrf_ws_bytes.restype = ctypes.c_size_t
    lib.geobo_potrf_ws_bytes.argtypes = [ctypes.c_int64]
    assert lib.geobo_version() == 212
    assert lib.geobo_pad_m(8242) == 8448 and lib.geobo_pad_m(256) == 256 and lib.geobo_pad_m(1) == 256
    assert lib.geobo_pad_n(480) == 512 and lib.geobo_pad_n(262144) == 262144
    # one T buffer per node [lo, mid, hi) of the L^-1 tree, (hi - mid) x (mid - lo) blocks of 128 x 128; split on even block counts
    def tree_blocks(n):
        if n <= 1:
            return 0
        mid = n // 2
        if n > 2 and mid & 1:
            mid += 1
        return (n - mid) * mid + tree_blocks(mid) + tree_blocks(n - mid)
    assert tree_blocks(8) == 16 + 2 * 4 + 4 * 1 and tree_blocks(66) == 2145
    # ... followed by the counters of the persistent tile-DAG kernel: 16 control words, nb row counters, nb x nb tile flags, 2 nb flags of
    # the partial sums parked for the chain walker (round 6)
    for m in (256, 1024, 2048, 8448, 33024):
        nb = m // 128
        assert lib.geobo_potrf_ws_bytes(m) == tree_blocks(nb) * 128 * 128 * 8 + 4 * (16 + 3 * nb + nb * nb)


def test_argument_validation_without_gpu(lib):
    """Entry points validate before touching the device: null pointers / misaligned dims give error codes."""
    from geobo_amd import _lib
    L = _lib.load()
    assert L.geobo_gemm_nt(256, 128, 16, 1.0, None, 16, None, 16, 0.0, None, 128, 0, 0, None) == -1
    assert L.geobo_ak_fused(1, None, 256, 256, 256, None, None, None, 0, 128, 1., 1., 1., 1., None, 128, None) == -1
    assert L.geobo_potrf_inv(100, None, 100, None, 100, None, None, 0, None, None) == -1


@pytest.mark.parametrize("nbi,nbj,tm,tn", [(33, 32, 256, 128), (33, 66, 256, 128), (66, 66, 128, 128), (5, 3, 128, 128), (4, 64, 256, 128),
                                            (17, 9, 256, 128)])
def test_tile_order_visits_every_tile_once(nbi, nbj, tm, tn):
    """The arithmetic item order the GEMM launches decode on the device (no tile lists in device memory any more): every
    valid tile exactly once, for the plain, lower-only and triangular-operand forms."""
    import numpy as np
    from geobo_amd import _lib
    L = _lib.load()
    for lower, xl, yl in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)):
        buf = (ctypes.c_int * (nbi * nbj))()
        n = L.geobo_tile_order(nbi, nbj, tm, tn, lower, xl, yl, buf, nbi * nbj)
        got = [(v >> 16, v & 0xffff) for v in list(buf)[:n]]
        want = {(bi, bj) for bi in range(nbi) for bj in range(nbj) if not lower or bj * tn < (bi + 1) * tm}
        assert n == len(want) and len(set(got)) == n and set(got) == want, (lower, xl, yl)
        if lower or not (xl or yl):
            # 32 consecutive items = at most 4 row tiles (one band): the supertile an XCD's L2 holds
            for g in range(0, n, 32):
                rows = {bi for bi, _ in got[g:g + 32]}
                assert max(rows) - min(rows) <= 7
        if xl:
            assert [bi for bi, _ in got] == sorted((bi for bi, _ in got), reverse=True)   # longest contraction first
        if yl:
            assert [bj for _, bj in got] == sorted(bj for _, bj in got)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "geobo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "geobo_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from geobo_amd import _lib, kernels
    with pytest.raises(_lib.GeoboHipUnavailable):
        kernels.gpkernel([1.0, 2.0], 3.0)
