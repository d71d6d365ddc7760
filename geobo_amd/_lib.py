"""ctypes binding of libgeobo_hip.so (include/geobo_hip.h).

The product path has NO CPU fallback: if the shared object is missing or a symbol cannot be
resolved, importing a compute function raises `GeoboHipUnavailable` with the build recipe.
"""
import ctypes as C
import os

LIB_PATH = os.environ.get("GEOBO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgeobo_hip.so")

_dp = C.c_void_p      # device pointers travel as integers
_i64 = C.c_int64
_f64 = C.c_double
_int = C.c_int
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/geobo_hip.h one to one
SIGNATURES = {
    "geobo_version": (_int, []),
    "geobo_pad_m": (_i64, [_i64]),
    "geobo_pad_n": (_i64, [_i64]),
    "geobo_k_block": (_int, [_int, _dp, _dp, _dp, _i64, _dp, _dp, _dp, _i64, _f64, _f64, _f64, _f64, _dp, _i64, _dp]),
    "geobo_k_block_f32": (_int, [_int, _dp, _dp, _dp, _i64, _dp, _dp, _dp, _i64, _f64, _f64, _f64, _f64, _dp, _i64, _dp]),
    "geobo_k_block_grid": (_int, [_int, _int, _int, _dp, _dp, _i64, _i64, _i64, _i64, _int, _dp, _i64, _dp]),
    "geobo_lattice_wbuild": (_int, [_i64, _int, _int, _int, _dp, _dp, _dp, _dp]),
    "geobo_lattice_wplanes": (_int, [_i64, _int, _int, _int, _dp, _dp, _dp, _dp]),
    "geobo_xz2d_fold_inv_strided": (_int, [_int, _i64, _int, _dp, _i64, _i64, _dp, _dp, _dp, _i64, _i64, _i64, _dp]),
    "geobo_xz2d_fold_inv_mul": (_int, [_int, _i64, _int, _dp, _i64, _dp, _i64, _dp, _dp, _dp, _i64, _i64, _i64, _dp]),
    "geobo_sumsq_accum": (_int, [_i64, _i64, _dp, _i64, _dp, _i64, _int, _dp, _i64, _dp]),
    "geobo_lamdot_z": (_int, [_i64, _int, _int, _int, _dp, _dp, _dp, _dp]),
    "geobo_colgemv_ws_bytes": (_sz, [_i64, _i64]),
    "geobo_colgemv": (_int, [_i64, _i64, _dp, _i64, _dp, _dp, _dp, _sz, _dp]),
    "geobo_rowgemv": (_int, [_i64, _i64, _dp, _i64, _dp, _dp, _dp]),
    "geobo_convert": (_int, [_int, _dp, _i64, _dp, _i64, _i64, _i64, _dp]),
    "geobo_round_f32": (_int, [_dp, _i64, _dp]),
    "geobo_k_eval": (_int, [_int, _dp, _i64, _f64, _f64, _f64, _f64, _dp, _dp]),
    "geobo_a_sens": (_int, [_int, C.POINTER(_f64), _dp, _i64, _int, _int, _int, _dp, _dp, _dp, _f64, _f64, _dp, _i64, _dp]),
    "geobo_a_sens_slab": (_int, [_int, C.POINTER(_f64), _dp, _i64, _int, _int, _int, _dp, _dp, _dp, _f64, _f64, _int, _int, _dp, _i64, _i64, _dp]),
    "geobo_potential": (_int, [_int, C.POINTER(_f64), _dp, _dp, _dp, _i64, _dp, _dp]),
    "geobo_ak_fused": (_int, [_int, _dp, _i64, _i64, _i64, _dp, _dp, _dp, _i64, _i64, _f64, _f64, _f64, _f64, _dp, _i64, _dp]),
    "geobo_cov_table": (_int, [_int, _int, _int, _int, _f64, _f64, _f64, _f64, _f64, _f64, _f64, _dp, _dp]),
    "geobo_ak_fused_grid": (_int, [_dp, _i64, _i64, _i64, _int, _int, _int, _dp, _i64, _i64, _dp, _i64, _dp]),
    "geobo_gemm_nt": (_int, [_i64, _i64, _i64, _f64, _dp, _i64, _dp, _i64, _f64, _dp, _i64, _int, _i64, _dp]),
    "geobo_tile_order": (_i64, [_int, _int, _int, _int, _int, _int, _int, C.POINTER(_int), _i64]),
    "geobo_gemm_nt_splitk": (_int, [_i64, _i64, _i64, _int, _dp, _i64, _dp, _i64, _dp, _i64, _int, _i64, _dp, _sz, _dp]),
    "geobo_gemm_nn": (_int, [_i64, _i64, _i64, _f64, _dp, _i64, _dp, _i64, _f64, _dp, _i64, _int, _int, _dp]),
    "geobo_gemm_batched": (_int, [_int, _i64, _i64, _i64, _f64, _dp, _i64, _i64, _dp, _i64, _i64, _f64, _dp, _i64, _i64, _i64, _i64, _int, _dp]),
    "geobo_gemm_fold_lamdot": (_int, [_i64, _i64, _i64, _dp, _i64, _dp, _i64, _i64, _dp, _int, _i64, _dp, _i64, _dp]),
    "geobo_gemm_fold": (_int, [_int, _int, _i64, _i64, _i64, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _i64, _i64, _i64, _i64, _i64, _dp]),
    "geobo_scale_broadcast": (_int, [_dp, _dp, _i64, _i64, _dp, _dp]),
    "geobo_scale_broadcast2": (_int, [_dp, _dp, _dp, _i64, _i64, _dp, _dp, _dp]),
    "geobo_a_sens_lattice_ws_bytes": (_sz, [_int, _int, _int]),
    "geobo_a_sens_lattice": (_int, [_int, C.POINTER(_f64), _i64, _int, _int, _int, _dp, _dp, _dp, _dp, _dp, _f64, _f64, _int, _int, _dp, _i64,
                                    _i64, _dp, _sz, _dp]),
    "geobo_xz2d": (_int, [_int, _int, _int, _i64, _int, _dp, _i64, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _i64, _dp]),
    "geobo_xz2d_fold": (_int, [_int, _int, _i64, _int, _dp, _i64, _i64, _dp, _dp, _dp, _i64, _i64, _dp]),
    "geobo_xz2d_fold_quad": (_int, [_int, _int, _i64, _int, _dp, _i64, _i64, _dp, _dp, _dp, _i64, _i64, _dp]),
    "geobo_xcorr_reduce_fold": (_int, [_int, _i64, _int, _dp, _i64, _i64, _dp, _dp, _dp, _i64, _i64, _dp]),
    "geobo_xz2d_fold_lattice": (_int, [_int, _i64, _int, _dp, _dp, _i64, _dp, _i64, _dp, _dp, _dp, _i64, _i64, _dp]),
    "geobo_xz2d_fold_inv_ss_slots": (_int, [_int, _i64, _int]),
    "geobo_xz2d_fold_inv_ss": (_int, [_int, _i64, _int, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _dp, _dp, _dp]),
    "geobo_ymul": (_int, [_int, _int, _i64, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _dp]),
    "geobo_ymul_fold": (_int, [_int, _int, _i64, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _dp]),
    "geobo_xcorr_reduce": (_int, [_int, _int, _i64, _int, _dp, _i64, _i64, _dp, _i64, _dp, _dp, _i64, _i64, _dp]),
    "geobo_toeplitz_y": (_int, [_int, _i64, _i64, _i64, _int, _dp, _dp, _dp, _dp, _dp, _int, _int, _dp]),
    "geobo_toeplitz_y2t": (_int, [_int, _i64, _i64, _i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "geobo_toeplitz_y2s": (_int, [_int, _i64, _i64, _i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "geobo_toeplitz_y3": (_int, [_int, _i64, _i64, _i64, _int, _dp, C.POINTER(_dp), C.POINTER(_dp), _int, _int, _dp]),
    "geobo_toeplitz_y3_add": (_int, [_int, _i64, _i64, _i64, _int, _dp, C.POINTER(_dp), C.POINTER(_dp), _int, _int, _dp]),
    "geobo_spectral_y_basis_doubles": (_i64, [_int]),
    "geobo_spectral_y_basis": (_int, [_int, _dp, _dp]),
    "geobo_spectral_y": (_int, [_int, _i64, _i64, _i64, _int, _dp, _dp, _dp, _dp, _dp, _int, _int, _dp, _dp]),
    "geobo_spectral_y2s": (_int, [_int, _i64, _i64, _i64, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "geobo_spectral_y3": (_int, [_int, _i64, _i64, _i64, _int, _dp, C.POINTER(_dp), C.POINTER(_dp), _int, _int, _int, _dp, _dp]),
    "geobo_spectral_y3t": (_int, [_int, _i64, _i64, _i64, _int, _dp, _dp, C.POINTER(_dp), C.POINTER(_dp), C.POINTER(_dp), _dp, _dp]),
    "geobo_spectral_axis": (_int, [_int, _int, _i64, _i64, _i64, _i64, _i64, _i64, _dp, _dp, _dp, _int, _dp]),
    "geobo_potrf_ws_bytes": (_sz, [_i64]),
    "geobo_potrf_ctx_create": (_int, [C.POINTER(C.c_void_p)]),
    "geobo_potrf_ctx_destroy": (_int, [_dp]),
    "geobo_potrf_inv": (_int, [_i64, _dp, _i64, _dp, _i64, _dp, _dp, _sz, _dp, _dp]),
    "geobo_posterior_ws_bytes": (_sz, [_i64, _i64]),
    "geobo_posterior_reduce": (_int, [_i64, _i64, _dp, _i64, _dp, _i64, _dp, _f64, _dp, _dp, _i64, _dp, _sz, _dp]),
    "geobo_trmv_stats": (_int, [_i64, _dp, _i64, _dp, _dp, _i64, _dp, _dp, _dp]),
    "geobo_mfma_f64_peak": (_int, [_int, _int, _dp, _dp]),
    "geobo_mfma_mix": (_int, [_int, _int, _int, _int, _dp, _dp]),
}


class GeoboHipUnavailable(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared object and declare every prototype.  Raises GeoboHipUnavailable loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeoboHipUnavailable(
            "%s not found. Build the HIP extension first: `python -m geobo_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the inversion hot path." % LIB_PATH)
    # torch first, always: its wheel bundles the HIP runtime (libamdhip64.so.7) that owns the device allocations and the
    # stream this library is handed.  Loaded the other way round, this library pulls in /opt/rocm's copy under the same
    # soname, torch then runs on a runtime it was not built against and the first kernel launch here fails.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise GeoboHipUnavailable("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GeoboHipUnavailable("libgeobo_hip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


ERRORS = {-1: "GEOBO_E_ARG", -2: "GEOBO_E_ALIGN", -3: "GEOBO_E_LAUNCH", -4: "GEOBO_E_UNSUPPORTED"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))
