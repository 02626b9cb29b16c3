"""ctypes binding of libbqhip.so (the C ABI in include/bqhip.h).

There is no arithmetic in this module and no fallback: if the shared library
is missing, fails to load, or no GPU is visible, the product raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BQHIP_LIBRARY: developer override used to A/B experimental builds of the same C ABI
LIB_PATH = os.environ.get("BQHIP_LIBRARY") or os.path.join(_HERE, "libbqhip.so")

BQ_OK, BQ_ERR_NOT_PD, BQ_ERR_BAD_ARG, BQ_ERR_HIP, BQ_ERR_NOMEM = 0, 1, 2, 3, 4
BQ_MAX_DIM = 8
K_CLASSES = ("gram", "potf2", "trsm", "gemm_panel", "syrk_trailing", "syrk_trailing_small",
             "reduce")

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p
_i64 = C.c_int64
_dbl = C.c_double

# name -> (restype, argtypes); every entry point of include/bqhip.h
SIGNATURES = {
    "bq_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bq_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "bq_ctx_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "bq_ctx_destroy": (None, [_vp]),
    "bq_ctx_sync": (C.c_int, [_vp]),
    "bq_last_error": (C.c_char_p, [_vp]),
    "bq_device_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_int)]),
    "bq_set_block": (C.c_int, [_vp, C.c_int]),
    "bq_set_lookahead": (C.c_int, [_vp, C.c_int]),
    "bq_set_lookahead_rows": (C.c_int, [_vp, C.c_int]),
    "bq_get_config": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bq_ctx_stats": (C.c_int, [_vp, _i64p, C.c_int]),
    "bq_ctx_trim": (C.c_int, [_vp]),
    "bq_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "bq_dev_free": (C.c_int, [_vp, _vp]),
    "bq_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "bq_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "bq_memset": (C.c_int, [_vp, _vp, C.c_int, C.c_size_t]),
    "bq_timer_start": (C.c_int, [_vp]),
    "bq_timer_stop_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "bq_profile_enable": (C.c_int, [_vp, C.c_int]),
    "bq_profile_reset": (C.c_int, [_vp]),
    "bq_profile_read": (C.c_int, [_vp, _dp, _i64p, _dp]),
    "bq_profile_timeline": (C.c_int, [_vp, C.c_int, _dp, _i64, _i64p]),
    "bq_cho_factor": (C.c_int, [_vp, _dp, _dp, _i64, _i64p]),
    "bq_cho_solve": (C.c_int, [_vp, _dp, _dp, _dp, _i64, _i64]),
    "bq_logdet": (C.c_int, [_vp, _dp, _i64, _dp]),
    "bq_gram_gauss": (C.c_int, [_vp, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp]),
    "bq_gram_gauss_dev": (C.c_int, [_vp, _vp, _i64, _i64, _dbl, _dp, _dbl, _vp, _i64]),
    "bq_gram_gauss_cross": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _i64, _dbl, _dp, _dp]),
    "bq_potrf_dev": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "bq_gp_fit": (C.c_int, [_vp, _dp, _dp, _i64, _i64, _dbl, _dp, _dbl, C.POINTER(_vp)]),
    "bq_gp_refit": (C.c_int, [_vp, _vp, _dbl, _dp, _dbl]),
    "bq_gp_set_y": (C.c_int, [_vp, _vp, _dp]),
    "bq_gp_refit_predict": (C.c_int, [_vp, _vp, _dbl, _dp, _dbl, _dp, _i64, _dp, _dp]),
    "bq_fit_destroy": (None, [_vp, _vp]),
    "bq_gp_logml": (C.c_int, [_vp, _vp, _dp]),
    "bq_gp_get": (C.c_int, [_vp, _vp, C.c_int, _dp]),
    "bq_gp_predict": (C.c_int, [_vp, _vp, _dp, _i64, _dp, _dp, _dp]),
    "bq_fit_predict": (C.c_int, [_vp, _dp, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp, _i64, _dp, _dp,
                                 _dp]),
    "bq_gp_logml_grid": (C.c_int, [_vp, _dp, _dp, _i64, _i64, _dp, _dp, _dbl, _i64, _dp, _i64]),
    "bq_batch_fit_predict": (C.c_int, [_vp, _i64, _dp, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp, _i64,
                                       _dp, _dp, _dp, _i32p]),
    "bq_int_K": (C.c_int, [_vp, _dp, _i64, _i64, _dbl, _dp, _dp, _dp, _dp]),
    "bq_int_K1_K2": (C.c_int, [_vp, _dp, _i64, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp, _dp, _dp,
                               _dp]),
    "bq_int_int_K1_K2_K1": (C.c_int, [_vp, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp, _dp, _dp, _dp]),
    "bq_int_int_K1_K2": (C.c_int, [_vp, _dp, _i64, _i64, _dbl, _dp, _dbl, _dp, _dp, _dp, _dp]),
    "bq_gp_solve": (C.c_int, [_vp, _vp, _dp, _i64, _dp]),
    "bq_bq_Z_mean": (C.c_int, [_vp, _vp, _dp, _dp, _dp]),
    "bq_bq_Z_var": (C.c_int, [_vp, _vp, _vp, _dp, _dp, _dp]),
    "bq_esm_batch": (C.c_int, [_vp, _dp, _dp, _i64, _i64, _dp, _i64, _dbl, _dbl, _dbl, _dp, _dp, _dp,
                               _dp, _i32p]),
    "bq_esm_border": (C.c_int, [_vp, _vp, _i64, _dp, _i64, _dbl, _dp, _dp, _dp, _dp, _i32p]),
    "bq_plan_create": (C.c_int, [_vp, _i64, _i64, _i64, _i64, C.POINTER(_vp)]),
    "bq_plan_destroy": (None, [_vp, _vp]),
    "bq_plan_set_inputs": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "bq_plan_run": (C.c_int, [_vp, _vp]),
    "bq_plan_results": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _i32p]),
    "bq_plan_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "bq_set_guard": (C.c_int, [C.c_int]),
    "bq_probe_xcd_hop": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_double), _i32p,
                                   C.POINTER(C.c_int64)]),
    "bq_plan_check_guards": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "bq_probe_mfma_f64": (C.c_int, [_vp, _dp]),
    "bq_probe_fma_f64": (C.c_int, [_vp, _dp]),
    "bq_pair_create": (C.c_int, [_vp, _dp, _dp, _dp, _i64, _dp, _i64, _dp, _i64, _i64,
                                 C.POINTER(_vp)]),
    "bq_pair_destroy": (None, [_vp, _vp]),
    "bq_pair_llh": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp, _i32p]),
    "bq_pair_esm": (C.c_int, [_vp, _vp, _dp, _dp, _dbl, _dp, _dp, _dp, _dp, _i32p, _dp, _dp, _dp,
                              _i32p]),
    "bq_probe_hbm": (C.c_int, [_vp, C.c_size_t, _dp, _dp]),
    "bq_probe_hbm_read8": (C.c_int, [_vp, C.c_size_t, _i64, _dp]),
    "bq_probe_gemm": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, _i64, C.c_int, _i64, _dp]),
    "bq_probe_mfma_variant": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "bq_probe_mfma444_layout": (C.c_int, [_vp, C.c_int, C.c_int, _i32p]),
    "bq_probe_rsq": (C.c_int, [_vp, _dp, _i64, _dp]),
    "bq_probe_launch": (C.c_int, [_vp, _i64, _dp]),
    "bq_probe_c2_timeline": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), _i64]),
    "bq_probe_exp": (C.c_int, [_vp, _dp, _i64, _dp]),
    "bq_probe_potf2": (C.c_int, [_vp, _dp, C.c_int, _i64, _dp, _dp, C.POINTER(C.c_int32), _dp,
                                 C.POINTER(C.c_int64)]),
    "bq_probe_mfma_layout": (C.c_int, [_vp, _dp]),
    "bq_probe_panel_solve": (C.c_int, [_vp, _i64, _i64, _i64, _dp, _dp, C.c_int, _i64, _dp]),
}

_lib = None


class LibraryMissing(ImportError):
    pass


def load_library():
    """Load libbqhip.so and declare every signature.  Raises LibraryMissing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "libbqhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bayesian-quadrature_amd/csrc` (there is no CPU fallback)")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # missing HIP runtime etc.
        raise LibraryMissing("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    """double* of a numpy float64 array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(_dp)


def f64(a, order="F"):
    return np.array(a, dtype=np.float64, order=order, copy=False) \
        if isinstance(a, np.ndarray) and a.dtype == np.float64 and \
        (a.flags.f_contiguous if order == "F" else a.flags.c_contiguous) \
        else np.array(a, dtype=np.float64, order=order)
