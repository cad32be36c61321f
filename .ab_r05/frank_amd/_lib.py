"""ctypes binding of libfrank_hip.so (include/frank_hip.h).  No torch, no fallbacks.

The shared object is built in-tree by __graft_entry__.build() / `make -C frank_amd/csrc`.
If it is missing, importing this module raises; if it is present but no GPU is usable,
every device entry point raises RuntimeError (FH_ERR_HIP) -- there is no CPU path.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRANK_AMD_LIB", os.path.join(_HERE, "libfrank_hip.so"))

FH_OK = 0
FH_ERR_INVALID = -1
FH_ERR_QRANGE = -2
FH_ERR_BAD_P = -3
FH_ERR_NOT_SPD = -4
FH_ERR_NOMEM = -5
FH_ERR_HIP = -6
FH_ERR_UNSUPPORTED = -7
FH_ERR_NUMERIC = -8

VIS_MODELS = {"opt_thick": 0, "opt_thin": 1, "debris": 2}

_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_i64 = ctypes.c_int64


class fh_geometry(ctypes.Structure):
    _fields_ = [("inc_deg", ctypes.c_double), ("PA_deg", ctypes.c_double), ("dRA_arcsec", ctypes.c_double),
                ("dDec_arcsec", ctypes.c_double)]


# name -> (restype, argtypes); every symbol include/frank_hip.h declares
SIGNATURES = {
    "fh_last_error": (ctypes.c_char_p, []),
    "fh_init": (ctypes.c_int, []),
    "fh_last_warning": (ctypes.c_char_p, []),
    "fh_version": (ctypes.c_char_p, []),
    "fh_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "fh_dht_create": (ctypes.c_int, [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    "fh_dht_destroy": (None, [_vp]),
    "fh_dht_size": (ctypes.c_int, [_vp]),
    "fh_dht_get": (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "fh_dht_bucket_tables": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp]),
    "fh_ctx_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_vp)]),
    "fh_ctx_destroy": (None, [_vp]),
    "fh_ctx_synchronize": (ctypes.c_int, [_vp]),
    "fh_ctx_stream": (_vp, [_vp]),
    "fh_dht_coefficients": (ctypes.c_int, [_vp, _dp, _i64, ctypes.c_int, ctypes.c_double, _dp]),
    "fh_predict_visibilities": (ctypes.c_int, [_vp, _dp, _i64, _dp, ctypes.c_double, _dp]),
    "fh_vis_upload": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _i64, _i64, ctypes.POINTER(_vp)]),
    "fh_vis_upload_f32": (ctypes.c_int, [ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _i64, _i64, ctypes.POINTER(_vp)]),
    "fh_vis_upload_c128": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, _dp, _i64, _i64, ctypes.POINTER(_vp)]),
    "fh_vis_destroy": (None, [_vp]),
    "fh_vis_size": (_i64, [_vp]),
    "fh_vis_set_multiplicity": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "fh_vis_residuals": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, _vp, _i64, _i64, _dp, _dp, _dp]),
    "fh_gauss_residuals": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    "fh_predict_sky": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, _dp, _dp, _i64, _dp, _dp, _dp]),
    "fh_vis_residuals_slot": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, _vp, _dp, ctypes.c_int, _dp]),
    "fh_residual_normal_equations": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _dp,
                                                    _dp, _dp]),
    "fh_gauss_normal_equations": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    "fh_bin_reset": (ctypes.c_int, [_vp]),
    "fh_bin_visibilities": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), _vp, _i64, _i64]),
    "fh_bin_last_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "fh_bin_last_prepass_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "fh_fit_last_kernel_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "fh_fit_cluster_info": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_i64)]),
    "fh_ctx_loop_clocks": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_i64)]),
    "fh_ctx_reload_env": (ctypes.c_int, [_vp]),
    "fh_ctx_bucket_tables": (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    "fh_cache_release": (ctypes.c_int, []),
    "fh_stats_upload": (ctypes.c_int, [_vp, _dp, _dp]),
    "fh_sweep_evidence": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_double, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "fh_stats_get_packed": (ctypes.c_int, [_vp, _dp, _i64, _dp]),
    "fh_stats_set_packed": (ctypes.c_int, [_vp, _dp, _i64, _dp]),
    "fh_ctx_set_arithmetic": (ctypes.c_int, [_vp, ctypes.c_int]),
    "fh_ctx_set_reproducible": (ctypes.c_int, [_vp, ctypes.c_int]),
    "fh_ctx_set_range_cache": (ctypes.c_int, [_vp, ctypes.c_int]),
    "fh_ctx_set_cu_partition": (ctypes.c_int, [_vp, ctypes.c_int]),
    "fh_ctx_set_lognormal_linesearch": (ctypes.c_int, [_vp, ctypes.c_int]),
    "fh_stats_device": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_i64), ctypes.POINTER(_vp)]),
    "fh_stats_finalize": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                         _dp, _dp]),
    "fh_map_visibilities": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, ctypes.c_int, _dp, _dp,
                                           _dp, _dp, _dp, _i64, _i64, _dp, _dp, _dp, _dp, _dp]),
    "fh_map_visibilities_c128": (ctypes.c_int, [_vp, ctypes.POINTER(fh_geometry), ctypes.c_int, ctypes.c_int, _dp, _dp,
                                                _dp, _dp, _i64, _i64, _dp, _dp, _dp, _dp, _dp]),
    "fh_gaussian_model": (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    "fh_cho_solve": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int]),
    "fh_svd_solve": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int]),
    "fh_svd_solve_as_reference": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int]),
    "fh_fit_normal": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_int, _dp, _dp, ctypes.POINTER(ctypes.c_int), _dp, _dp]),
    "fh_fit_normal_batched": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double, ctypes.c_int,
                                             _dp, _dp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "fh_fit_slots": (ctypes.c_int, []),
    "fh_fit_flush": (ctypes.c_int, [_vp]),
    "fh_fit_submit": (ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "fh_fit_collect": (ctypes.c_int, [_vp, ctypes.c_int, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    "fh_update_power_spectrum": (ctypes.c_int, [_vp, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                _dp, _dp]),
    "fh_lognormal_model": (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, ctypes.c_double, _dp, _dp, ctypes.POINTER(_i64)]),
    "fh_fit_lognormal": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_double, _dp, _dp,
                                        ctypes.POINTER(ctypes.c_int), _dp, ctypes.POINTER(_i64), _dp, _dp]),
    "fh_fit_lognormal_batched": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double,
                                                ctypes.c_int, ctypes.c_double, _dp, _dp, ctypes.POINTER(ctypes.c_int),
                                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_i64)]),
    "fh_posterior_update": (ctypes.c_int, [_vp, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp]),
    "fh_uvbin_create": (ctypes.c_int, [ctypes.c_int, _dp, _dp, _dp, _dp, _i64, ctypes.c_double, ctypes.POINTER(_vp)]),
    "fh_uvbin_destroy": (None, [_vp]),
    "fh_uvbin_nbins": (ctypes.c_int, [_vp]),
    "fh_uvbin_kernel_ms": (ctypes.c_float, [_vp]),
    "fh_uvbin_get": (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, ctypes.POINTER(_i64), _dp, _dp]),
    "fh_uvbin_determine": (ctypes.c_int, [_vp, _dp, _i64, ctypes.POINTER(ctypes.c_int32)]),
    "fh_uvbin_quantities": (ctypes.c_int, [_vp, _dp, _dp, _dp, _dp, _i64, _dp, _dp, ctypes.POINTER(_i64)]),
    "fh_ctx_set_scale_height": (ctypes.c_int, [_vp, _dp]),
    "fh_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "fh_comm_create": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(_vp)]),
    "fh_comm_destroy": (None, [_vp]),
    "fh_comm_allreduce_stats": (ctypes.c_int, [_vp, _vp]),
    "fh_comm_last_allreduce_ms": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "fh_comm_size": (ctypes.c_int, [_vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "frank_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
        "`make -C frank_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)  # AttributeError here = header / library mismatch
    _f.restype = _res
    _f.argtypes = _args


# The one thing this package changes in the process: GPU_MAX_HW_QUEUES=24 unless the variable is set (include/frank_hip.h:
# fh_init) -- HIP reads it at its first call, and the launches of a pipeline of fits want more than the default four queues.
_queues_preset = "GPU_MAX_HW_QUEUES" in os.environ
lib.fh_init()
# ... which only helps if HIP has not been initialised yet.  The library cannot see that (its check reads the variable it has just
# set); the one common way to get there from Python -- torch imported first and its HIP runtime already up -- is checked here.
_torch = sys.modules.get("torch")
try:
    if _torch is not None and not _queues_preset and _torch.cuda.is_initialized():
        import warnings
        warnings.warn("frank_amd: torch initialised the HIP runtime before frank_amd was imported: GPU_MAX_HW_QUEUES=24 comes too "
                      "late (the runtime keeps its 4 hardware queues and the launches of a pipeline of fits will share them). "
                      "Import frank_amd first or export GPU_MAX_HW_QUEUES=24.", RuntimeWarning, stacklevel=2)
except Exception:
    pass


def last_error():
    return lib.fh_last_error().decode("utf-8", "replace")


def warn_if_any():
    """Turn the library's warning about the last context (too few hardware queues) into a Python RuntimeWarning."""
    msg = lib.fh_last_warning().decode("utf-8", "replace")
    if msg:
        import warnings
        warnings.warn("frank_amd: " + msg, RuntimeWarning, stacklevel=3)


def check(rc, value_error_codes=(FH_ERR_INVALID, FH_ERR_QRANGE, FH_ERR_BAD_P, FH_ERR_NUMERIC)):
    """Map FH_ERR_* to the exception class the reference raises for the same condition."""
    if rc == FH_OK:
        return
    msg = last_error()
    if rc in value_error_codes:
        raise ValueError(msg)
    if rc == FH_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError("frank_amd [%d]: %s" % (rc, msg))


def f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def f4(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a):
    return None if a is None else a.ctypes.data_as(_fp)


def all_float32(u, v, V, weights):
    """True when the whole table is single precision: float32 u, v, weights and complex64 / float32 V."""
    dts = [np.asarray(x).dtype for x in (u, v, weights)]
    vd = np.asarray(V).dtype
    return all(d == np.float32 for d in dts) and vd in (np.dtype(np.complex64), np.dtype(np.float32))


def require_scipy(what):
    """scipy.optimize for the few callers that hand work to SciPy as the reference does (the 'scipy' optimizer of the geometry
    fits, NNLS); the hot path and every default need NumPy + ctypes only."""
    try:
        import scipy.optimize
    except ImportError as e:  # pragma: no cover
        raise ImportError("frank_amd: %s needs SciPy, an optional dependency of this package (everything else runs on NumPy "
                          "+ the HIP library); install scipy or use the default options" % what) from e
    return scipy.optimize


def device_count():
    n = ctypes.c_int(0)
    lib.fh_device_count(ctypes.byref(n))
    return n.value


def make_geometry(geometry):
    """fh_geometry from any object with inc / PA / dRA / dDec attributes (degrees, arcsec)."""
    return fh_geometry(float(geometry.inc), float(geometry.PA), float(geometry.dRA), float(geometry.dDec))


LOGNORMAL_LINESEARCH = ("linear", "reference")


def set_lognormal_linesearch(ctx, mode):
    """'linear': S^-1 (x + lam p) by linearity along a line search (default); 'reference': every trial point multiplied
    out as minimizer.py / statistical_models.py:1075-1085 do (include/frank_hip.h, fh_ctx_set_lognormal_linesearch)."""
    if mode not in LOGNORMAL_LINESEARCH:
        raise ValueError("lognormal_linesearch must be one of %r, not %r" % (LOGNORMAL_LINESEARCH, mode))
    check(lib.fh_ctx_set_lognormal_linesearch(ctx, 1 if mode == "reference" else 0))
