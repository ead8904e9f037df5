"""ctypes binding of libplp_hip.so (C ABI declared in include/plp.h).

The HIP library is the only compute path of this package: if it cannot be loaded, or no
gfx950 device is visible, every entry point raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLP_LIB overrides the path (A/B runs of kernel variants); it must still be a libplp_hip build
LIB_PATH = os.environ.get("PLP_LIB") or os.path.join(_HERE, "libplp_hip.so")

PLP_OK, PLP_EINVAL, PLP_EUNSUPPORTED, PLP_EHIP, PLP_ENODEVICE, PLP_ENONFINITE = 0, 1, 2, 3, 4, 5
RF_EMPTY, RF_EARLY, RF_MINREP, RF_LPFAIL = 1, 2, 4, 8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/plp.h declares
SIGNATURES = {
    "plp_version": (C.c_int, []),
    "plp_device_count": (C.c_int, []),
    "plp_last_error": (C.c_char_p, []),
    "plp_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "plp_ctx_set_check_finite": (C.c_int, [_vp, C.c_int]),
    "plp_ctx_destroy": (C.c_int, [_vp]),
    "plp_ctx_synchronize": (C.c_int, [_vp, _vp]),
    "plp_lp_solve_batch": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_lp_solve_batch_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_cheby_batch": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_cheby_batch_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_bbox_batch": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_bbox_batch_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "plp_reduce_batch": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "plp_reduce_batch_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "plp_reduce_counters": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "plp_verify_counters": (C.c_int, [_vp, _vp, _vp]),
    "plp_reduce_wide_batch": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "plp_reduce_wide_batch_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "plp_contains": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, C.c_double, C.c_int, _vp]),
    "plp_contains_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, C.c_double, C.c_int, _vp]),
    "plp_assign": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "plp_assign_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "plp_hull_create": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, C.POINTER(_vp)]),
    "plp_hull_destroy": (C.c_int, [_vp]),
    "plp_hull_drop": (C.c_int, [_vp, C.c_int64, _vp]),
    "plp_hull_reassign": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_double, _ip, _vp, _vp, _vp]),
    "plp_hull_read": (C.c_int, [_vp, _vp, _vp]),
    "plp_hull_reassign_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp,
                                        C.c_double, _vp, _vp, _vp]),
    "plp_adjacent_pairs": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_adjacent_pairs_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_overlap_pairs": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_overlap_pairs_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_overlap_cross": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_overlap_cross_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, _vp]),
    "plp_adjacent_pairs_range": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, C.c_int64,
                                           C.c_int64, _vp]),
    "plp_adjacent_pairs_range_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double,
                                               C.c_int64, C.c_int64, _vp]),
    "plp_region_diff_search": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, C.POINTER(_vp)]),
    "plp_rdiff_result_sizes": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "plp_rdiff_result_copy": (C.c_int, [_vp, _vp, _vp, _vp]),
    "plp_rdiff_result_free": (C.c_int, [_vp]),
    "plp_quickhull_run": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp, C.c_double, _vp, C.POINTER(_vp)]),
    "plp_qh_result_sizes": (C.c_int, [_vp, _vp, _vp, _vp]),
    "plp_qh_result_copy": (C.c_int, [_vp, _vp, _vp, _vp]),
    "plp_qh_result_free": (C.c_int, [_vp]),
    "plp_quickhull_last_error": (C.c_char_p, []),
    "plp_selftest": (C.c_int, [_vp, C.c_int, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()
_tls = threading.local()


class PlpError(RuntimeError):
    pass


class UnsupportedSize(ValueError):
    """PLP_EUNSUPPORTED: a size outside what the kernels (or their staging buffers) hold."""


def load():
    """dlopen libplp_hip.so (once).  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PlpError(
                "polytope_amd: %s is missing -- build it with `make -C polytope_amd/csrc -j8` "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                % LIB_PATH)
        try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def device_count():
    return int(load().plp_device_count())


def available():
    """True iff the library loads and a gfx950 device is visible."""
    try:
        return device_count() > 0
    except Exception:
        return False


def check(rc, what):
    if rc != PLP_OK:
        msg = load().plp_last_error().decode("utf-8", "replace")
        if rc == PLP_EUNSUPPORTED:
            raise UnsupportedSize("%s: %s" % (what, msg))
        if rc == PLP_EINVAL:
            raise ValueError("%s: %s" % (what, msg))
        if rc == PLP_ENONFINITE:  # the exception class scipy.optimize.linprog raises on inf/nan input
            raise ValueError("%s: %s" % (what, msg))
        raise PlpError("%s failed (code %d): %s" % (what, rc, msg))


class Context:
    """Owner of one plp_ctx (HIP stream + device scratch arena)."""

    def __init__(self, device=0):
        lib = load()
        h = _vp()
        check(lib.plp_ctx_create(int(device), C.byref(h)), "plp_ctx_create")
        # inf / nan in the inputs of the host-pointer LP batches: found by the library while it stages them (ValueError)
        check(lib.plp_ctx_set_check_finite(h, 1), "plp_ctx_set_check_finite")
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            load().plp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def context(device=None):
    """Per-thread default context on `device` (default: torch's current device, else 0)."""
    if device is None:
        device = 0
        torch = sys.modules.get("torch")  # (a caller that never imported torch never chose a device through it)
        if torch is not None:
            try:
                if torch.cuda.is_available():
                    device = torch.cuda.current_device()
            except Exception:  # pragma: no cover
                pass
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    if device not in cache:
        cache[device] = Context(device)
    return cache[device]
