# -*- coding: utf-8 -*-
"""
ctypes binding of ``libapgp.so`` (include/apgp.h): the hand-written HIP kernels
for gfx950.  There is NO fallback: if the shared library is missing or was
built for another ABI the import fails loudly, and every call that returns a
non-zero status raises ``ApgpError`` with the library's message.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libapgp.so")
ABI_VERSION = 8
MAX_DIM = 32

UTIL_AGP, UTIL_BAPE, UTIL_JONES, UTIL_NONE = 0, 1, 2, 3


class ApgpError(RuntimeError):
    pass


class KernelStruct(ctypes.Structure):
    """``apgp_kernel_t`` (include/apgp.h)."""
    _fields_ = [("ndim", ctypes.c_int32), ("lin_order", ctypes.c_int32),
                ("amp", ctypes.c_double), ("diag_add", ctypes.c_double),
                ("inv_metric", ctypes.c_double * MAX_DIM), ("lin_coef", ctypes.c_double)]


class BestStruct(ctypes.Structure):
    """``apgp_best_t``."""
    _fields_ = [("u", ctypes.c_double), ("index", ctypes.c_int64)]


_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I32 = ctypes.c_int32
_F64 = ctypes.c_double
_KP = ctypes.POINTER(KernelStruct)

# name -> (restype, argtypes); every symbol include/apgp.h declares
SIGNATURES = {
    "apgp_abi_version": (ctypes.c_int, []),
    "apgp_last_error": (ctypes.c_char_p, []),
    "apgp_npad": (_I64, [_I64]),
    "apgp_packed_linv_len": (_I64, [_I64]),
    "apgp_acquire_work_len": (_I64, [_I64, _I64]),
    "apgp_packed_train_len": (_I64, [_I64, _I32]),
    "apgp_trtri_work_len": (_I64, [_I64]),
    "apgp_grad_work_len": (_I64, [_I64]),
    "apgp_gram": (ctypes.c_int, [_P, _I64, _KP, _P, _I64, _P]),
    "apgp_kernel_cross": (ctypes.c_int, [_P, _I64, _P, _I64, _KP, _P, _I64, _P]),
    "apgp_potrf": (ctypes.c_int, [_P, _I64, _I64, _P, _F64, _P, _P, _P]),
    "apgp_logdet": (ctypes.c_int, [_P, _I64, _I64, _P, _P]),
    "apgp_fit_summary": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _P, _P]),
    "apgp_nll_eval": (ctypes.c_int, [_P, _I64, _KP, _P, _F64, _P, _P, _P, _P, _P, _P]),
    "apgp_potrf_mode": (ctypes.c_int, [ctypes.c_int]),
    "apgp_potrf_fallbacks": (_I64, []),
    "apgp_potrf_backoff_skips": (_I64, []),
    "apgp_nll_side_batches": (_I64, []),
    "apgp_nll_eval_batch": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "apgp_trsv": (ctypes.c_int, [_P, _I64, _I64, _P, _F64, ctypes.c_int, _P, _P, _P]),
    "apgp_trsv_ex": (ctypes.c_int, [_P, _I64, _I64, _P, _F64, ctypes.c_int, _P, _P, ctypes.c_int, _P]),
    "apgp_trsv_mode": (ctypes.c_int, [ctypes.c_int]),
    "apgp_append_diag": (ctypes.c_int, [_P, _P, _F64, _P, _I64, _P]),
    "apgp_winv_apply_work_len": (_I64, [_I64]),
    "apgp_winv_apply": (ctypes.c_int, [_P, _I64, _I64, _P, _F64, ctypes.c_int, _P, _P, _P, _P]),
    "apgp_trtri_pack": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _P, _P]),
    "apgp_pack_train": (ctypes.c_int, [_P, _P, _I64, _KP, _P, _P]),
    "apgp_acquire": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _I64, _KP, _F64, _I32,
                                    ctypes.POINTER(_F64), ctypes.POINTER(_F64), _P,
                                    _F64, _F64, _P, _P, _P, _P, _P, _P]),
    "apgp_packed_lsolve_len": (_I64, [_I64]),
    "apgp_pack_lsolve": (ctypes.c_int, [_P, _I64, _I64, _P, _P]),
    "apgp_acquire_solve": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _I64, _KP, _F64, _I32,
                                          ctypes.POINTER(_F64), ctypes.POINTER(_F64), _P,
                                          _F64, _F64, _P, _P, _P, _P, _P, _P]),
    "apgp_predict1_work_len": (_I64, [_I64]),
    "apgp_predict1_host": (ctypes.c_int, [_P, _P, _I64, _KP, _F64, _P, _I64, _P, _I64, _P, _P, _P]),
    "apgp_predict_mean": (ctypes.c_int, [_P, _I64, _P, _I64, _KP, _F64, _P, _P]),
    "apgp_predict_mean_host": (ctypes.c_int, [_P, _I64, _P, _I64, _KP, _F64, _P, _P, _P]),
    "apgp_ensemble_sample": (ctypes.c_int, [_P, _I64, _KP, _F64, ctypes.POINTER(_F64), ctypes.POINTER(_F64),
                                            _I32, _I32, _I64, _F64, ctypes.c_uint64, _P, _P, _P, _P, _P, _P]),
    "apgp_ensemble_sample_ex": (ctypes.c_int, [_P, _I64, _KP, _F64, ctypes.POINTER(_F64), ctypes.POINTER(_F64),
                                               _I32, _I32, _I64, _F64, ctypes.c_uint64, _P, _P, _P, _P, _P, ctypes.c_int, _P]),
    "apgp_ensemble_mode": (ctypes.c_int, [ctypes.c_int]),
    "apgp_box_candidates": (ctypes.c_int, [_P, _I64, _I32, ctypes.POINTER(_F64), ctypes.POINTER(_F64), ctypes.c_uint64,
                                           _I64, _P]),
    "apgp_release_scratch": (ctypes.c_int, [_P]),
    "apgp_kinv_solve_work_len": (_I64, [_I64]),
    "apgp_kinv_solve": (ctypes.c_int, [_P, _I64, _I64, _P, _P, _P]),
    "apgp_grad_loglik": (ctypes.c_int, [_P, _P, _P, _I64, _I64, _KP, _P, _P, _P]),
}

_lib = None


def load():
    """Load libapgp.so (once).  torch is imported first so that the HIP runtime
    already mapped by PyTorch-ROCm is the one the library binds to (same
    SONAME), which lets it launch on torch's streams and device buffers."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ApgpError(
            "libapgp.so not found at %s -- build it with "
            "`make -C approxposterior_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback." % LIB_PATH)
    import torch  # noqa: F401  (maps libamdhip64 first)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.apgp_abi_version()
    if v != ABI_VERSION:
        raise ApgpError("libapgp.so ABI %d != expected %d" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().apgp_last_error()
        raise ApgpError("%s failed (%d): %s" % (what, status, msg.decode() if msg else ""))
