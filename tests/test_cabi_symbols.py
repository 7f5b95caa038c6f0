"""CPU: the C-ABI library loads and exports every symbol include/apgp.h declares
(no compute calls -- there is no GPU here), sizes are consistent, and bad
arguments are refused with a negative status (argument checks run before any
HIP call)."""
import ctypes
import os
import re

from approxposterior_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "apgp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apgp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libapgp.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding missing for %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_sizes():
    lib = _lib.load()
    assert lib.apgp_abi_version() == _lib.ABI_VERSION == 8
    assert lib.apgp_npad(1) == 512 and lib.apgp_npad(512) == 512 and lib.apgp_npad(513) == 1024
    # packed L^-1: row block ib holds (ib+1)*32 tiles of 512 x 16 doubles
    for n, nrb in ((100, 1), (4096, 8), (4097, 9)):
        assert lib.apgp_packed_linv_len(n) == 32 * nrb * (nrb + 1) // 2 * 512 * 16
        assert lib.apgp_packed_lsolve_len(n) == lib.apgp_packed_linv_len(n)
    assert lib.apgp_packed_train_len(4096, 8) == 4096 * 10
    assert lib.apgp_packed_train_len(100, 3) == 512 * 6
    assert lib.apgp_trtri_work_len(100) == 2 * 128 * 128
    assert lib.apgp_grad_work_len(64) == 64 * 64 + 2 + _lib.MAX_DIM
    assert lib.apgp_winv_apply_work_len(4096) == 32 * 4096 and lib.apgp_winv_apply_work_len(129) == 2 * 129
    assert ctypes.sizeof(_lib.KernelStruct) == 8 + 16 + _lib.MAX_DIM * 8 + 8 and _lib.MAX_DIM == 32
    assert ctypes.sizeof(_lib.BestStruct) == 16


def test_bad_arguments_are_refused_without_a_gpu():
    lib = _lib.load()
    ks = _lib.KernelStruct()
    ks.ndim = 2
    ks.amp = 1.0
    assert lib.apgp_gram(None, 4, ctypes.byref(ks), None, 4, None) == -1
    assert b"null pointer" in lib.apgp_last_error()
    assert lib.apgp_logdet(None, 4, 4, None, None) == -1
    assert lib.apgp_trsv(None, 4, 4, None, 0.0, 0, None, None, None) == -1
    assert lib.apgp_trtri_pack(None, 4, 4, None, None, None, None) == -1
    assert lib.apgp_winv_apply(None, 64, 4, None, 0.0, 0, None, None, None, None) == -1
    assert lib.apgp_pack_train(None, None, 4, ctypes.byref(ks), None, None) == -1
    assert lib.apgp_predict_mean(None, 1, None, 4, ctypes.byref(ks), 0.0, None, None) == -1
    assert lib.apgp_grad_loglik(None, None, None, 64, 4, ctypes.byref(ks), None, None, None) == -1
    assert lib.apgp_acquire(None, 1, 0, None, None, 4, ctypes.byref(ks), 0.0, 0, None, None, None,
                            0.01, 0.0, None, None, None, None, None, None) == -1
    assert lib.apgp_acquire_solve(None, 1, 0, None, None, 4, ctypes.byref(ks), 0.0, 0, None, None, None,
                                  0.01, 0.0, None, None, None, None, None, None) == -1
    assert lib.apgp_pack_lsolve(None, 4, 4, None, None) == -1
    assert lib.apgp_predict1_host(None, None, 4, ctypes.byref(ks), 0.0, None, 0, None, 0, None, None, None) == -1
    assert lib.apgp_predict1_work_len(100) == 2 * 512 + 1 + 8
    # round 5 entry points
    lo = (ctypes.c_double * _lib.MAX_DIM)(); hi = (ctypes.c_double * _lib.MAX_DIM)(*([1.0] * _lib.MAX_DIM))
    assert lib.apgp_box_candidates(None, 4, 2, lo, hi, 1, 0, None) == -1
    assert b"null pointer" in lib.apgp_last_error()
    buf = (ctypes.c_double * 8)()
    assert lib.apgp_box_candidates(ctypes.addressof(buf), 4, _lib.MAX_DIM + 1, lo, hi, 1, 0, None) == -1
    assert lib.apgp_box_candidates(ctypes.addressof(buf), 4, 2, lo, hi, 1, -5, None) == -1
    assert lib.apgp_box_candidates(ctypes.addressof(buf), 0, 2, lo, hi, 1, 0, None) == 0        # nothing to draw: no launch
    assert lib.apgp_nll_eval(None, 4, ctypes.byref(ks), None, 0.0, None, None, None, None, None, None) == -1
    assert lib.apgp_ensemble_mode(-1) in (0, 1) and lib.apgp_potrf_backoff_skips() >= 0
    ks.ndim = _lib.MAX_DIM + 1
    assert lib.apgp_gram(ctypes.addressof(buf), 4, ctypes.byref(ks), ctypes.addressof(buf), 4, None) == -1
    ks.ndim = 2
