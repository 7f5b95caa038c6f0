"""The C ABI's host side under AddressSanitizer + UBSan, in the CPU container (never on the GPU box).

    tools/run_asan.sh            # make asan + this script under LD_PRELOAD=libasan:libubsan

Runs (1) the canaries -- a deliberate heap overflow and a deliberate signed overflow inside the sanitized library must
each produce a report, else the run proves nothing; (2) tests/test_cabi_symbols.py against the sanitized library; (3) every
exported entry point once more with null / out-of-range arguments (the checks that run before any HIP call) and the
host-only helpers (size functions, mode switches, counters, error string, scratch release).  Exit code 0 = the canaries
reported and nothing else did."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "approxposterior_amd", "csrc", "asan", "libapgp_asan.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def canary(what):
    """Runs one canary in a child (an ASan report ends the process); returns its stderr."""
    code = "import ctypes; l = ctypes.CDLL(%r); l.apgp_asan_canary(%d)" % (LIB, what)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    return subprocess.run([sys.executable, "-c", code], stderr=subprocess.PIPE, text=True, env=env).stderr


def main():
    if "libasan" not in os.environ.get("LD_PRELOAD", ""):
        sys.exit("run through tools/run_asan.sh (LD_PRELOAD of the sanitizer runtimes)")
    err = canary(1)
    assert "AddressSanitizer: heap-buffer-overflow" in err, "ASan canary did not report:\n" + err
    print("canary 1: AddressSanitizer reports a heap-buffer-overflow inside the library  [as it must]")
    err = canary(2)
    assert "signed integer overflow" in err, "UBSan canary did not report:\n" + err
    print("canary 2: UBSan reports a signed integer overflow inside the library  [as it must]")

    from approxposterior_amd import _lib
    _lib.LIB_PATH = LIB
    import test_cabi_symbols as t
    # (the sanitized library exports one symbol the header does not declare: the canary)
    t.test_abi_version_and_sizes()
    t.test_bad_arguments_are_refused_without_a_gpu()
    lib = _lib.load()
    for name in t.declared_symbols():
        assert hasattr(lib, name), name
    print("tests/test_cabi_symbols.py: passed against %s" % os.path.relpath(LIB, ROOT))

    # every entry point with nothing but nulls / zeros / out-of-range sizes: argument checks only
    ks = _lib.KernelStruct()
    ks.ndim, ks.amp = 2, 1.0
    calls = 0
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        fn = getattr(lib, name)
        for fill in (0, -1, 1 << 40):
            argv = []
            for a in args:
                if a is _lib._KP:
                    argv.append(ctypes.byref(ks))
                elif a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "contents"):
                    argv.append(None)
                elif a is ctypes.c_double:
                    argv.append(float(fill))
                else:
                    argv.append(max(min(fill, 2 ** 31 - 1), -2 ** 31) if a in (ctypes.c_int, ctypes.c_int32) else fill)
            if name in ("apgp_potrf_mode", "apgp_trsv_mode", "apgp_ensemble_mode") and fill > 3:
                argv[0] = 7
            fn(*argv)
            calls += 1
        for ndim in (0, _lib.MAX_DIM + 1, -3):          # kernel structs the checks must refuse
            bad = _lib.KernelStruct()
            bad.ndim = ndim
            if any(a is _lib._KP for a in args):
                buf = (ctypes.c_double * 64)()
                argv = [ctypes.byref(bad) if a is _lib._KP else
                        (ctypes.addressof(buf) if a is ctypes.c_void_p else
                         (ctypes.cast(buf, a) if hasattr(a, "contents") else (0.0 if a is ctypes.c_double else 4)))
                        for a in args]
                rc = fn(*argv)
                assert rc != 0 or res is not ctypes.c_int, (name, ndim, rc)
                calls += 1
    lib.apgp_potrf_mode(0); lib.apgp_trsv_mode(0); lib.apgp_ensemble_mode(0)
    msg = lib.apgp_last_error()
    assert msg is None or isinstance(msg, bytes)
    print("%d argument-check calls over %d entry points: no sanitizer report" % (calls, len(_lib.SIGNATURES)))
    print("OK")


if __name__ == "__main__":
    main()
