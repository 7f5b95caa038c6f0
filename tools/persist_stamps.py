"""Timeline of the persistent Cholesky's last row workgroup (csrc/potrf_persist.h, -DPP_STAMPS): builds a stamped copy
of the library under tools/tmp/ (the shipped libapgp.so is untouched), runs apgp_nll_eval at the given sizes and
prints, per step, the 100 MHz stamps relative to the step's start.  Usage: python tools/persist_stamps.py [n ...]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "approxposterior_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "tmp", "libapgp_stamps.so")
SG16 = True      # (round 5's step: potrf_persist_sg.h; build with -DPP_SG16=0 and set False for round 4's)


def build(extra=(), out=OUT):
    """the stamped library = the shipped objects with potrf.hip recompiled (-DPP_STAMPS and any experiment macros)"""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = [os.path.join(CSRC, f) for f in ("gram.o", "linalg.o", "sweep.o", "grad.o", "ensemble.o")]
    obj = out.replace(".so", "_potrf.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DPP_STAMPS"] + list(extra) +
                          ["-c", os.path.join(CSRC, "potrf.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        for tag in sys.argv[2:]:          # experiment variants: build EXP_NAME ... -> libapgp_stamps_EXP_NAME.so
            build(["-DPP_" + tag], OUT.replace(".so", "_" + tag + ".so"))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1].startswith("EXP_"):
        OUT = OUT.replace(".so", "_" + sys.argv[1] + ".so")
        del sys.argv[1]
    import torch
    from approxposterior_amd import _lib
    _lib.LIB_PATH = OUT
    lib = _lib.load()
    lib.apgp_debug_read_stamps.restype = ctypes.c_int
    lib.apgp_debug_read_stamps.argtypes = [ctypes.c_void_p]
    from approxposterior_amd import gp as agp
    dev = torch.device("cuda:0")
    names = ["start", "factor", "solve", "recv16", "k15", "flags", "handover", "zappl", "L-out", "k11", "recv1", "k0", "k12", "k13", "k14", "recv15", "helper-done", "tiles-final", "tile-staged", "diag-staged"]
    for n in [int(a) for a in sys.argv[1:]] or [1152]:
        D = 8
        rs = np.random.RandomState(n)
        X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
        k = agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
        g = agp.GP(kernel=k, fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False)
        g._x = X; g._yerr2 = 0.0
        ks = g._kernel_struct()
        X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
        K = torch.zeros((n, n), dtype=torch.float64, device=dev); z = torch.empty(n, dtype=torch.float64, device=dev)
        info = torch.empty(1, dtype=torch.int32, device=dev); o5 = torch.empty(5, dtype=torch.float64, device=dev); o = np.empty(5)
        for _ in range(5):
            lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), 0.0, K.data_ptr(), z.data_ptr(), info.data_ptr(),
                              o5.data_ptr(), o.ctypes.data, None)
        torch.cuda.synchronize()
        st = np.zeros(64 * 24, dtype=np.uint64)
        assert lib.apgp_debug_read_stamps(st.ctypes.data) == 0
        st = st.reshape(64, 24).astype(np.int64)
        nb = (n + 63) // 64
        print("n = %d (%d block columns), fallbacks %d; microseconds after the step's start (last row workgroup)" % (n, nb, lib.apgp_potrf_fallbacks()))
        fs = np.zeros(64 * 8, dtype=np.uint64)
        if hasattr(lib, "apgp_debug_read_fstamps") and SG16:
            lib.apgp_debug_read_fstamps.argtypes = [ctypes.c_void_p]
            lib.apgp_debug_read_gstamps.argtypes = [ctypes.c_void_p]
            gs = np.zeros(64 * 8, dtype=np.uint64)
            assert lib.apgp_debug_read_fstamps(fs.ctypes.data) == 0 and lib.apgp_debug_read_gstamps(gs.ctypes.data) == 0
            fs = fs.reshape(64, 8).astype(np.int64); gs = gs.reshape(64, 8).astype(np.int64)
            print("super-group step, us after the step's start: entry | SG0 done | boundary 1 in | SG1 done | boundary 2 in | SG2 done | boundary 3 in | done")
            for sidx in range(2, min(nb - 1, 9)):
                print("  step %2d  factor: %s" % (sidx, " ".join("%6.2f" % ((fs[sidx, i] - st[sidx, 0]) / 100.0) for i in range(8))))
                print("           solve : %s" % " ".join("%6.2f" % ((gs[sidx, i] - st[sidx, 0]) / 100.0) for i in range(8)))
        elif hasattr(lib, "apgp_debug_read_fstamps"):
            lib.apgp_debug_read_fstamps.argtypes = [ctypes.c_void_p]
            assert lib.apgp_debug_read_fstamps(fs.ctypes.data) == 0
            fs = fs.reshape(64, 8).astype(np.int64)
            print("factorising wavefront, us after ITS entry: at the helper wait | columns back | group 4 / 12 / 15 published;  entry after step start")
            for sidx in range(4, min(nb - 1, 10)):
                print("  step %2d: %s ; %6.2f" % (sidx, " ".join("%6.2f" % ((fs[sidx, i] - fs[sidx, 0]) / 100.0) for i in (1, 2, 3, 4, 5)), (fs[sidx, 0] - st[sidx, 0]) / 100.0))
        if SG16 and hasattr(lib, "apgp_debug_read_estamps"):
            es = np.zeros(4 * 32 * 3, dtype=np.uint64)
            lib.apgp_debug_read_estamps.argtypes = [ctypes.c_void_p]
            assert lib.apgp_debug_read_estamps(es.ctypes.data) == 0
            es = es.reshape(4, 32, 3).astype(np.int64)
            print("matrix wavefronts of the last row workgroup at step 5: event (D = diagonal catch-up, T = tile catch-up, k = k-step) start-end, us after the step's start")
            for mwi in range(4):
                row = []
                for e in range(32):
                    if es[mwi, e, 0] == 0:
                        break
                    kind = int(es[mwi, e, 2])
                    name = "D%d" % (kind // 1000) if kind >= 1000 else ("T%d" % (kind // 100) if kind >= 100 else "k%d" % (kind - 1))
                    row.append("%s %.2f-%.2f" % (name, (es[mwi, e, 0] - st[5, 0]) / 100.0, (es[mwi, e, 1] - st[5, 0]) / 100.0))
                print("  w%d: %s" % (4 + mwi, " | ".join(row)))
        print("the step's producer workgroup (s + 1), us after the last workgroup's step start: its step start | its solve done | last group's granules stored")
        for sidx in range(4, min(nb - 2, 10)):
            print("  step %2d: %s" % (sidx, " ".join("%6.2f" % ((st[sidx, i] - st[sidx, 0]) / 100.0) for i in (20, 21, 22))))
        ws = np.zeros(64 * 8, dtype=np.uint64)
        lib.apgp_debug_read_wstamps.argtypes = [ctypes.c_void_p]
        assert lib.apgp_debug_read_wstamps(ws.ctypes.data) == 0
        ws = ws.reshape(64, 8).astype(np.int64)
        print("arrival at the step's barrier, us after the step's start: wavefronts 0 .. 7 (factor, solve, helper, receiver, matrix x 4) | next step's start")
        for sidx in range(4, min(nb - 2, 10)):
            print("  step %2d: %s | %6.2f" % (sidx, " ".join("%6.2f" % ((ws[sidx, i] - st[sidx, 0]) / 100.0) for i in range(8)), (st[sidx + 1, 0] - st[sidx, 0]) / 100.0))
        gs = np.zeros(64 * 8, dtype=np.uint64)
        if hasattr(lib, "apgp_debug_read_gstamps") and not SG16:
            lib.apgp_debug_read_gstamps.argtypes = [ctypes.c_void_p]
            assert lib.apgp_debug_read_gstamps(gs.ctypes.data) == 0
            gs = gs.reshape(64, 8).astype(np.int64)
            print("group of columns 52 .. 55 of the factorising wavefront, shader cycles: broadcast | 4 pivots | own entries | publish | trailing (2 groups of 4 columns)")
            for sidx in range(4, min(nb - 1, 10)):
                print("  step %2d: %s" % (sidx, " ".join("%6d" % (gs[sidx, i + 1] - gs[sidx, i]) for i in range(5))))
        print("SIMD of wavefronts 0..7 (HW_ID bits 5:4): " + " ".join(str((int(v) >> 4) & 3) for v in st[63, :8]))
        print("step  len   " + " ".join("%8s" % s for s in names[1:]))
        for s in range(nb):
            t0 = st[s, 0]
            nxt = st[s + 1, 0] if s + 1 < nb else t0
            print("%4d %6.2f " % (s, (nxt - t0) * 0.01) + " ".join("%8.2f" % ((st[s, i] - t0) * 0.01) if st[s, i] else "       -" for i in range(1, 20)))
        print("total (first step start -> last factor done): %.1f us" % ((st[nb - 1, 1] - st[0, 0]) * 0.01))
        us = np.zeros(64 * 8, dtype=np.uint64)
        lib.apgp_debug_read_ustamps.restype = ctypes.c_int
        lib.apgp_debug_read_ustamps.argtypes = [ctypes.c_void_p]
        assert lib.apgp_debug_read_ustamps(us.ctypes.data) == 0
        us = us.reshape(64, 8).astype(np.int64)
        print("update workgroup 0, per update step; microseconds after row step s+1's start (last row workgroup's clock)")
        print("ustep  rows-seen acquired  in-LDS  products  stored   done")
        for s_ in range(0, nb - 2):
            t0 = st[s_ + 1, 0]
            print("%4d  " % s_ + " ".join("%8.2f" % ((us[s_, i] - t0) * 0.01) for i in range(6)))
