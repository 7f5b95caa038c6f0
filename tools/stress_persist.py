"""Randomised stress of the default Cholesky path (persistent / hybrid / paired) against the launch-per-step path without
pairing: random training-set sizes, dimensions, hyper-parameters and right-hand sides, every factor, z and record compared bit
for bit, twice per case (call-unique tags: nothing may leak from one call into the next).
With --noise a second stream runs fp64 matrix products all the while (foreign kernels hold CUs, shift every timing and make
some persistent launches give up: those evaluations are redone on the launch-per-step path -- results must not change).
Usage (GPU box): python tools/stress_persist.py [--noise] [cases] [max_n]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import _lib, gp as agp
lib = _lib.load(); dev = torch.device("cuda:0")
noise = "--noise" in sys.argv
if noise:
    sys.argv.remove("--noise")
    side = torch.cuda.Stream()
    NA = torch.randn((3072, 3072), dtype=torch.float64, device=dev)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
rs = np.random.RandomState(20261003)
bad = 0
fb0 = lib.apgp_potrf_fallbacks()
t0 = time.time()
for c in range(cases):
    n = int(rs.randint(65, max_n)) if rs.rand() < 0.8 else int(64 * rs.randint(2, max_n // 64))
    D = int(rs.randint(1, 9))
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    k = agp.ExpSquaredKernel(np.exp(rs.uniform(0.5, 3.0, size=D)), ndim=D)
    g = agp.GP(kernel=k, fit_mean=True, mean=0.0, white_noise=float(rs.uniform(-14, -6)), fit_white_noise=False)
    g._x = X; g._yerr2 = 0.0
    ks = g._kernel_struct()
    X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
    res = []
    for mode in (17, 0, 0):
        if noise:
            with torch.cuda.stream(side):
                for _ in range(2):
                    NB = NA @ NA
        lib.apgp_potrf_mode(mode)
        K = torch.zeros((n, n), dtype=torch.float64, device=dev); z = torch.empty(n, dtype=torch.float64, device=dev)
        info = torch.empty(1, dtype=torch.int32, device=dev); o5 = torch.empty(5, dtype=torch.float64, device=dev); o = np.empty(5)
        rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), float(rs.normal()) if False else 0.125,
                               K.data_ptr(), z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        res.append((torch.tril(K), z, o.copy(), int(info.item())))
    lib.apgp_potrf_mode(0)
    for r in res[1:]:
        same = r[3] == res[0][3] and (np.array_equal(r[2], res[0][2]) if r[3] == 0 else r[2][4] == res[0][2][4])
        if r[3] == 0:
            same = same and torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
        if not same:
            bad += 1
            print("MISMATCH n=%d D=%d info %d/%d" % (n, D, r[3], res[0][3]), flush=True)
    if (c + 1) % 50 == 0:
        print("%d cases, %d mismatches, fallbacks +%d, %.0f s" % (c + 1, bad, lib.apgp_potrf_fallbacks() - fb0, time.time() - t0), flush=True)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
