"""Randomised stress of the side-by-side batch (apgp_nll_eval_batch, 128 < n <= 3200, 2 <= batch <= 6: ONE launch of
potrf_persist_batch_kernel) against single apgp_nll_eval calls: random sizes, dimensions, batch sizes, hyper-parameters
(incl. non-positive-definite members), every factor, z vector, record and info word compared bit for bit, twice per case
(call-unique tags: nothing may leak from one call -- or one matrix of the batch -- into another).
With --noise a second stream runs fp64 matrix products all the while (foreign kernels hold CUs: some batches give up and
are redone on the batched launch-per-step path -- results must not change).
Usage (GPU box): python tools/stress_batch.py [--noise] [cases] [max_n]"""
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
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 2200
rs = np.random.RandomState(20261004)
bad = 0
fb0, sb0 = lib.apgp_potrf_fallbacks(), lib.apgp_nll_side_batches()
t0 = time.time()
for c in range(cases):
    n = int(rs.randint(129, max_n)) if rs.rand() < 0.8 else int(64 * rs.randint(3, max_n // 64))
    D = int(rs.randint(1, 9))
    B = int(rs.randint(2, 7))
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    karr = (_lib.KernelStruct * B)()
    means = np.empty(B)
    for b in range(B):
        k = agp.ExpSquaredKernel(np.exp(rs.uniform(0.5, 3.0, size=D)), ndim=D)
        g = agp.GP(kernel=k, fit_mean=True, mean=0.0, white_noise=float(rs.uniform(-14, -6)), fit_white_noise=False)
        g._x = X; g._yerr2 = 0.0
        ks = g._kernel_struct()
        if rs.rand() < 0.08:
            ks.diag_add = -float(rs.uniform(0.2, 1.5)) * ks.amp      # not positive definite: info > 0 for this member only
        karr[b] = ks
        means[b] = float(rs.normal())
    X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
    lib.apgp_potrf_mode(0)
    singles = []
    for b in range(B):
        K = torch.zeros((n, n), dtype=torch.float64, device=dev); z = torch.empty(n, dtype=torch.float64, device=dev)
        info = torch.empty(1, dtype=torch.int32, device=dev); o5 = torch.empty(5, dtype=torch.float64, device=dev); o = np.empty(5)
        rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(karr[b]), y_d.data_ptr(), float(means[b]), K.data_ptr(), z.data_ptr(),
                               info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        singles.append((torch.tril(K), z, o.copy(), int(info.item())))
    for rep in range(2):
        if noise:
            with torch.cuda.stream(side):
                for _ in range(2):
                    NB = NA @ NA
        K = torch.zeros((B, n, n), dtype=torch.float64, device=dev); z = torch.empty((B, n), dtype=torch.float64, device=dev)
        info = torch.empty(B, dtype=torch.int32, device=dev); o5 = torch.empty((B, 5), dtype=torch.float64, device=dev)
        o = np.empty((B, 5))
        rc = lib.apgp_nll_eval_batch(X_d.data_ptr(), n, B, ctypes.addressof(karr), y_d.data_ptr(), means.ctypes.data, K.data_ptr(),
                                     z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        infos = info.cpu().numpy()
        for b in range(B):
            s = singles[b]
            same = int(infos[b]) == s[3] and (np.array_equal(o[b], s[2]) if s[3] == 0 else o[b, 4] == s[2][4])
            if s[3] == 0:
                same = same and torch.equal(torch.tril(K[b]), s[0]) and torch.equal(z[b], s[1])
            same = same and o5[b].cpu().numpy().tobytes() == o[b].tobytes()       # (device record == host record; NaNs included)
            if not same:
                bad += 1
                print("MISMATCH case %d n=%d D=%d B=%d member %d info %d/%d" % (c, n, D, B, b, int(infos[b]), s[3]), flush=True)
    if (c + 1) % 25 == 0:
        print("%d cases, %d mismatches, side-by-side batches +%d, fallbacks +%d, %.0f s"
              % (c + 1, bad, lib.apgp_nll_side_batches() - sb0, lib.apgp_potrf_fallbacks() - fb0, time.time() - t0), flush=True)
    lib.apgp_potrf_mode(0)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
