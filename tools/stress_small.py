"""Randomised stress of the fused single-workgroup evaluations (n <= 64: nll_small_kernel, 64 < n <= 128: nll_two_kernel, and
their batched launches) against the separate Gram / panel / step / finish launches (mode 1): random n, D, hyper-parameters,
shifts, white noise (also rank-deficient matrices); record, factor, z and LAPACK info compared bit for bit.
Usage (GPU box): python tools/stress_small.py [cases]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import _lib, gp as agp
lib = _lib.load()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rs = np.random.RandomState(123)
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rs.choice([rs.randint(1, 65), rs.randint(65, 129), 64, 65, 128, 127, 66]))
    D = int(rs.choice([1, 2, 3, 5, 8, 13, 16, 24, 32]))
    B = int(rs.choice([1, 1, 2, 3, 5, 9]))
    X = rs.uniform(-5, 5, size=(n, D))
    if rs.rand() < 0.15 and n > 3:
        j = rs.randint(1, n - 1); X[j] = X[j - 1]; X[j + 1] = X[j - 1]
    y = rs.normal(size=n)
    wn = float(rs.choice([-12.0, -6.0, -30.0, -60.0]))
    structs, means = [], []
    for b in range(B):
        g = agp.GP(kernel=float(np.exp(rs.uniform(-2, 3))) * agp.ExpSquaredKernel(np.exp(rs.uniform(-1, 3, size=D)), ndim=D),
                   fit_mean=True, mean=float(rs.normal()), white_noise=wn, fit_white_noise=False)
        g._x = X; g._yerr2 = 0.0
        structs.append(g._kernel_struct()); means.append(float(g.mean.value))
    X_d, y_d = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    res = {}
    for mode in (0, 1):
        lib.apgp_potrf_mode(mode)
        K = torch.zeros((B, n, n), dtype=torch.float64, device="cuda"); z = torch.zeros((B, n), dtype=torch.float64, device="cuda")
        info = torch.zeros(B, dtype=torch.int32, device="cuda"); o5 = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
        o = np.zeros((B, 5))
        if B == 1:
            rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(structs[0]), y_d.data_ptr(), means[0], K.data_ptr(), z.data_ptr(),
                                   info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        else:
            karr = (_lib.KernelStruct * B)(*structs); marr = np.array(means)
            rc = lib.apgp_nll_eval_batch(X_d.data_ptr(), n, B, ctypes.addressof(karr), y_d.data_ptr(), marr.ctypes.data, K.data_ptr(),
                                         z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        ok_rows = [b for b in range(B) if o[b, 4] == 0]
        res[mode] = (o.tobytes(), info.cpu().numpy().tobytes(), torch.tril(K)[ok_rows].cpu().numpy().tobytes(), z[ok_rows].cpu().numpy().tobytes(),
                     o5.cpu().numpy().tobytes())
    lib.apgp_potrf_mode(0)
    if res[0] != res[1] or res[0][0] != res[0][4]:
        bad += 1
        print("MISMATCH case %d: n=%d D=%d B=%d wn=%g" % (c, n, D, B, wn))
    if (c + 1) % 250 == 0:
        print("%d cases, %d mismatches, %.0f s" % (c + 1, bad, time.time() - t0))
print("FAILURES: %d" % bad)
