#!/usr/bin/env python
"""Where one gpUtils._nll evaluation's time goes at README sizes (developer script, GPU box):
the bare C call (apgp_nll_eval: launch + kernel + 40-byte result + synchronisation), the Python
path on top of it (set_parameter_vector + log_likelihood), and a cProfile of the latter."""
import os, sys, time, ctypes, cProfile, pstats, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp, gpUtils, _lib
from bench import synthetic_c3

lib = _lib.load()
for n, d in ((50, 2), (64, 2), (90, 2), (128, 8), (512, 8)):
    X, y = synthetic_c3(n, d)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    g.log_likelihood(y)
    ks = g._kernel_struct(); st = g._stream(torch)
    K = torch.empty((n, n), dtype=torch.float64, device="cuda"); z = torch.empty(n, dtype=torch.float64, device="cuda")
    info = torch.empty(1, dtype=torch.int32, device="cuda"); o5 = torch.empty(5, dtype=torch.float64, device="cuda")
    yd = torch.from_numpy(y).cuda(); o = np.empty(5)
    args = (g._x_d.data_ptr(), n, ctypes.byref(ks), yd.data_ptr(), float(g.mean.value), K.data_ptr(), z.data_ptr(),
            info.data_ptr(), o5.data_ptr(), o.ctypes.data, st)
    for _ in range(20): lib.apgp_nll_eval(*args)
    R = 300
    t0 = time.perf_counter()
    for _ in range(R): lib.apgp_nll_eval(*args)
    t_c = (time.perf_counter() - t0) / R
    p = g.get_parameter_vector()
    def nll(i):
        g.set_parameter_vector(p + 1e-3 * (i % 3))
        return g.log_likelihood(y, quiet=True)
    for i in range(20): nll(i)
    t0 = time.perf_counter()
    for i in range(R): nll(i)
    t_py = (time.perf_counter() - t0) / R
    t0 = time.perf_counter()
    for i in range(R): gpUtils._nll(p + 1e-3 * (i % 3), g, y, None)
    t_nll = (time.perf_counter() - t0) / R
    print("N=%4d D=%d: C call apgp_nll_eval %.1f us | set_parameter_vector + log_likelihood %.1f us | gpUtils._nll %.1f us"
          % (n, d, t_c * 1e6, t_py * 1e6, t_nll * 1e6), flush=True)
    if n == 50:
        pr = cProfile.Profile(); pr.enable()
        for i in range(2000): nll(i)
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
        print(s.getvalue())
