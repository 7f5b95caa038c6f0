"""Latency of GP.nll_batch (the lock-step restarts of optimizeGP: B hyper-vectors per call) against B single evaluations, at
README sizes.  Usage (GPU box): python tools/nll_batch_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from approxposterior_amd import gp as agp, gpUtils
rs = np.random.RandomState(0)
for n, D in ((50, 2), (90, 2), (128, 2), (256, 2), (512, 8)):
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 3.0), ndim=D), fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False)
    g.compute(X)
    p = g.get_parameter_vector()
    for B in (1, 3, 5):
        P = np.array([p + 0.01 * rs.randn(len(p)) for _ in range(B)])
        for _ in range(20): g.nll_batch(P, y)
        t0 = time.perf_counter()
        for _ in range(500): g.nll_batch(P, y)
        tb = (time.perf_counter() - t0) / 500 * 1e6
        for _ in range(20): gpUtils._nll(P[0], g, y, None)
        t0 = time.perf_counter()
        for _ in range(500):
            g._nllMemo = None
            gpUtils._nll(P[0] + 1e-9 * rs.randn(len(p)), g, y, None)
        ts = (time.perf_counter() - t0) / 500 * 1e6
        print("N=%d B=%d: nll_batch %.1f us, one _nll %.1f us (x B = %.1f)" % (n, B, tb, ts, ts * B))
