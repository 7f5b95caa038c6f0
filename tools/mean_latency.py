"""Latency of small mean-only predictions (the host-loop sampler's half-steps: GP.predict(y, T, return_cov=False, return_var=False) with a
few points).  Usage (GPU box): python tools/mean_latency.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from approxposterior_amd import gp as agp
rs = np.random.RandomState(0)
for n, D in ((90, 2), (832, 8)):
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 3.0), ndim=D), fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False)
    g.compute(X)
    for m in (10, 32, 33, 64):
        T = rs.uniform(-5, 5, size=(m, D))
        for _ in range(50): g.predict(y, T, return_cov=False, return_var=False)
        t0 = time.perf_counter()
        for _ in range(2000): g.predict(y, T, return_cov=False, return_var=False)
        print("N=%d m=%d: %.1f us per call" % (n, m, (time.perf_counter() - t0) / 2000 * 1e6))
