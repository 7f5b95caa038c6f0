#!/usr/bin/env python
"""Latency of the reference-faithful scalar path (developer script, GPU box): one candidate per call through
gp.predict(return_var=True) and through utility.AGPUtility -- what utility.minimizeObjective's Nelder-Mead
evaluates 372-429 times per restart (utility.py:131,336-372) -- for both variance forms."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp, utility as ut
from bench import synthetic_c3

for n, d in ((50, 2), (512, 8), (1152, 8), (4096, 8)):
    X, y = synthetic_c3(n, d)
    for mode in ("inverse", "solve"):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
                   white_noise=-12, fit_white_noise=False)
        g.variance_mode = mode
        g.compute(X)
        rs = np.random.RandomState(3)
        T = rs.uniform(-5, 5, size=(400, d))
        prior = lambda t: 0.0
        for i in range(20): g.predict(y, T[i:i + 1], return_var=True)
        t0 = time.perf_counter()
        for i in range(300): mu, var = g.predict(y, T[i:i + 1], return_var=True)
        t_pred = (time.perf_counter() - t0) / 300
        t0 = time.perf_counter()
        for i in range(300): u = ut.AGPUtility(T[i], y, g, prior)
        t_util = (time.perf_counter() - t0) / 300
        m64, v64 = g.predict(y, T[:64], return_var=True)
        one = np.array([g.predict(y, T[i:i + 1], return_var=True) for i in range(64)]).reshape(64, 2)
        print("N=%4d D=%d %-7s: predict(1 candidate, return_var) %.1f us | AGPUtility %.1f us | max |1-by-1 - batch| mu %.1e var %.1e"
              % (n, d, mode, t_pred * 1e6, t_util * 1e6, np.abs(one[:, 0] - m64).max(), np.abs(one[:, 1] - v64).max()), flush=True)
