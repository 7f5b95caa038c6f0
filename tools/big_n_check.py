#!/usr/bin/env python
"""Parity above the benchmark sizes (developer script, GPU box): N = 8192 and N = 6000, both variance forms, against the oracle."""
import os, sys, time
import numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/approxposterior_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import george_oracle as go
from approxposterior_amd import gp as agp
from scipy.optimize import rosen
for n, d, m in ((8192, 8, 30000), (6000, 3, 20000)):
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(n, d)); y = np.array([-rosen(x) / 100 for x in X])
    T = np.random.RandomState(1).uniform(-5, 5, size=(m, d))
    def mk(mod):
        g = mod.GP(kernel=mod.ExpSquaredKernel(np.full(d, 8.0 if d == 8 else 2.0), ndim=d), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
        g.compute(X); return g
    t0 = time.time(); gpo = mk(go); gp = mk(agp)
    mo, vo = gpo.predict(y, T, return_var=True)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    for mode in ("inverse", "solve"):
        gp.variance_mode = mode
        mu, var = gp.predict(y, T, return_var=True)
        bi, bu = gp.acquire(y, T, "agp", bounds=[(-5, 5)] * d)
        uo = -(mo + 0.5 * np.log(2 * np.pi * np.e * vo))
        print("N=%d D=%d M=%d %s: cond_est %.2e  |dmu|/sum|alpha| %.2e  |dvar| %.2e  argmin ok %s  ll rel %.1e  (%.0f s)" % (
            n, d, m, mode, gp.cond_estimate, np.abs(mu - mo).max() / asum, np.abs(var - vo).max(), bi == int(np.nanargmin(uo)),
            abs(gp.log_likelihood(y) - gpo.log_likelihood(y)) / abs(gpo.log_likelihood(y)), time.time() - t0))
