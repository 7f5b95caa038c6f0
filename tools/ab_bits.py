#!/usr/bin/env python
"""Bit pattern of the fused sweep's outputs (developer script, GPU box): sha1 of (mu, var, u) over a few
shapes, so that two builds of libapgp.so can be compared for bit-identity inside one gpurun call."""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp
from bench import synthetic_c3

for n, d, m in ((1152, 8, 200000), (1024, 2, 150000), (700, 4, 100000), (2304, 16, 60000), (4096, 8, 70000), (300, 8, 50000)):
    X, y = synthetic_c3(n, d)
    T = np.random.RandomState(1).uniform(-5, 5, size=(m, d))
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
               white_noise=-12, fit_white_noise=False)
    g.variance_mode = "inverse"
    g.compute(X)
    mu, var = g.predict(y, T, return_var=True)
    b = g.acquire(y, torch.from_numpy(T).cuda(), "bape", bounds=[(-5, 5)] * d)
    h = hashlib.sha1(np.ascontiguousarray(mu).tobytes() + np.ascontiguousarray(var).tobytes()).hexdigest()[:16]
    print("N=%d D=%d M=%d sha1(mu|var) %s best %s nan %d" % (n, d, m, h, (int(b[0]), float(b[1])), int(np.isnan(var).sum())))
