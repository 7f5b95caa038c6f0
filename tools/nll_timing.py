"""_nll latency (single and batches of eight) at N = 50 ... 4096 with checksums, for A/B runs of
the Cholesky (python tools/nll_timing.py on the GPU box)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from approxposterior_amd import gp as agp, gpUtils
from scipy.optimize import rosen
rs = np.random.RandomState(0)
for N in (50, 512, 1152, 4096):
    D = 8 if N > 50 else 2
    X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
    gp = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(X)
    p = np.array(gp.get_parameter_vector())
    vals = []
    for i in range(3): gpUtils._nll(p + 1e-3 * i, gp, y, None)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(20): vals.append(gpUtils._nll(p + 1e-3 * (i % 5), gp, y, None))
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    P = np.array([p + 1e-3 * i for i in range(8)])
    gp.nll_batch(P, y); t0 = time.time(); b = gp.nll_batch(P, y); tb = time.time() - t0
    print("N=%d: _nll %.3f ms, batch of 8 %.3f ms, checksum %.17g %.17g" % (N, 1e3 * dt, 1e3 * tb, sum(vals), b.sum()))
