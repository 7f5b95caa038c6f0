"""Fit-side workload for rocprofv3 (profiles/<tag>_fit_kernel_stats.csv):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fit -o fit -- python3 tools/fit_prof.py
One compute() + five _nll evaluations at N=4096, D=8 (single calls), then batches of eight
hyper-vectors at N=1152 (apgp_nll_eval_batch) next to the same eight one by one, then the
packed-factor build and a gradient."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from approxposterior_amd import gp as agp, gpUtils
from scipy.optimize import rosen

rs = np.random.RandomState(0)


def make(N, D):
    X = rs.uniform(-5, 5, size=(N, D))
    y = np.array([-rosen(x) / 100 for x in X])
    gp = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=np.median(y),
                white_noise=-12, fit_white_noise=False)
    gp.compute(X)
    return gp, y


gp, y = make(4096, 8)
p = gp.get_parameter_vector()
for i in range(5):
    gp.set_parameter_vector(p + 1e-3 * (i % 3))
    gp.log_likelihood(y, quiet=True)
gp.grad_log_likelihood(y, quiet=True)
gp.predict(y, rs.uniform(-5, 5, size=(64, 8)), return_var=True)      # packed factor + one sweep launch
torch.cuda.synchronize()

gp, y = make(1152, 8)
p = np.array(gp.get_parameter_vector())
P = np.array([p + 1e-2 * rs.randn(len(p)) for _ in range(8)])
gp.nll_batch(P, y)
for rep in range(3):
    t0 = time.time(); b = gp.nll_batch(P, y); tb = time.time() - t0
    t0 = time.time(); s = np.array([gpUtils._nll(q, gp, y, None) for q in P]); ts = time.time() - t0
assert np.array_equal(b, s)
print("N=1152: 8 _nll evaluations batched %.2f ms, one by one %.2f ms" % (1e3 * tb, 1e3 * ts))
torch.cuda.synchronize()
