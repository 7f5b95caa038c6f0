"""gpUtils._nll-style evaluations (set_parameter_vector + log_likelihood) at N = 512 / 1152 / 3072 (persistent Cholesky)
and 4096 (launch per 64-column step), for rocprofv3:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_prof -o fit -- python3 tools/fit_nll_prof.py
profiles/r04_fit_kernel_stats.csv is the *_kernel_stats.csv of that run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from approxposterior_amd import gp as agp
from scipy.optimize import rosen
D = 8
for N in (512, 1152, 3072, 4096):
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
    k = agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
    gp = agp.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(X)
    p = gp.get_parameter_vector()
    for i in range(10):
        gp.set_parameter_vector(p + 1e-3 * (i % 3))
        gp.log_likelihood(y, quiet=True)
    if N <= 1152:
        # round 6: what a Powell look-ahead asks for -- 5 hyper-vectors, their persistent factorisations side by side in ONE
        # launch (gram_batch_kernel, potrf_persist_batch_kernel, potrf_finish_kernel with gridDim.y = 5)
        P = np.array([p + 1e-3 * j for j in range(5)])
        for i in range(10):
            gp.nll_batch(P, y)
    torch.cuda.synchronize()
