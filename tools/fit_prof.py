import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from approxposterior_amd import gp as agp, _lib
from scipy.optimize import rosen
N, D = 4096, 8
rs = np.random.RandomState(0)
X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
gp = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
gp.compute(X)
p = gp.get_parameter_vector()
for i in range(5):
    gp.set_parameter_vector(p + 1e-3 * (i % 3)); gp.log_likelihood(y, quiet=True)
torch.cuda.synchronize()
