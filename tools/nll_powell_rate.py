"""Wall time per gpUtils._nll evaluation INSIDE scipy's Powell search (the reference's optimizeGP, gpUtils.py:184-257):
what an evaluation costs when the host does its own work between two of them.
Usage: python tools/nll_powell_rate.py [--lib path] [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from approxposterior_amd import _lib
if "--lib" in sys.argv:
    i = sys.argv.index("--lib"); _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1]); del sys.argv[i:i + 2]
from approxposterior_amd import gpUtils
from scipy.optimize import rosen, minimize
D = 8
for N in [int(a) for a in sys.argv[1:]] or [512, 1152]:
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
    g = gpUtils.defaultGP(X, y)
    p0 = g.get_parameter_vector()
    cnt = [0]
    def f(p):
        cnt[0] += 1
        return gpUtils._nll(p, g, y)
    minimize(f, p0, method="powell", options={"maxfev": 200})
    torch.cuda.synchronize(); cnt[0] = 0
    t0 = time.perf_counter()
    minimize(f, p0, method="powell", options={"maxfev": 1500})
    t1 = time.perf_counter()
    print("N = %4d: %.3f ms per evaluation inside scipy Powell (%d evaluations)" % (N, (t1 - t0) / cnt[0] * 1e3, cnt[0]), flush=True)
