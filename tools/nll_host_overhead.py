"""Wall time of one gpUtils._nll-style evaluation (set_parameter_vector + log_likelihood through approxposterior_amd.gp)
beside the raw C-ABI call apgp_nll_eval on the same inputs: what the host side adds per evaluation of the hyper-parameter
optimisation (gpUtils.py:46-80, 184-257).  Usage: python tools/nll_host_overhead.py [n ...]"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from approxposterior_amd import gp as agp, gpUtils
from scipy.optimize import rosen
D = 8
for N in [int(a) for a in sys.argv[1:]] or [512, 1152]:
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
    g = gpUtils.defaultGP(X, y)
    p = g.get_parameter_vector()
    for i in range(20):
        gpUtils._nll(p + 1e-3 * (i % 3), g, y)
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for i in range(reps):
        gpUtils._nll(p + 1e-3 * (i % 3), g, y)
    t1 = time.perf_counter()
    print("N = %4d: gpUtils._nll %.3f ms per evaluation (wall, %d calls)" % (N, (t1 - t0) / reps * 1e3, reps))
    pr = cProfile.Profile(); pr.enable()
    for i in range(reps):
        gpUtils._nll(p + 1e-3 * (i % 3), g, y)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:2400])
