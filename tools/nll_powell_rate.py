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
WIDTHS = [0, None]
if "--width" in sys.argv:
    i = sys.argv.index("--width"); WIDTHS = [0] + [int(w) for w in sys.argv[i + 1].split(",")]; del sys.argv[i:i + 2]
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
    lib = _lib.load()
    for ahead in WIDTHS:          # without / with the Powell look-ahead (gpUtils._powellAhead; width by size, or --width a,b,..)
        g.lookahead = ahead
        g._nllMemo = None
        minimize(f, p0, method="powell", options={"maxfev": 200})
        g._nllMemo = None
        torch.cuda.synchronize(); cnt[0] = 0
        sb = lib.apgp_nll_side_batches()
        t0 = time.perf_counter()
        res = minimize(f, p0, method="powell", options={"maxfev": 1500})
        t1 = time.perf_counter()
        print("N = %4d, look-ahead %-4s: %.1f ms for %d evaluations of scipy's Powell = %.3f ms each (side-by-side batches %d; f = %.10g)"
              % (N, "off" if ahead == 0 else "on(%d)" % g.lookahead_width(), (t1 - t0) * 1e3, cnt[0], (t1 - t0) / cnt[0] * 1e3,
                 lib.apgp_nll_side_batches() - sb, res["fun"]), flush=True)
