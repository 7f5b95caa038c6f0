"""Round 6: what does a small batch of mid-size _nll evaluations cost when their persistent launches run side by side
(apgp_nll_eval_batch, 128 < n <= 3200, batch <= 8)?  Prints us per call of GP.nll_batch(P[:B]) for B = 1 .. 8, the single
evaluation beside it, whether the side-by-side path served the call, and checks the values against single calls bit for bit.
Usage (GPU box): python tools/nll_side_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from approxposterior_amd import gp as agp, gpUtils, _lib
from bench import synthetic_c3
lib = _lib.load()
for n in (90, 512, 832, 1152, 1664, 2048, 3072):
    D = 8
    X, y = synthetic_c3(n, D)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    g.compute(X)
    g.lookahead = 0
    p = g.get_parameter_vector()
    rs = np.random.RandomState(n)
    P = np.array([p + 0.05 * rs.randn(len(p)) for _ in range(8)])
    single = np.array([gpUtils._nll(q, g, y, None) for q in P])
    for _ in range(10):
        g._nllMemo = None; gpUtils._nll(P[0], g, y, None)
    R = 200 if n <= 1152 else 60
    t0 = time.perf_counter()
    for i in range(R):
        g._nllMemo = None
        gpUtils._nll(P[i % 8], g, y, None)
    ts = (time.perf_counter() - t0) / R * 1e6
    line = "N=%4d  one _nll %7.1f us |" % (n, ts)
    for B in (1, 2, 3, 4, 5, 6, 7):
        for _ in range(5): g.nll_batch(P[:B], y)
        s0, f0 = lib.apgp_nll_side_batches(), lib.apgp_potrf_fallbacks()
        t0 = time.perf_counter()
        for _ in range(R): got = g.nll_batch(P[:B], y)
        tb = (time.perf_counter() - t0) / R * 1e6
        side = lib.apgp_nll_side_batches() - s0
        ok = np.array_equal(got, single[:B])
        line += " B=%d %7.1f%s%s" % (B, tb, "s" if side == R else ("-" if side == 0 else "?"), "" if ok else " MISMATCH")
        if lib.apgp_potrf_fallbacks() != f0: line += " fb%d" % (lib.apgp_potrf_fallbacks() - f0)
    print(line, flush=True)
    lib.apgp_potrf_mode(0)
