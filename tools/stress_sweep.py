"""Repeat-launch stress of the hand-synchronised sweep: seven shapes x 60 launches, every output
bit-identical to the first launch (python tools/stress_sweep.py on the GPU box)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from approxposterior_amd import gp as agp
from scipy.optimize import rosen
rs = np.random.RandomState(1)
bad = 0
t00 = time.time()
for (n, d, m, kind) in [(4096, 8, 70000, "agp"), (1152, 8, 200000, "bape"), (2100, 5, 33000, "jones"), (700, 16, 50000, "agp"),
                        (300, 2, 100000, "bape"), (4096, 8, 3000, "agp"), (5000, 3, 20000, "bape")]:
    X = rs.uniform(-5, 5, size=(n, d)); y = np.array([-rosen(x) / 100 for x in X]) if d > 1 else np.sin(X[:, 0])
    gp = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y), white_noise=-10, fit_white_noise=False)
    gp.compute(X)
    T = rs.uniform(-5, 5, size=(m, d))
    for mode in ("inverse", "solve"):     # both variance forms (the substitution form parks V from the matrix wavefronts)
        gp.variance_mode = mode
        ref = gp.acquire(y, T, kind, bounds=[(-5, 5)] * d, return_all=True)
        reps = 40
        for r in range(reps):
            out = gp.acquire(y, T, kind, bounds=[(-5, 5)] * d, return_all=True)
            ok = out[0] == ref[0] and out[1] == ref[1] and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[2:], ref[2:]))
            if not ok:
                bad += 1
        print("n=%d d=%d m=%d %s %s: %d launches, mismatches so far %d, %.1f s" % (n, d, m, kind, mode, reps, bad, time.time() - t00), flush=True)
# the mailbox path of apgp_nll_eval: 3000 evaluations at alternating hyper-parameters, each against its first value
from approxposterior_amd import gpUtils
for n, d in ((50, 2), (90, 2), (600, 8)):
    X = rs.uniform(-5, 5, size=(n, d)); y = np.array([-rosen(x) / 100 for x in X])
    gp = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y), white_noise=-10, fit_white_noise=False)
    gp.compute(X)
    p = gp.get_parameter_vector()
    want = [gpUtils._nll(p + 1e-3 * k, gp, y, None) for k in range(3)]
    for i in range(3000):
        v = gpUtils._nll(p + 1e-3 * (i % 3), gp, y, None)
        if v != want[i % 3]:
            bad += 1
    print("nll n=%d: 3000 evaluations, mismatches so far %d, %.1f s" % (n, bad, time.time() - t00), flush=True)
print("STRESS", "FAILED" if bad else "OK")
