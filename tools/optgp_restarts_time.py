import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from approxposterior_amd import gpUtils, likelihood as lh
for n in [int(a) for a in sys.argv[1:]] or (50, 70, 90, 128, 200, 400):
    np.random.seed(57)
    theta = lh.rosenbrockSample(n)
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    res = {}
    for batch in ((True, False, True, False) if n <= 400 else (True, False)):
        gp = gpUtils.defaultGP(theta, y, white_noise=-12)
        t0 = time.perf_counter()
        gp = gpUtils.optimizeGP(gp, theta, y, seed=3, nGPRestarts=3, method="powell", batchRestarts=batch)
        dt = time.perf_counter() - t0
        res.setdefault(batch, []).append(dt)
        p = gp.get_parameter_vector()
    print("N=%d: lock-step %s ms, sequential %s ms" % (n, ["%.1f" % (1e3 * v) for v in res[True]], ["%.1f" % (1e3 * v) for v in res[False]]))
