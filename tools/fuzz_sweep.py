#!/usr/bin/env python
"""Randomised parity fuzz of the sweep (developer script, GPU box): random N (incl. the tile / row-block / kernel
switch boundaries), D, M, utilities, box bounds, masks, NaN candidates and the LinearKernel term -- substitution
form vs inverse form vs the NumPy oracle (mu, sigma^2, utility, arg-min), and the single-candidate path."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import george_oracle as go
from approxposterior_amd import gp as agp
from scipy.optimize import rosen

EPS = 2.2e-16
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
special_n = [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 2304]
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
for case in range(ncase):
    n = int(rs.choice(special_n)) if rs.rand() < 0.6 else int(rs.randint(1, 2600))
    d = int(rs.randint(1, 17))
    m = int(rs.choice([1, 2, 63, 64, 65, 300, 4097, 16500, 20000]))
    if n > 1500:
        m = min(m, 4097)
    kind = str(rs.choice(["agp", "bape", "jones"]))
    lin = rs.rand() < 0.2 and n <= 600
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.array([-rosen(x) / 100 for x in X]) if d > 1 else np.sin(X[:, 0])
    metric = rs.uniform(2.0, 12.0, size=d)
    amp = float(rs.choice([1.0, 3.0]))
    def mk(mod):
        k = mod.ExpSquaredKernel(metric, ndim=d)
        if amp != 1.0:
            k = amp * k
        if lin:
            k = k + 0.5 * mod.LinearKernel(log_gamma2=0.3, order=int(rs_order), ndim=d)
        g = mod.GP(kernel=k, fit_mean=True, mean=float(np.median(y)), white_noise=-10, fit_white_noise=False)
        g.compute(X)
        return g
    rs_order = rs.randint(1, 3)
    try:
        gpo = mk(go)
    except Exception as e:
        print("case %d skipped (oracle: %s)" % (case, type(e).__name__)); continue
    T = rs.uniform(-5.5, 5.5, size=(m, d))
    if m > 3:
        T[1, d - 1] = np.nan
    mask = (rs.rand(m) > 0.1) if rs.rand() < 0.3 else None
    bounds = [(-5, 5)] * d if rs.rand() < 0.7 else None
    K = gpo.kernel.get_value(gpo._x); K[np.diag_indices_from(K)] += np.exp(-10.0)
    cond = np.linalg.cond(K)
    tol = max(1e-12, 500 * cond * EPS)
    with np.errstate(all="ignore"):
        mo, vo = gpo.predict(y, np.nan_to_num(T), return_var=True)
    asum = max(np.abs(gpo._compute_alpha(y, False)).sum(), 1e-300)
    amp_tot = float(np.max(np.diag(gpo.kernel.get_value(np.nan_to_num(T[:min(m, 50)])))))
    res = {}
    for mode in ("inverse", "solve"):
        g = mk(agp); g.variance_mode = mode
        bi, bu, u, mu, var = g.acquire(y, T, kind, bounds=bounds, mask=mask, return_all=True)
        res[mode] = (bi, bu, u, mu, var)
        ok = np.isfinite(T).all(axis=1)
        e_mu = np.abs(mu[ok] - mo[ok]).max() / asum
        e_var = np.abs(var[ok] - vo[ok]).max() / max(amp_tot, 1.0)
        nan_ok = (not (~ok).any()) or (np.isnan(mu[~ok]).all() and np.isnan(var[~ok]).all())
        adm = ok.copy()
        if bounds is not None:
            adm &= np.all((np.nan_to_num(T) >= -5) & (np.nan_to_num(T) <= 5), axis=1)
        if mask is not None:
            adm &= mask
        inf_ok = np.all(np.isposinf(u[~adm & ok])) if (~adm & ok).any() else True
        fin = np.where(np.isfinite(u), u, np.inf)
        arg_ok = (bi == -1 and not np.isfinite(fin).any()) or (bi >= 0 and bu == u[bi] and bi == int(np.argmin(fin)))
        one = g.predict(y, np.nan_to_num(T[0:1]), return_var=True)
        e_one = max(abs(one[0][0] - mo[0]) / asum, abs(one[1][0] - vo[0]) / max(amp_tot, 1.0))
        good = e_mu <= tol and e_var <= tol and nan_ok and inf_ok and arg_ok and e_one <= tol
        if not good:
            bad += 1
            print("MISMATCH case %d n=%d d=%d m=%d %s %s lin=%s: e_mu %.2e e_var %.2e tol %.2e nan %s inf %s arg %s one %.2e"
                  % (case, n, d, m, kind, mode, lin, e_mu, e_var, tol, nan_ok, inf_ok, arg_ok, e_one), flush=True)
    print("case %2d n=%4d d=%2d m=%5d %-5s lin=%d cond %.1e: ok (%.0f s)" % (case, n, d, m, kind, lin, cond, time.time() - t0), flush=True)
print("FUZZ", "FAILED %d" % bad if bad else "OK")
