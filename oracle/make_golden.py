# -*- coding: utf-8 -*-
"""
oracle/make_golden.py -- TEST INFRASTRUCTURE (build container only).

Generates the golden fixtures under ``tests/golden/`` by driving the REFERENCE's
own ``gpUtils`` / ``utility`` / ``approx`` / ``likelihood`` modules (imported
from /root/reference through ``oracle/ref_harness.py``; never copied) on top of
the george restatement in ``oracle/george_oracle.py``.

A fixture is data only: seeded inputs (theta, y, hyper-parameter vectors,
candidate sets) and the outputs the reference produced for them (parameter
vectors/names, per-candidate mu / var / AGP / BAPE / Jones utilities obtained by
calling the reference's scalar utilities one candidate at a time exactly as
``minimizeObjective`` does, log-likelihood, its gradient, ``_gpll`` tuples,
``optimizeGP`` and ``findNextPoint`` results).  Run:

    python -B oracle/make_golden.py

The GPU box never runs this (no /root/reference there); it only reads the
committed ``tests/golden/*.npz`` / ``*.json``.
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

from ref_harness import load_reference  # noqa: E402

load_reference()
from approxposterior import utility as ut, gpUtils, likelihood as lh, approx  # noqa: E402
import george  # noqa: E402  (stub bound to the oracle)
from scipy.optimize import rosen  # noqa: E402


def rosen_set(m0, corners=False, dim=2):
    """Inputs of the reference's known-answer tests (test_GPUtil.py:30-40)."""
    theta = np.array(lh.rosenbrockSample(m0, dim) if dim != 2 else lh.rosenbrockSample(m0))
    if corners:
        theta = np.array(list(theta) + [[-5, 5], [5, 5]])
    y = np.zeros(len(theta))
    for ii in range(len(theta)):
        y[ii] = lh.rosenbrockLnlike(theta[ii]) + lh.rosenbrockLnprior(theta[ii])
    return theta, y


def f(x):
    """Reference utilities return float, 0.0, inf or a (1,)-array (Q2)."""
    return float(np.asarray(x, dtype=float).ravel()[0])


def box_prior(lo, hi):
    def prior(theta):
        t = np.asarray(theta)
        if np.any(t < lo) or np.any(t > hi):
            return -np.inf
        return 0.0
    return prior


def sweep_case(name, theta, y, gp, cands, lo, hi, meta):
    """Per-candidate outputs through the reference's scalar utilities."""
    prior = box_prior(lo, hi)
    M = len(cands)
    mu = np.zeros(M); var = np.zeros(M)
    u_agp = np.zeros(M); u_bape = np.zeros(M); u_jones = np.zeros(M)
    for i in range(M):
        t = cands[i]
        m_, v_ = gp.predict(y, t.reshape(1, -1), return_var=True)
        mu[i] = m_[0]; var[i] = v_[0]
        with np.errstate(all="ignore"):
            u_agp[i] = f(ut.AGPUtility(t, y, gp, prior))
            u_bape[i] = f(ut.BAPEUtility(t, y, gp, prior))
            u_jones[i] = f(ut.JonesUtility(t, y, gp, prior))
    x = gp._x
    K = gp.kernel.get_value(x)
    K[np.diag_indices_from(K)] += np.exp(gp.white_noise.value)
    cond = float(np.linalg.cond(K))
    ll = gp.log_likelihood(y, quiet=True)
    grad = gp.grad_log_likelihood(y, quiet=True)
    alpha = gp._compute_alpha(y, False)
    out = dict(theta=np.atleast_2d(theta.T).T if theta.ndim == 1 else theta, y=y,
               p=gp.get_parameter_vector(),
               names=np.array(gp.get_parameter_names()),
               white_noise=gp.white_noise.value, cands=cands,
               lo=np.asarray(lo, dtype=float), hi=np.asarray(hi, dtype=float),
               mu=mu, var=var, u_agp=u_agp, u_bape=u_bape, u_jones=u_jones,
               ll=ll, grad=grad, alpha=alpha, logdet=gp.log_determinant,
               cond=cond, fit_amp=int(hasattr(gp.kernel, "k1")))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    meta[name] = dict(N=int(len(y)), D=int(x.shape[1]), M=int(M), cond=cond,
                      ll=float(ll))
    print("wrote", name, meta[name])


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {"generator": "oracle/make_golden.py",
            "reference": "dflemin3/approxposterior v0.4 (/root/reference)",
            "george": george.__version__,
            "shims": ["Q1/Q2: utility.minimize rebound to ravel x0 / float objective"]}
    pins = {}

    # ---- known-answer constants held by the reference's own tests ----------
    pins["reference_test_constants"] = {
        "test_InitGP.py:43": [-31.02658091, 9.78479362, -1.0552327, -1.16092752],
        "test_InitGP.py:76": [-31.02658091, -1.0552327, -1.16092752],
        "test_GPUtil.py:50": 31.92055252, "test_GPUtil.py:56": -114623.57332731,
        "test_GPUtil.py:62": -77.37826545, "test_GPUtil.py:101": 37.41585067,
        "test_GPUtil.py:107": 76.15271103, "test_GPUtil.py:113": 0.0,
        "test_OptimizeGP.py:50": [19.99668368, 4.18856645, 10.78000803],
        "test_OptimizeGP.py:91": [-1.54256578, 3.24723589],
        "test_findNewPoint.py:60": [-2.03449242, -3.07172107],
        "test_findNewPoint.py:107": [0.79813416, 0.85542199],
        "theta_test": [-2.3573, 4.673],
    }

    # ---- replay of the reference tests on the harness ----------------------
    replay = {}
    for amp in (True, False):
        tag = "amp" if amp else "noamp"
        np.random.seed(57)
        theta, y = rosen_set(50)
        gp = gpUtils.defaultGP(theta, y, fitAmp=amp)
        replay["initgp_" + tag] = dict(p=gp.get_parameter_vector().tolist(),
                                       names=list(gp.get_parameter_names()))
        np.random.seed(57)
        theta, y = rosen_set(20)
        gp = gpUtils.defaultGP(theta, y, fitAmp=amp)
        tt = np.array([-2.3573, 4.673])
        replay["util_" + tag] = dict(
            theta=theta.tolist(), y=y.tolist(), p=gp.get_parameter_vector().tolist(),
            agp=f(ut.AGPUtility(tt, y, gp, lh.rosenbrockLnprior)),
            bape=f(ut.BAPEUtility(tt, y, gp, lh.rosenbrockLnprior)),
            jones=f(ut.JonesUtility(tt, y, gp, lh.rosenbrockLnprior)))
        np.random.seed(57)
        theta, y = rosen_set(50)
        gp = gpUtils.defaultGP(theta, y, fitAmp=amp)
        with np.errstate(all="ignore"):
            gp = gpUtils.optimizeGP(gp, theta, y, seed=57, nGPRestarts=5)
        replay["optgp_" + tag] = dict(p=gp.get_parameter_vector().tolist(),
                                      ll=float(gp.log_likelihood(y, quiet=True)))
        # findNextPoint (test_findNewPoint.py)
        np.random.seed(57)
        theta, y = rosen_set(50, corners=True)
        gp = gpUtils.defaultGP(theta, y, fitAmp=amp)
        apo = approx.ApproxPosterior(theta=theta, y=y, gp=gp,
                                     lnprior=lh.rosenbrockLnprior,
                                     lnlike=lh.rosenbrockLnlike,
                                     priorSample=lh.rosenbrockSample,
                                     bounds=((-5, 5), (-5, 5)), algorithm="bape")
        with np.errstate(all="ignore"):
            thetaT = apo.findNextPoint(computeLnLike=False, bounds=((-5, 5), (-5, 5)),
                                       seed=57)
        replay["findnext_" + tag] = dict(thetaT=np.asarray(thetaT).tolist(),
                                         p=gp.get_parameter_vector().tolist())
        # _gpll guard cases (approx.py:148-189)
        gpll = []
        for t in ([0.5, 0.5], [-2.3573, 4.673], [6.0, 0.0], [np.inf, np.nan],
                  [np.nan, 1.0]):
            with np.errstate(all="ignore"):
                try:
                    r = apo._gpll(np.array(t))
                except Exception as e:  # pragma: no cover
                    r = ("raise", type(e).__name__)
            gpll.append(dict(theta=[None if not np.isfinite(v) else v for v in t],
                             theta_repr=[repr(float(v)) for v in t],
                             out=[repr(float(np.ravel(v)[0])) if not isinstance(v, str) else v
                                  for v in r]))
        replay["gpll_" + tag] = gpll
    pins["harness_replay"] = replay

    # ---- per-candidate sweep fixtures --------------------------------------
    rng = np.random.RandomState(1234)

    # S1/S2: 2-D Rosenbrock, N=50, default (seeded) hypers, with / without amp.
    for amp in (False, True):
        np.random.seed(57)
        theta, y = rosen_set(50)
        gp = gpUtils.defaultGP(theta, y, fitAmp=amp)
        cands = rng.uniform(-5.5, 5.5, size=(192, 2))
        cands[:8] = theta[:8] + 1e-3  # near-training points: tiny variance
        cands[8] = theta[3]           # exact training point: var ~ white noise
        sweep_case("rosen2d_n50_" + ("amp" if amp else "noamp"), theta, y, gp, cands,
                   [-5, -5], [5, 5], meta)

    # S3: 2-D Rosenbrock, N=50, optimised hypers (fitAmp=False: cond ~ 5e3).
    np.random.seed(57)
    theta, y = rosen_set(50)
    gp = gpUtils.defaultGP(theta, y, fitAmp=False)
    with np.errstate(all="ignore"):
        gp = gpUtils.optimizeGP(gp, theta, y, seed=57, nGPRestarts=2)
    cands = rng.uniform(-5.5, 5.5, size=(192, 2))
    sweep_case("rosen2d_n50_noamp_opt", theta, y, gp, cands, [-5, -5], [5, 5], meta)

    # S4: C2-small -- 2-D Rosenbrock, N=200 (not a multiple of any tile).
    np.random.seed(57)
    theta, y = rosen_set(200)
    gp = gpUtils.defaultGP(theta, y, fitAmp=False)
    gp.set_parameter_vector([np.median(y), np.log(2.0), np.log(3.0)])
    gp.recompute()
    cands = rng.uniform(-5.2, 5.2, size=(300, 2))
    sweep_case("c2small_d2_n200", theta, y, gp, cands, [-5, -5], [5, 5], meta)

    # S5: C3-small -- synthetic D=8, N=300, BASELINE.md section 4 recipe.
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(300, 8))
    y8 = np.array([-rosen(x) / 100.0 for x in X])
    kernel = george.kernels.ExpSquaredKernel(metric=np.full(8, 8.0), ndim=8)
    gp = george.GP(kernel=kernel, fit_mean=True, mean=np.median(y8),
                   white_noise=-12, fit_white_noise=False)
    gp.compute(X)
    cands = np.random.RandomState(1).uniform(-5.2, 5.2, size=(320, 8))
    sweep_case("c3small_d8_n300", X, y8, gp, cands, [-5] * 8, [5] * 8, meta)

    # S6: D=5 (padded feature dim), N=130, with amplitude, anisotropic metric.
    rs = np.random.RandomState(5)
    X = rs.uniform(-5, 5, size=(130, 5))
    y5 = np.array([-rosen(x) / 100.0 for x in X])
    kernel = 2.5 * george.kernels.ExpSquaredKernel(metric=[3.0, 5.0, 8.0, 2.0, 6.0], ndim=5)
    gp = george.GP(kernel=kernel, fit_mean=True, mean=np.median(y5),
                   white_noise=-12, fit_white_noise=False)
    gp.compute(X)
    cands = rs.uniform(-5.2, 5.2, size=(257, 5))
    sweep_case("d5_n130_amp", X, y5, gp, cands, [-5] * 5, [5] * 5, meta)

    # S7: D=1 Bayesian-optimisation test function (likelihood.py:120-170), Jones.
    np.random.seed(91)
    th1 = np.array(lh.testBOFnSample(12))
    y1 = np.array([lh.testBOFn(t) + lh.testBOFnLnPrior(t) for t in th1])
    gp = gpUtils.defaultGP(th1, y1, fitAmp=True)
    cands = np.linspace(-1.2, 2.2, 150).reshape(-1, 1)
    sweep_case("bo1d_n12_amp", th1, y1, gp, cands, [-1], [2], meta)

    # S8: ILL-CONDITIONED -- 2-D Rosenbrock, N=50, fitAmp=True at the optimised hypers
    # (cond(K) ~ 1e16, SURVEY.md section 7).  Besides the oracle outputs the fixture
    # carries a 60-digit mpmath truth for mu / var so that the solve-based HIP path is
    # judged against exact arithmetic, not against the reference's own noise.
    np.random.seed(57)
    theta, y = rosen_set(50)
    gp = gpUtils.defaultGP(theta, y, fitAmp=True)
    gp.set_parameter_vector(replay["optgp_amp"]["p"])
    gp.recompute()
    cands = rng.uniform(-5.0, 5.0, size=(24, 2))
    sweep_case("rosen2d_n50_amp_opt_illcond", theta, y, gp, cands, [-5, -5], [5, 5], meta)
    import mpmath as mp
    mp.mp.dps = 60
    p = gp.get_parameter_vector()
    amp = mp.mpf(2) * mp.exp(mp.mpf(float(p[1])))
    w = [mp.exp(-mp.mpf(float(v))) for v in p[2:]]
    wn = mp.exp(mp.mpf(-12))
    X = [[mp.mpf(float(v)) for v in row] for row in theta]
    def kfun(a, b):
        return amp * mp.exp(-mp.mpf(1) / 2 * sum(w[d] * (a[d] - b[d]) ** 2 for d in range(2)))
    n = len(X)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = kfun(X[i], X[j]) + (wn if i == j else 0)
    r = mp.matrix([mp.mpf(float(v)) - mp.mpf(float(p[0])) for v in y])
    alpha = mp.lu_solve(K, r)
    mu_t, var_t = [], []
    for c in cands:
        cc = [mp.mpf(float(v)) for v in c]
        ks = mp.matrix([kfun(cc, X[i]) for i in range(n)])
        sol = mp.lu_solve(K, ks)
        mu_t.append(float(sum(ks[i] * alpha[i] for i in range(n)) + mp.mpf(float(p[0]))))
        var_t.append(float(amp - sum(ks[i] * sol[i] for i in range(n))))
    d = dict(np.load(os.path.join(OUT, "rosen2d_n50_amp_opt_illcond.npz")))
    d["mu_truth"] = np.array(mu_t)
    d["var_truth"] = np.array(var_t)
    np.savez_compressed(os.path.join(OUT, "rosen2d_n50_amp_opt_illcond.npz"), **d)
    print("ill-conditioned: oracle var rel err vs truth (median, max):",
          np.median(np.abs(d["var"] - d["var_truth"]) / np.abs(d["var_truth"])),
          np.max(np.abs(d["var"] - d["var_truth"]) / np.abs(d["var_truth"])))

    with open(os.path.join(OUT, "pins.json"), "w") as fh:
        json.dump(pins, fh, indent=1)
    with open(os.path.join(OUT, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("done")


def d8_truth(ncand=30, dps=50):
    """Independent truth at D = 8 (VERDICT round 1, item 6): for the ``c3small_d8_n300``
    fixture (N = 300, D = 8, ExpSquaredKernel without amplitude, white noise e^-12) compute
    the GP algebra of SURVEY.md Appendix A.2-A.7 in %d-digit mpmath arithmetic -- Gram matrix,
    Cholesky factor, alpha, log-likelihood, and mu / sigma^2 at the first ``ncand`` candidates --
    straight from the formulas, sharing NO code with the oracle or the HIP path, and store it
    in the fixture (``truth_idx``, ``mu_truth``, ``var_truth``, ``ll_truth``).  Both the oracle
    (tests/test_oracle_pins.py) and the HIP path (tests/test_gpu_parity.py) are asserted
    against it.""" % 50
    import mpmath as mp
    mp.mp.dps = dps
    path = os.path.join(OUT, "c3small_d8_n300.npz")
    d = dict(np.load(path))
    X = [[mp.mpf(float(v)) for v in row] for row in d["theta"]]
    y = [mp.mpf(float(v)) for v in d["y"]]
    p = d["p"]
    assert int(d["fit_amp"]) == 0 and len(p) == 9
    mean = mp.mpf(float(p[0]))
    w = [mp.exp(-mp.mpf(float(v))) for v in p[1:]]          # 1 / M_d
    wn = mp.exp(mp.mpf(float(d["white_noise"])))
    n, D = len(X), 8
    half = mp.mpf(1) / 2

    def kfun(a, b):
        return mp.exp(-half * sum(w[k] * (a[k] - b[k]) ** 2 for k in range(D)))
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(i + 1):
            v = kfun(X[i], X[j])
            K[i, j] = v
            K[j, i] = v
        K[i, i] += wn
    L = mp.cholesky(K)

    def fwd(b):                                             # L x = b
        x = [mp.mpf(0)] * n
        for i in range(n):
            x[i] = (b[i] - sum(L[i, k] * x[k] for k in range(i))) / L[i, i]
        return x

    def bwd(b):                                             # L^T x = b
        x = [mp.mpf(0)] * n
        for i in reversed(range(n)):
            x[i] = (b[i] - sum(L[k, i] * x[k] for k in range(i + 1, n))) / L[i, i]
        return x
    r = [y[i] - mean for i in range(n)]
    z = fwd(r)
    alpha = bwd(z)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    ll = -half * sum(v * v for v in z) - half * logdet - mp.mpf(n) / 2 * mp.log(2 * mp.pi)
    idx = list(range(ncand))
    mu_t, var_t = [], []
    for c in d["cands"][:ncand]:
        cc = [mp.mpf(float(v)) for v in c]
        ks = [kfun(cc, X[i]) for i in range(n)]
        v = fwd(ks)
        mu_t.append(float(sum(ks[i] * alpha[i] for i in range(n)) + mean))
        var_t.append(float(1 - sum(t * t for t in v)))
    d["truth_idx"] = np.array(idx)
    d["mu_truth"] = np.array(mu_t)
    d["var_truth"] = np.array(var_t)
    d["ll_truth"] = np.array(float(ll))
    d["alpha_truth"] = np.array([float(a) for a in alpha])
    np.savez_compressed(path, **d)
    print("D=8 truth: oracle |mu - truth| max %.3e, |var - truth| max %.3e, |ll - truth| %.3e" % (
        np.abs(d["mu"][:ncand] - d["mu_truth"]).max(), np.abs(d["var"][:ncand] - d["var_truth"]).max(),
        abs(float(d["ll"]) - float(ll))))


def _mp_truth_amp2d(theta, y, p, cands, dps=60):
    """mu / sigma^2 / alpha of a 2-D amplitude * ExpSquared GP (white noise e^-12) in
    ``dps``-digit mpmath arithmetic, straight from SURVEY.md Appendix A.2-A.7 (LU solves on
    the Gram matrix; no code shared with the oracle or the HIP path)."""
    import mpmath as mp
    mp.mp.dps = dps
    amp = mp.mpf(2) * mp.exp(mp.mpf(float(p[1])))
    w = [mp.exp(-mp.mpf(float(v))) for v in p[2:]]
    wn = mp.exp(mp.mpf(-12))
    X = [[mp.mpf(float(v)) for v in row] for row in theta]

    def kfun(a, b):
        return amp * mp.exp(-mp.mpf(1) / 2 * sum(w[d] * (a[d] - b[d]) ** 2 for d in range(2)))
    n = len(X)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = kfun(X[i], X[j]) + (wn if i == j else 0)
    r = mp.matrix([mp.mpf(float(v)) - mp.mpf(float(p[0])) for v in y])
    alpha = mp.lu_solve(K, r)
    mu_t, var_t = [], []
    for c in cands:
        cc = [mp.mpf(float(v)) for v in c]
        ks = mp.matrix([kfun(cc, X[i]) for i in range(n)])
        sol = mp.lu_solve(K, ks)
        mu_t.append(float(sum(ks[i] * alpha[i] for i in range(n)) + mp.mpf(float(p[0]))))
        var_t.append(float(amp - sum(ks[i] * sol[i] for i in range(n))))
    return np.array(mu_t), np.array(var_t), np.array([float(a) for a in alpha])


def cond_ladder(targets=(1e8, 1e11, 1e13)):
    """Conditioning ladder (VERDICT round 2, item 1b): the reference's own 2-D Rosenbrock
    training set (N = 50, fitAmp=True) at hyper-parameters on the straight line between
    ``defaultGP``'s initial vector and the reference's optimum (test_OptimizeGP.py:50, true
    cond(K) 8.5e15), stopped where the TRUE 2-norm condition number of K reaches 1e8 / 1e11 /
    1e13.  Each fixture carries the oracle's outputs through the reference's scalar utilities
    and a 60-digit mpmath truth for mu, sigma^2 and alpha, so that both variance formulations
    of the HIP path (explicit L^-1 contraction and blocked substitution) can be judged against
    exact arithmetic between the well-conditioned fixtures (cond <= 4.7e6) and the optimum."""
    from scipy.optimize import brentq
    with open(os.path.join(OUT, "pins.json")) as fh:
        pins = json.load(fh)
    with open(os.path.join(OUT, "meta.json")) as fh:
        meta = json.load(fh)
    popt = np.array(pins["harness_replay"]["optgp_amp"]["p"])
    np.random.seed(57)
    theta, y = rosen_set(50)
    gp = gpUtils.defaultGP(theta, y, fitAmp=True)
    p0 = gp.get_parameter_vector().copy()

    def cond_at(s):
        gp.set_parameter_vector(p0 + s * (popt - p0))
        K = gp.kernel.get_value(theta)
        K[np.diag_indices_from(K)] += np.exp(gp.white_noise.value)
        return float(np.linalg.cond(K))
    cands = np.random.RandomState(4321).uniform(-5.0, 5.0, size=(24, 2))
    for tgt in targets:
        s = brentq(lambda v: np.log10(cond_at(v)) - np.log10(tgt), 0.2, 1.0, xtol=1e-8)
        p = p0 + s * (popt - p0)
        gp.set_parameter_vector(p)
        gp.recompute()
        name = "rosen2d_n50_amp_cond1e%d" % int(round(np.log10(tgt)))
        sweep_case(name, theta, y, gp, cands, [-5, -5], [5, 5], meta)
        mu_t, var_t, alpha_t = _mp_truth_amp2d(theta, y, p, cands)
        d = dict(np.load(os.path.join(OUT, name + ".npz")))
        d["mu_truth"], d["var_truth"], d["alpha_truth"] = mu_t, var_t, alpha_t
        d["path_s"] = np.array(s)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        rel = np.abs(d["var"] - var_t) / np.abs(var_t)
        print("%s: cond %.3e  oracle var rel err vs truth median %.2e max %.2e; alpha %.2e" % (
            name, d["cond"], np.median(rel), rel.max(),
            np.abs(d["alpha"] - alpha_t).max() / np.abs(alpha_t).max()))
    # the optimum's fixture gets its alpha truth too
    name = "rosen2d_n50_amp_opt_illcond"
    d = dict(np.load(os.path.join(OUT, name + ".npz")))
    if "alpha_truth" not in d:
        _, _, d["alpha_truth"] = _mp_truth_amp2d(d["theta"], d["y"], d["p"], d["cands"][:1])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    with open(os.path.join(OUT, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def _mp_grad_amp2d(theta, y, p, dps=60):
    """Gradient of the log-likelihood w.r.t. george's parameter vector (mean, log_constant, log_M_0_0, log_M_1_1) of a
    2-D amplitude * ExpSquared GP (white noise e^-12) in ``dps``-digit mpmath arithmetic, straight from SURVEY.md
    Appendix A.6: g_mean = sum(alpha), g_k = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k) with dK/dlog_constant = K
    (without the white noise) and dK/dlog_M_d = K o (w_d (x_id - x_jd)^2 / 2).  No code shared with the oracle or
    the HIP path."""
    import mpmath as mp
    mp.mp.dps = dps
    amp = mp.mpf(2) * mp.exp(mp.mpf(float(p[1])))
    w = [mp.exp(-mp.mpf(float(v))) for v in p[2:]]
    wn = mp.exp(mp.mpf(-12))
    X = [[mp.mpf(float(v)) for v in row] for row in theta]
    n = len(X)
    Kc = mp.matrix(n, n)           # the kernel part (no white noise)
    for i in range(n):
        for j in range(n):
            Kc[i, j] = amp * mp.exp(-mp.mpf(1) / 2 * sum(w[d] * (X[i][d] - X[j][d]) ** 2 for d in range(2)))
    K = Kc.copy()
    for i in range(n):
        K[i, i] += wn
    Kinv = mp.inverse(K)
    r = mp.matrix([mp.mpf(float(v)) - mp.mpf(float(p[0])) for v in y])
    alpha = Kinv * r
    g = [sum(alpha[i] for i in range(n))]
    g.append(sum((alpha[i] * alpha[j] - Kinv[i, j]) * Kc[i, j] for i in range(n) for j in range(n)) / 2)
    for d in range(2):
        g.append(sum((alpha[i] * alpha[j] - Kinv[i, j]) * Kc[i, j] * w[d] * (X[i][d] - X[j][d]) ** 2 / 2
                     for i in range(n) for j in range(n)) / 2)
    return np.array([float(v) for v in g])


def grad_truth():
    """VERDICT round 3, item 2: every rung of the conditioning ladder and the reference's optimum (8.5e15) get a
    60-digit mpmath gradient (``grad_truth``) beside the oracle's (``grad``, already in the fixtures), so that the HIP
    gradient -- through the explicit inverse below the conditioning gate, through triangular solves above it, as george
    forms K^-1 by cho_solve -- can be judged against exact arithmetic.  Needs no reference import."""
    for name in ("rosen2d_n50_amp_cond1e8", "rosen2d_n50_amp_cond1e11", "rosen2d_n50_amp_cond1e13",
                 "rosen2d_n50_amp_opt_illcond"):
        d = dict(np.load(os.path.join(OUT, name + ".npz")))
        d["grad_truth"] = _mp_grad_amp2d(d["theta"], d["y"], d["p"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        err = np.abs(d["grad"] - d["grad_truth"]) / np.abs(d["grad_truth"])
        print("%s: cond %.3e  gradient truth %s; oracle rel err %s" % (name, float(d["cond"]), d["grad_truth"], err))


if __name__ == "__main__":
    if "--d8-truth" in sys.argv:
        d8_truth()          # (augments the committed fixture; needs no reference import)
    elif "--cond-ladder" in sys.argv:
        cond_ladder()       # (adds the three mid-conditioning fixtures; leaves the others alone)
    elif "--grad-truth" in sys.argv:
        grad_truth()        # (augments the ladder fixtures and the optimum's; needs no reference import)
    else:
        main()
        d8_truth()
        cond_ladder()
        grad_truth()
