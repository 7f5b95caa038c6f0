"""GPU (``-m gpu``): the callers on either side of the hot path, end to end on
the HIP-backed GP, tested the way the reference tests them (statistical /
integration tests, SURVEY.md section 4 style 2):
  test_APRun.py:58-73        posterior means within one 'true sigma' of (0.0, 1.31)
  test_1DBayesOpt.py:55-73   bayesOpt reaches the optimum within 5 %
  test_MAP.py:51-65          findMAP within 1e-3 of the sphere optimum
plus the batched extras: sweep-based findNextPoint vs restarted Nelder-Mead and
_gpllBatch vs scalar _gpll."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import minimize

pytestmark = pytest.mark.gpu


def _rosen_ap(m0, fit_amp=False, algorithm="bape"):
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    theta = np.array(lh.rosenbrockSample(m0))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y, fitAmp=fit_amp)
    return approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                  lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                  bounds=[(-5, 5), (-5, 5)], algorithm=algorithm)


def test_run_rosenbrock_posterior(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    np.random.seed(57)
    ap = _rosen_ap(50)
    with np.errstate(all="ignore"):
        ap.run(m=20, nmax=3, estBurnin=True, nGPRestarts=3, mcmcKwargs={"iterations": 5000},
               cache=False, samplerKwargs={"nwalkers": 20}, verbose=False, thinChains=False,
               onlyLastMCMC=False, kmax=2, eps=1, convergenceCheck=True)
    samples = ap.sampler.get_chain(discard=ap.iburns[-1], flat=True, thin=ap.ithins[-1])
    means = np.mean(samples, axis=0)
    z = np.fabs((means - np.array([0.0, 1.31])) / np.array([1.5, 1.75]))
    assert np.all(z < 1), (means, z)
    assert len(ap.y) >= 50 + 40


def test_c1_readme_example_as_written(tmp_path, monkeypatch):
    """BASELINE.json configs[0] / the README run EXACTLY as the reference writes it
    (/root/reference/examples/inference/example.py:16-52): seed 57, m0 = 50 prior draws, m = 20, nmax = 2,
    BAPE, defaultGP(white_noise=-12), 20 walkers x 2e4 iterations, estBurnin, nGPRestarts = 3, onlyLastMCMC -- on
    the HIP-backed GP, with the statistical assertion of the reference's own run test (test_APRun.py:66-73:
    posterior means within one 'true sigma' of (0.0, 1.31))."""
    monkeypatch.chdir(tmp_path)
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    m0, m, nmax = 50, 20, 2
    np.random.seed(57)
    theta = lh.rosenbrockSample(m0)
    y = np.zeros(len(theta))
    for ii in range(len(theta)):
        y[ii] = lh.rosenbrockLnlike(theta[ii]) + lh.rosenbrockLnprior(theta[ii])
    gp = gpUtils.defaultGP(theta, y, white_noise=-12)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lh.rosenbrockLnlike,
                                priorSample=lh.rosenbrockSample, bounds=[(-5, 5), (-5, 5)], algorithm="bape")
    with np.errstate(all="ignore"):
        ap.run(m=m, nmax=nmax, estBurnin=True, nGPRestarts=3, mcmcKwargs={"iterations": int(2.0e4)},
               cache=False, samplerKwargs={"nwalkers": 20}, verbose=False, thinChains=False, onlyLastMCMC=True)
    assert len(ap.y) == m0 + m * nmax and ap.theta.shape == (m0 + m * nmax, 2)
    assert ap.sampler.get_chain().shape == (20000, 20, 2)
    samples = ap.sampler.get_chain(discard=ap.iburns[-1], flat=True, thin=ap.ithins[-1])
    means = np.mean(samples, axis=0)
    z = np.fabs((means - np.array([0.0, 1.31])) / np.array([1.5, 1.75]))
    assert np.all(z < 1), (means, z)


def test_bayesopt_1d(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    np.random.seed(57)
    fn = lambda x: -(lh.testBOFn(x) + lh.testBOFnLnPrior(x))   # noqa: E731
    true = minimize(lambda x: float(np.ravel(fn(x))[0]), np.ravel(lh.testBOFnSample(1)), method="nelder-mead")
    theta = lh.testBOFnSample(3)
    y = np.array([lh.testBOFn(t) + lh.testBOFnLnPrior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y, fitAmp=True)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.testBOFnLnPrior,
                                lnlike=lh.testBOFn, priorSample=lh.testBOFnSample,
                                bounds=[[-1, 2]], algorithm="jones")
    with np.errstate(all="ignore"):
        soln = ap.bayesOpt(nmax=10, tol=1.0e-3, seed=57, verbose=False, cache=False,
                           gpMethod="powell", optGPEveryN=1, nGPRestarts=3, nMinObjRestarts=5,
                           initGPOpt=True, minObjMethod="nelder-mead", findMAP=True)
    assert np.allclose(soln["thetaBest"], true["x"], rtol=5.0e-2)
    assert np.allclose(soln["valBest"], -true["fun"], rtol=5.0e-2)
    assert np.allclose(soln["thetaMAPBest"], true["x"], rtol=5.0e-2)
    assert np.allclose(soln["valMAPBest"], -true["fun"], rtol=5.0e-2)


def test_find_map_sphere(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    np.random.seed(57)
    theta = np.array(lh.sphereSample(20))
    y = np.array([lh.sphereLnlike(t) + lh.sphereLnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y, fitAmp=True)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.sphereLnprior,
                                lnlike=lh.sphereLnlike, priorSample=lh.sphereSample,
                                bounds=[(-5, 5), (-5, 5)], algorithm="jones")
    with np.errstate(all="ignore"):
        ap.optGP(seed=57, method="powell", nGPRestarts=3)
        ap.findNextPoint(numNewPoints=5, nGPRestarts=3, cache=False, verbose=False)
        testMAP, testVal = ap.findMAP(nRestarts=15)
    assert np.allclose([0.0, 0.0], testMAP, atol=1.0e-3)
    assert np.allclose(0.0, testVal, atol=1.0e-3)


def test_sweep_point_search_vs_nelder_mead():
    """The fused sweep over 2e5 prior draws finds a utility at least as good as
    the reference-style 5-restart Nelder-Mead search (up to the grid resolution),
    and gpllBatch equals the scalar guard-by-guard path."""
    from approxposterior_amd import utility as ut
    np.random.seed(57)
    ap = _rosen_ap(60)
    with np.errstate(all="ignore"):
        ap.optGP(seed=57, nGPRestarts=2)
        thetaNM, uNM = ut.minimizeObjective(ap.utility, ap.y, ap.gp, sampleFn=ap.priorSample,
                                            priorFn=ap._lnprior, nRestarts=5,
                                            args=(ap.y, ap.gp, ap._lnprior))
        thetaS = ap.findNextPoint(computeLnLike=False, nCandidates=200000, verbose=False)
        uS = ut.BAPEUtility(thetaS, ap.y, ap.gp, ap._lnprior)
        thetaP = ap.findNextPoint(computeLnLike=False, nCandidates=200000, polish=True, verbose=False)
        uP = ut.BAPEUtility(thetaP, ap.y, ap.gp, ap._lnprior)
    uNM, uS, uP = (float(np.ravel(v)[0]) for v in (uNM, uS, uP))
    assert uS <= uNM + 0.05 * abs(uNM) + 1e-6, (uS, uNM)
    assert uP <= uS + 1e-9 and uP <= uNM + 1e-6 * abs(uNM) + 1e-9, (uP, uS, uNM)
    thetas = np.array([[0.5, 0.5], [-2.3573, 4.673], [6.0, 0.0], [np.inf, np.nan], [np.nan, 1.0]])
    with np.errstate(all="ignore"):
        lp, blob = ap._gpllBatch(thetas)
        for i, t in enumerate(thetas):
            a, b = ap._gpll(t)
            a = float(np.ravel(a)[0]); b = float(np.ravel(b)[0])
            assert (np.isnan(b) and np.isnan(blob[i])) or b == blob[i]
            assert a == lp[i] or np.isclose(a, lp[i], rtol=1e-12)


def test_device_ensemble_sampler_matches_host_sampler():
    """The persistent-kernel sampler (GP.sample_ensemble / runMCMC(onDevice=True)):
    (1) every stored log-probability IS the GP mean at the stored coordinates
    (deterministic check of the in-kernel surrogate), (2) the chain never leaves
    the box prior, (3) its posterior moments and acceptance rate agree with the
    host stretch-move sampler driving the same GP within Monte-Carlo error,
    (4) independent ensembles (replicas) are statistically consistent."""
    from approxposterior_amd import mcmcUtils
    np.random.seed(57)
    ap = _rosen_ap(80)
    with np.errstate(all="ignore"):
        ap.optGP(seed=57, nGPRestarts=2)
    W, iters = 40, 6000
    p0 = np.array(ap.priorSample(W))
    res = ap.gp.sample_ensemble(ap.y, p0, iters, ap.bounds, seed=11)
    chain, lpc = res["chain"], res["log_prob"]
    assert chain.shape == (iters, W, 2) and lpc.shape == (iters, W)
    assert np.all(np.abs(chain) <= 5.0)
    idx = np.random.RandomState(0).randint(0, iters, size=60)
    for i in idx[:6]:
        mu = ap.gp.predict(ap.y, chain[i], return_cov=False, return_var=False)
        assert np.allclose(mu, lpc[i], rtol=1e-9, atol=1e-9)
    # host sampler on the same surrogate
    with np.errstate(all="ignore"):
        sampler, iburn, ithin = ap.runMCMC(samplerKwargs={"nwalkers": W, "seed": 3},
                                           mcmcKwargs={"iterations": iters, "initial_state": p0},
                                           cache=False, estBurnin=True, thinChains=True)
        dsamp, dburn, dthin = ap.runMCMC(samplerKwargs={"nwalkers": W},
                                         mcmcKwargs={"iterations": iters, "initial_state": p0},
                                         cache=False, estBurnin=True, thinChains=True, onDevice=True)
    h = sampler.get_chain(discard=max(iburn, 500), flat=True)
    d = dsamp.get_chain(discard=max(dburn, 500), flat=True)
    tau = max(np.max(sampler.get_autocorr_time(tol=0)), np.max(dsamp.get_autocorr_time(tol=0)))
    neff = min(len(h), len(d)) / (2.0 * tau)
    se = np.sqrt(h.var(axis=0) / neff + d.var(axis=0) / neff)
    print("tau %.1f neff %.0f host mean %s std %s | device mean %s std %s" % (
        tau, neff, h.mean(axis=0), h.std(axis=0), d.mean(axis=0), d.std(axis=0)))
    assert np.all(np.abs(h.mean(axis=0) - d.mean(axis=0)) < 5 * se), (h.mean(axis=0), d.mean(axis=0), se)
    # a standard deviation estimated from neff effective samples has relative error ~1/sqrt(2 neff)
    assert np.allclose(h.std(axis=0), d.std(axis=0), rtol=max(0.1, 6.0 / np.sqrt(2.0 * neff)))
    assert abs(sampler.acceptance_fraction.mean() - dsamp.acceptance_fraction.mean()) < 0.08
    assert 0.5 * iburn <= dburn <= 2.0 * iburn + 20
    # replicas: 4 independent ensembles in one launch
    p4 = np.array(ap.priorSample(4 * W)).reshape(4, W, 2)
    r4 = ap.gp.sample_ensemble(ap.y, p4, 3000, ap.bounds, seed=5)
    c4 = r4["chain"][500:].reshape(-1, 4, W, 2)
    m4 = c4.mean(axis=(0, 2))
    assert m4.shape == (4, 2) and np.all(np.abs(m4 - d.mean(axis=0)) < 8 * se + 0.15)
    assert not np.allclose(c4[:, 0], c4[:, 1])


def test_default_gp_with_linear_order_end_to_end():
    """defaultGP(order=1) (gpUtils.py:167-173): the reference's optional linear-regression
    term through defaultGP -> optGP -> findNextPoint on the device path; the initial GP is
    the oracle's for the same random draw."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import george_oracle as go
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    np.random.seed(57)
    theta = np.array(lh.rosenbrockSample(40))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    np.random.seed(3)
    gp = gpUtils.defaultGP(theta, y, order=1, fitAmp=True)
    np.random.seed(3)
    metric = np.fabs(np.random.randn(2))
    ko = np.var(y) * go.ExpSquaredKernel(metric, ndim=2) + (np.var(y) / 10.0) * go.kernels.LinearKernel(
        log_gamma2=metric[0], order=1, bounds=None, ndim=2)
    gpo = go.GP(kernel=ko, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gpo.compute(theta)
    assert gp.get_parameter_names() == gpo.get_parameter_names() and len(gp) == 6
    assert np.allclose(gp.get_parameter_vector(), gpo.get_parameter_vector())
    assert abs(gp.log_likelihood(y) - gpo.log_likelihood(y)) <= 1e-6 * abs(gpo.log_likelihood(y))
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                bounds=[(-5, 5), (-5, 5)], algorithm="bape")
    with np.errstate(all="ignore"):
        ap.optGP(seed=1, method="powell", nGPRestarts=1)
        tT, yT = ap.findNextPoint(computeLnLike=True, seed=5)
    assert np.all(np.isfinite(tT)) and np.all(np.abs(tT) <= 5) and np.isfinite(yT)
    assert len(ap.gp.get_parameter_vector()) == 6 and ap.gp.computed and len(ap.y) == 41
    assert np.isfinite(ap.gp.log_likelihood(ap.y))


@pytest.mark.parametrize("n,d", [(50, 2), (64, 3), (65, 2), (90, 2), (128, 8), (129, 4), (700, 5), (1152, 8), (1601, 3)])
def test_nll_batch_is_bit_identical_to_single_evaluations(n, d):
    """apgp_nll_eval_batch: several hyper-vectors through ONE batched
    Gram + Cholesky + solve.  Every entry equals gpUtils._nll of that vector exactly --
    including +inf for a non-positive-definite matrix -- and the GP's own state survives.
    The single evaluation is ONE fused launch for n <= 128 (nll_small_kernel / nll_two_kernel), the batch too (a workgroup
    per matrix); above, and in mode 1, the separate Gram / panel / step / finish launches: same code and operation order,
    so the same bits, at the block boundaries +- 1."""
    import time
    from approxposterior_amd import gpUtils
    rs = np.random.RandomState(5)
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.sin(X).sum(axis=1) + 0.1 * rs.randn(n)
    np.random.seed(2)
    gp = gpUtils.defaultGP(X, y, fitAmp=True)
    p_own = np.array(gp.get_parameter_vector())
    P = np.array([p_own + 0.3 * rs.randn(len(p_own)) for _ in range(7)])
    P[3, 1] = 750.0                 # amplitude overflows: the factorisation must fail
    t0 = time.time()
    with np.errstate(all="ignore"):
        batch = gp.nll_batch(P, y)
    t_batch = time.time() - t0
    assert np.array_equal(gp.get_parameter_vector(), p_own)
    t0 = time.time()
    with np.errstate(all="ignore"):
        single = np.array([gpUtils._nll(p, gp, y, None) for p in P])
    t_single = time.time() - t0
    assert np.array_equal(batch, single), (batch, single)
    assert np.isinf(batch[3]) and np.all(np.isfinite(np.delete(batch, 3)))
    # round 5: up to n = 128 the batch is ONE launch of the fused evaluation (a workgroup per matrix, records through the
    # stream's pinned staging area) -- against the separate Gram / panel / step / finish launches (mode 1): the same bits
    from approxposterior_amd import _lib
    lib = _lib.load()
    lib.apgp_potrf_mode(1)
    try:
        with np.errstate(all="ignore"):
            batch_ml = gp.nll_batch(P, y)
    finally:
        lib.apgp_potrf_mode(0)
    assert np.array_equal(batch, batch_ml), (batch, batch_ml)
    if n > 128:
        # round 6: a batch of 2 .. 6 mid-size matrices = their persistent factorisations side by side in ONE launch (own
        # scratch, flags and record each, 1 / batch of the CUs each) -- the batches a Powell look-ahead asks for; the same
        # bits as the single calls (7 matrices, above: the batched launch-per-step path)
        before = lib.apgp_nll_side_batches()
        for B in (2, 3, 4, 6):
            with np.errstate(all="ignore"):
                assert np.array_equal(gp.nll_batch(P[:B], y), single[:B])
        assert lib.apgp_nll_side_batches() == before + 4 or lib.apgp_potrf_backoff_skips() > 0
    print("nll_batch N=%d: 7 evaluations %.2f ms batched, %.2f ms one by one" % (n, 1e3 * t_batch, 1e3 * t_single))


def test_nll_batch_oversubscribed_is_bit_identical_to_single_evaluations():
    """More matrices in one batched Cholesky than the chip holds workgroups for (120 x N=450:
    120 x 28 tiles per step against ~512 resident workgroups), so sibling workgroups of one
    matrix start at different times.  A panel workgroup reads the raw diagonal block when it
    starts and the factored block is stored when another one ends: the result must not depend
    on that order (potrf.hip routes the factored blocks through a scratch).  Bit-identical to
    one-by-one evaluation."""
    from approxposterior_amd import gpUtils
    n, d, m = 450, 3, 120
    rs = np.random.RandomState(11)
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.sin(X).sum(axis=1) + 0.1 * rs.randn(n)
    np.random.seed(3)
    gp = gpUtils.defaultGP(X, y, fitAmp=True)
    p_own = np.array(gp.get_parameter_vector())
    P = np.array([p_own + 0.2 * rs.randn(len(p_own)) for _ in range(m)])
    with np.errstate(all="ignore"):
        batch = np.array([gp.nll_batch(P, y) for _ in range(3)])
        single = np.array([gpUtils._nll(p, gp, y, None) for p in P])
    assert np.all(np.isfinite(single))
    for b in batch:
        assert np.array_equal(b, single)


def test_nll_from_concurrent_host_threads():
    """Two host threads evaluating _nll on their own GP objects at the same time (ctypes releases
    the GIL; both enqueue on the current stream): each factorisation's launches must stay a unit --
    the Cholesky keeps per-stream scratch -- and every value must equal the single-threaded one."""
    import threading
    from approxposterior_amd import gpUtils
    n, d = 200, 3
    rs = np.random.RandomState(21)
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.sin(X).sum(axis=1) + 0.1 * rs.randn(n)
    np.random.seed(4)
    gps = [gpUtils.defaultGP(X, y, fitAmp=True) for _ in range(2)]
    p0 = np.array(gps[0].get_parameter_vector())
    P = [np.array([p0 + 0.1 * rs.randn(len(p0)) for _ in range(60)]) for _ in range(2)]
    with np.errstate(all="ignore"):
        want = [np.array([gpUtils._nll(p, gps[k], y, None) for p in P[k]]) for k in range(2)]
    got = [None, None]

    def work(k):
        with np.errstate(all="ignore"):
            got[k] = np.array([gpUtils._nll(p, gps[k], y, None) for p in P[k]])

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert np.array_equal(got[k], want[k])


def test_nll_and_scalar_predict_on_two_streams_concurrently():
    """Two host threads, each under its OWN torch stream, evaluating _nll (mailbox + stream scratch), the
    single-candidate predict and the pinned mean path at the same time: scratch, mailbox and enqueue lock
    are kept per (device, stream), so nothing may leak between the streams and every value must equal
    the single-threaded, default-stream one.  Sizes on both sides of the fused-launch boundary."""
    import threading
    import torch
    from approxposterior_amd import gpUtils, _lib
    rs = np.random.RandomState(31)
    cases = []
    for n, d in ((60, 2), (200, 3)):
        X = rs.uniform(-5, 5, size=(n, d))
        y = np.sin(X).sum(axis=1) + 0.1 * rs.randn(n)
        np.random.seed(6)
        gp = gpUtils.defaultGP(X, y, fitAmp=False)
        p0 = np.array(gp.get_parameter_vector())
        P = np.array([p0 + 0.05 * rs.randn(len(p0)) for _ in range(40)])
        T = rs.uniform(-5, 5, size=(40, d))
        cases.append((gp, y, P, T))

    def run(gp, y, P, T):
        out = []
        gp._nllMemo = None        # (round 5: an exact repeat of P[i] would be answered from gpUtils._nll's table and leave the
        #                           factorisation to predict(): equally valid, other last bits -- not what this test is about)
        with np.errstate(all="ignore"):
            for i in range(len(P)):
                v = gpUtils._nll(P[i], gp, y, None)
                mu, var = gp.predict(y, T[i:i + 1], return_var=True)
                m8 = gp.predict(y, T[:8], return_cov=False)
                out.append((v, mu[0], var[0]) + tuple(m8))
        return np.array(out)
    want = [run(*c) for c in cases]
    got = [None, None]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def work(k):
        with torch.cuda.stream(streams[k]):
            got[k] = run(*cases[k])
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert np.array_equal(got[k], want[k])
    lib = _lib.load()
    for st in streams:                       # the streams' scratch / mailboxes are released explicitly
        st.synchronize()
        assert lib.apgp_release_scratch(ctypes.c_void_p(st.cuda_stream)) >= 1


def test_optimizegp_batched_restarts_equal_sequential():
    """gpUtils.optimizeGP with its restarts evaluated in lock-step on the device returns
    exactly the hyper-parameters of the reference's sequential loop (gpUtils.py:223-254)."""
    import time
    from approxposterior_amd import gpUtils, likelihood as lh
    out = {}
    for mode in (False, "always", True):       # ("always": lock-step whatever the size; True: by size -- sequential at N = 60)
        np.random.seed(57)
        theta = lh.rosenbrockSample(60)
        y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
        gp = gpUtils.defaultGP(theta, y, white_noise=-12)
        t0 = time.time()
        with np.errstate(all="ignore"):
            gp = gpUtils.optimizeGP(gp, theta, y, seed=57, nGPRestarts=4, batchRestarts=mode)
        out[mode] = (np.array(gp.get_parameter_vector()), time.time() - t0)
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][0], out["always"][0])
    print("optimizeGP 4 restarts N=60: sequential %.2f s, batched %.2f s" % (out[False][1], out["always"][1]))


@pytest.mark.parametrize("n,d,amp,method", [(90, 2, True, "powell"), (400, 3, False, "powell"), (1152, 8, False, "powell"),
                                            (90, 2, True, "nelder-mead"), (700, 4, False, "nelder-mead")])
def test_powell_lookahead_same_optimum_fewer_device_rounds(n, d, amp, method):
    """gpUtils._nll's look-ahead (round 6): when SciPy's Powell asks for f(1) of a line search, the abscissae it asks for
    next ride along in ONE batched device call -- side-by-side persistent factorisations above n = 128, a workgroup per
    matrix below -- and SciPy sees the same values in the same order: the optimum of optimizeGP (gpUtils.py:184-257) is the
    same in every bit with and without, in fewer device rounds.  The same for ``method="nelder-mead"`` (the points behind a
    reflection, the initial simplex, a shrink: gpUtils._nelderMeadAhead)."""
    from approxposterior_amd import gpUtils, _lib
    lib = _lib.load()
    rs = np.random.RandomState(n)
    X = rs.uniform(-5, 5, size=(n, d))
    y = np.sin(X).sum(axis=1) + 0.1 * rs.randn(n)
    got = {}
    for ahead in (0, None):
        np.random.seed(4)
        gp = gpUtils.defaultGP(X, y, fitAmp=amp)
        gp.lookahead = ahead
        seen = []
        inner = gpUtils._nll

        def spy(p, *args):
            v = inner(p, *args)
            seen.append((np.array(p).tobytes(), v))
            return v
        gpUtils._nll = spy
        try:
            before = lib.apgp_nll_side_batches()
            with np.errstate(all="ignore"):
                gp = gpUtils.optimizeGP(gp, X, y, seed=1, nGPRestarts=1, method=method,
                                        options={"maxiter": 3} if method == "powell" else {"maxiter": 120, "adaptive": True})
            side = lib.apgp_nll_side_batches() - before
        finally:
            gpUtils._nll = inner
        got[ahead] = (np.array(gp.get_parameter_vector()), seen, side, gp.log_likelihood(y))
    assert np.array_equal(got[0][0], got[None][0])                  # the optimum: every bit
    assert got[0][1] == got[None][1] and len(got[0][1]) > 50        # every point and value SciPy saw, in order
    assert got[0][3] == got[None][3]
    assert got[0][2] == 0
    if n > 128:
        assert got[None][2] >= len(got[None][1]) // 14 or lib.apgp_potrf_backoff_skips() > 0     # ~ one batch per line search
