"""CPU: host-side logic of the package (parameter-vector protocol, kernel
objects, utilities' control flow, minimizeObjective, ApproxPosterior glue)
exercised through the duck-typed boundary with the ORACLE GP injected as ``gp=``
-- exactly how the reference is injected a george.GP.  No HIP compute runs here;
the product GP must refuse to compute without a GPU."""
import json
import os

import numpy as np
import pytest

import george_oracle as go
from approxposterior_amd import _lib, approx, gp as agp, gpUtils, likelihood as lh, utility as ut


def rosen_set(m0, corners=False):
    theta = np.array(lh.rosenbrockSample(m0))
    if corners:
        theta = np.array(list(theta) + [[-5, 5], [5, 5]])
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    return theta, y


def oracle_default_gp(theta, y, fit_amp):
    ndim = theta.shape[-1]
    metric = np.fabs(np.random.randn(ndim))
    k = go.ExpSquaredKernel(metric=metric, ndim=ndim)
    if fit_amp:
        k = np.var(y) * k
    gp = go.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(theta)
    return gp


def test_parameter_vector_protocol_matches_george_names():
    for amp in (True, False):
        k = agp.ExpSquaredKernel(metric=[0.5, 2.0], ndim=2)
        if amp:
            k = 7.0 * k
        gp = agp.GP(kernel=k, fit_mean=True, mean=-3.0, white_noise=-12, fit_white_noise=False)
        names = gp.get_parameter_names()
        if amp:
            assert names == ("mean:value", "kernel:k1:log_constant",
                             "kernel:k2:metric:log_M_0_0", "kernel:k2:metric:log_M_1_1")
            assert np.allclose(gp.get_parameter_vector(), [-3.0, np.log(7.0 / 2), np.log(0.5), np.log(2.0)])
        else:
            assert names == ("mean:value", "kernel:metric:log_M_0_0", "kernel:metric:log_M_1_1")
        p = gp.get_parameter_vector() + 0.25
        gp.set_parameter_vector(p)
        assert np.allclose(gp.get_parameter_vector(), p) and len(gp) == len(p)
        assert not gp.computed
        amp_v, logM = agp._flatten_kernel(gp.kernel)
        assert np.isclose(amp_v, 2 * np.exp(p[1]) if amp else 1.0)
        assert np.allclose(logM, p[-2:])
        with pytest.raises(ValueError):
            gp.set_parameter_vector(p[:-1])
    assert isinstance(gp.mean, agp.ConstantModel) and gp.white_noise.value == -12.0


def test_product_gp_refuses_to_compute_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    gp = agp.GP(kernel=agp.ExpSquaredKernel([1.0], ndim=1), fit_mean=True, mean=0.0,
                white_noise=-12, fit_white_noise=False)
    with pytest.raises(_lib.ApgpError):
        gp.compute(np.linspace(0, 1, 5))


def test_likelihood_functions_match_reference_tests():
    # test_TestFns.py:22-43 of the reference: values at fixed points
    assert np.isclose(lh.rosenbrockLnlike(np.array([1.0, 1.0])), 0.0)
    assert np.isclose(lh.rosenbrockLnlike(np.array([0.0, 0.0, 0.0, 0.0, 0.0])), -0.04)
    assert lh.rosenbrockLnprior(np.array([5.1, 0.0])) == -np.inf
    assert lh.rosenbrockLnprob(np.array([0.0, 6.0])) == -np.inf
    assert np.isclose(lh.sphereLnlike([1.0, 2.0]), -5.0) and lh.sphereLnprior([0, 2.1]) == -np.inf
    assert np.isclose(lh.testBOFn(0.0), 0.0) and lh.testBOFnLnPrior(2.5) == -np.inf
    np.random.seed(3)
    a = lh.rosenbrockSample(4)
    np.random.seed(3)
    assert np.array_equal(a, np.random.uniform(-5, 5, size=(4, 2)))


def test_utilities_and_logsubexp_on_oracle_gp(golden_dir):
    pins = json.load(open(os.path.join(golden_dir, "pins.json")))
    c = pins["reference_test_constants"]
    tt = np.array(c["theta_test"])
    assert ut.logsubexp(1.0, 2.0) == -np.inf
    assert np.isclose(ut.logsubexp(2.0, 1.0), np.log(np.exp(2.0) - np.exp(1.0)))
    for amp, keys in ((True, ("test_GPUtil.py:50", "test_GPUtil.py:56", "test_GPUtil.py:62")),
                      (False, ("test_GPUtil.py:101", "test_GPUtil.py:107", "test_GPUtil.py:113"))):
        np.random.seed(57)
        theta, y = rosen_set(20)
        gp = oracle_default_gp(theta, y, amp)
        vals = (ut.AGPUtility(tt, y, gp, lh.rosenbrockLnprior),
                ut.BAPEUtility(tt, y, gp, lh.rosenbrockLnprior),
                ut.JonesUtility(tt, y, gp, lh.rosenbrockLnprior))
        for v, k in zip(vals, keys):
            assert np.allclose(v, c[k], rtol=1e-4)
        assert ut.AGPUtility(np.array([9.0, 0.0]), y, gp, lh.rosenbrockLnprior) == np.inf
    gp.kernel.dirty = True
    with pytest.raises(RuntimeError):
        ut.BAPEUtility(tt, y, gp, lh.rosenbrockLnprior)
    assert ut.utilityKind(ut.JonesUtility) == "jones" and ut.utilityKind("AGP") == "agp"
    with pytest.raises(ValueError):
        ut.utilityKind(len)


def test_optimizegp_and_findnextpoint_reproduce_reference_goldens(golden_dir):
    """test_OptimizeGP.py:91 and test_findNewPoint.py:107 (fitAmp=False) through
    this package's gpUtils / ApproxPosterior with the oracle GP injected."""
    pins = json.load(open(os.path.join(golden_dir, "pins.json")))
    c = pins["reference_test_constants"]
    np.random.seed(57)
    theta, y = rosen_set(50)
    gp = oracle_default_gp(theta, y, False)
    with np.errstate(all="ignore"):
        gp = gpUtils.optimizeGP(gp, theta, y, seed=57, nGPRestarts=5)
    assert np.allclose(gp.get_parameter_vector()[1:], c["test_OptimizeGP.py:91"], rtol=1e-2)
    assert np.allclose(gp.get_parameter_vector(), pins["harness_replay"]["optgp_noamp"]["p"], rtol=1e-6)

    np.random.seed(57)
    theta, y = rosen_set(50, corners=True)
    gp = oracle_default_gp(theta, y, False)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                bounds=((-5, 5), (-5, 5)), algorithm="bape")
    with np.errstate(all="ignore"):
        thetaT = ap.findNextPoint(computeLnLike=False, bounds=((-5, 5), (-5, 5)), seed=57)
    assert np.allclose(thetaT, c["test_findNewPoint.py:107"], rtol=1e-3)
    assert np.allclose(thetaT, pins["harness_replay"]["findnext_noamp"]["thetaT"], rtol=1e-6)


def test_gpll_guards_with_oracle_gp(golden_dir):
    pins = json.load(open(os.path.join(golden_dir, "pins.json")))
    np.random.seed(57)
    theta, y = rosen_set(50, corners=True)
    gp = oracle_default_gp(theta, y, False)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                bounds=((-5, 5), (-5, 5)), algorithm="bape")
    thetas = []
    for case in pins["harness_replay"]["gpll_noamp"]:
        t = np.array([float(v) for v in case["theta_repr"]])
        thetas.append(t)
        want = [float(v) for v in case["out"]]
        with np.errstate(all="ignore"):
            got = [float(np.ravel(v)[0]) for v in ap._gpll(t)]
        for a_, b_ in zip(got, want):
            assert (np.isnan(a_) and np.isnan(b_)) or a_ == b_ or np.isclose(a_, b_, rtol=1e-10)
    # vectorised form agrees row by row
    with np.errstate(all="ignore"):
        lp, blob = ap._gpllBatch(np.array(thetas))
    for i, case in enumerate(pins["harness_replay"]["gpll_noamp"]):
        want = [float(v) for v in case["out"]]
        assert (np.isnan(blob[i]) and np.isnan(want[1])) or blob[i] == want[1]
        assert lp[i] == want[0] or np.isclose(lp[i], want[0], rtol=1e-10)


def test_vectorised_prior_twins_equal_the_scalar_priors():
    """``prior.batch(T)`` (likelihood.py; what ``_gpllBatch`` calls once per walker ensemble) == ``[prior(t) for t in T]``
    on random, boundary, NaN and infinite rows -- and ``_gpllBatch`` returns the same with a prior that has no twin."""
    rs = np.random.RandomState(4)
    for f, D, edge in ((lh.rosenbrockLnprior, 2, 5.0), (lh.sphereLnprior, 2, 2.0), (lh.rosenbrockLnprior, 8, 5.0),
                       (lh.testBOFnLnPrior, 1, 2.0)):
        T = rs.uniform(-1.3 * edge, 1.3 * edge, size=(400, D))
        T[0, 0] = edge; T[1, 0] = -edge; T[2, -1] = np.nextafter(edge, np.inf); T[3, 0] = np.nan
        T[4, -1] = np.inf; T[5, 0] = -np.inf; T[6] = 0.0; T[7, 0] = -1.0; T[8, 0] = np.nextafter(-1.0, -np.inf)
        with np.errstate(all="ignore"):
            want = np.array([f(t) for t in T], dtype=float)
            got = np.asarray(f.batch(T), dtype=float)
        assert got.shape == want.shape and np.array_equal(got, want)
    np.random.seed(57)
    theta, y = rosen_set(50, corners=True)
    gp = oracle_default_gp(theta, y, False)
    mk = lambda prior: approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=prior, lnlike=lh.rosenbrockLnlike,   # noqa: E731
                                              priorSample=lh.rosenbrockSample, bounds=((-5, 5), (-5, 5)), algorithm="bape")
    a1, a2 = mk(lh.rosenbrockLnprior), mk(lambda t: lh.rosenbrockLnprior(t))
    assert getattr(a2._lnprior, "batch", None) is None
    W = rs.uniform(-6, 6, size=(64, 2))
    for pts in (W, W[:10], np.vstack([W[:5], [[np.nan, np.nan]], [[np.nan, 1.0]], [[np.inf, 0.0]]])):
        with np.errstate(all="ignore"):
            l1, b1 = a1._gpllBatch(pts)
            l2, b2 = a2._gpllBatch(pts)
        assert np.array_equal(l1, l2) and np.array_equal(b1, b2, equal_nan=True)


def test_approxposterior_input_validation():
    th = np.zeros((3, 2)); y = np.zeros(3)
    kw = dict(lnprior=lh.rosenbrockLnprior, lnlike=lh.rosenbrockLnlike,
              priorSample=lh.rosenbrockSample, gp=object())
    with pytest.raises(ValueError):
        approx.ApproxPosterior(theta=None, y=y, bounds=((-5, 5),) * 2, **kw)
    with pytest.raises(ValueError):
        approx.ApproxPosterior(theta=th, y=np.array([0, np.nan, 0]), bounds=((-5, 5),) * 2, **kw)
    with pytest.raises(ValueError):
        approx.ApproxPosterior(theta=th, y=y, bounds=((-5, 5),), **kw)
    with pytest.raises(ValueError):
        approx.ApproxPosterior(theta=th, y=y, bounds=((-5, 5),) * 2, algorithm="nope", **kw)
    ap = approx.ApproxPosterior(theta=th, y=y, bounds=((-5, 5),) * 2, algorithm="Alternate", **kw)
    assert ap.utility is ut.AGPUtility and ap.ndim == 2


def test_on_device_sampler_requires_the_box_prior():
    """runMCMC(onDevice=True) samples mu(theta) under the box prior only; _gpll adds lnprior(theta)
    (/root/reference/approxposterior/approx.py:167-188).  Anything but 'constant inside self.bounds, -inf outside'
    must be refused before the device is touched (the check itself needs no GPU)."""
    th = np.zeros((3, 2)); y = np.zeros(3)
    kw = dict(theta=th, y=y, lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample, gp=object(), bounds=((-5, 5),) * 2)
    state = np.random.get_state()[1].copy()
    approx.ApproxPosterior(lnprior=lh.rosenbrockLnprior, **kw)._requireBoxPrior()            # the box: accepted
    assert np.array_equal(np.random.get_state()[1], state)                                   # (NumPy's global stream untouched)
    approx.ApproxPosterior(lnprior=lambda t: lh.rosenbrockLnprior(t) + 3.5, **kw)._requireBoxPrior()   # any constant
    gauss = lambda t: -0.5 * float(np.sum(np.asarray(t) ** 2)) if np.all(np.abs(t) <= 5) else -np.inf
    wide = lambda t: 0.0 if np.all(np.abs(t) <= 6) else -np.inf                              # support wider than bounds
    narrow = lambda t: 0.0 if np.all(np.abs(t) <= 4.999) else -np.inf                        # narrower: corners excluded
    for bad in (gauss, wide, narrow, lambda t: 0.0):
        with pytest.raises(ValueError):
            approx.ApproxPosterior(lnprior=bad, **kw)._requireBoxPrior()
    ap = approx.ApproxPosterior(lnprior=gauss, **kw)
    with pytest.raises(ValueError):
        ap.runMCMC(onDevice=True, cache=False, mcmcKwargs={"iterations": 10, "initial_state": np.zeros((4, 2))},
                   samplerKwargs={"nwalkers": 4})


def test_linear_kernel_sum_protocol_and_flattening():
    """defaultGP(order=...) kernel tree (gpUtils.py:167-173): same names / order / values
    as the oracle's george restatement, and the evaluated form handed to the C ABI."""
    def make(mod, fit_amp):
        k = mod.ExpSquaredKernel(np.array([1.5, 0.7, 2.0]), ndim=3)
        if fit_amp:
            k = 2.0 * k
        return k + 0.3 * mod.kernels.LinearKernel(log_gamma2=0.4, order=2, bounds=None, ndim=3)
    for fit_amp in (False, True):
        ko, ka = make(go, fit_amp), make(agp, fit_amp)
        assert ka.get_parameter_names() == ko.get_parameter_names()
        assert np.allclose(ka.get_parameter_vector(), ko.get_parameter_vector())
        amp, log_M, lin_coef, lin_order = agp._flatten_kernel(ka, with_linear=True)
        assert amp == pytest.approx(2.0 if fit_amp else 1.0) and lin_order == 2
        assert lin_coef == pytest.approx(0.3 * np.exp(-0.4))        # ndim*exp(log(c/ndim)) / gamma^2
        assert np.allclose(log_M, np.log([1.5, 0.7, 2.0]))
        p = ka.get_parameter_vector() + 0.1
        ka.set_parameter_vector(p); ko.set_parameter_vector(p)
        assert np.allclose(ka.get_parameter_vector(), ko.get_parameter_vector()) and ka.dirty
    assert agp._flatten_kernel(agp.ExpSquaredKernel([1.0], ndim=1), with_linear=True)[2:] == (0.0, 0)
    with pytest.raises(NotImplementedError):
        agp.kernels.LinearKernel(log_gamma2=0.0, order=1.5, ndim=2)
    with pytest.raises(NotImplementedError):
        agp._flatten_kernel(agp.kernels.LinearKernel(log_gamma2=0.0, order=1, ndim=2))   # no SE term


class _BatchingOracleGP(object):
    """The oracle GP plus an ``nll_batch`` that loops over ``gpUtils._nll`` (what the device
    batch computes, one vector at a time) and records the batch sizes it was handed."""

    def __init__(self, gp):
        self._gp = gp
        self.batches = []

    def __getattr__(self, name):
        return getattr(self._gp, name)

    def nll_batch(self, P, y):
        self.batches.append(len(P))
        saved = self._gp.get_parameter_vector()
        out = np.array([gpUtils._nll(np.array(p), self._gp, y, None) for p in P])
        self._gp.set_parameter_vector(saved)
        return out


def test_lockstep_restarts_follow_the_sequential_trajectories():
    """optimizeGP's restarts run concurrently with their _nll calls
    batched.  Start points, every restart's solution and the selected optimum must be those
    of the reference's sequential loop (gpUtils.py:223-254), bit for bit."""
    np.random.seed(57)
    theta, y = rosen_set(30)
    with np.errstate(all="ignore"):
        np.random.seed(11)
        gp_a = oracle_default_gp(theta, y, False)
        np.random.seed(3)
        seq = gpUtils.optimizeGP(gp_a, theta, y, nGPRestarts=3, batchRestarts=False)
        p_seq = np.array(seq.get_parameter_vector())
        np.random.seed(11)
        gp_b = _BatchingOracleGP(oracle_default_gp(theta, y, False))
        np.random.seed(3)
        bat = gpUtils.optimizeGP(gp_b, theta, y, nGPRestarts=3, batchRestarts="always")
        p_bat = np.array(bat.get_parameter_vector())
    assert np.array_equal(p_seq, p_bat)
    assert max(gp_b.batches) == 3 and min(gp_b.batches) >= 1      # batched while >1 restart is running
    assert gp_b.batches[-1] == 3                                   # the final mll batch
    # by size (the default): below gpUtils._LOCKSTEP_MIN_N training points the restarts run one after the other -- a
    # single evaluation is one fused launch there and the threads' rendezvous costs more than a batch saves -- same result
    with np.errstate(all="ignore"):
        np.random.seed(11)
        gp_c = _BatchingOracleGP(oracle_default_gp(theta, y, False))
        np.random.seed(3)
        dflt = gpUtils.optimizeGP(gp_c, theta, y, nGPRestarts=3)
    assert len(y) < gpUtils._LOCKSTEP_MIN_N and gp_c.batches == []
    assert np.array_equal(np.array(dflt.get_parameter_vector()), p_seq)


def test_lockstep_evaluator_edge_cases():
    """Workers with different numbers of evaluations retire without deadlock, the prior gate
    answers without a device call, and a failing batch raises in the caller."""
    calls = []

    class Quad(object):
        def nll_batch(self, P, y):
            calls.append(len(P))
            return np.array([float(np.sum((np.asarray(p) - 1.0) ** 2)) for p in P])

    x0s = [np.full(3, 1.0 + 1e-9), np.full(3, 70.0), np.array([2.0, -3.0, 0.5])]
    sols = gpUtils._minimizeLockStep(Quad(), None, x0s, "nelder-mead", {"xatol": 1e-8, "fatol": 1e-12}, None)
    for s_, x0 in zip(sols, x0s):
        assert np.allclose(s_, 1.0, atol=1e-3) and len(s_) == len(x0)
    assert 3 in calls and min(calls) < 3          # the restarts finish at different times
    # prior that forbids everything: no batch is ever evaluated
    calls[:] = []
    gpUtils._minimizeLockStep(Quad(), None, [np.zeros(2), np.ones(2)], "powell", {"maxiter": 2},
                              lambda p: -np.inf)
    assert calls == []

    class Broken(object):
        def nll_batch(self, P, y):
            raise RuntimeError("device lost")

    with pytest.raises(RuntimeError):
        gpUtils._minimizeLockStep(Broken(), None, [np.zeros(2), np.ones(2)], "powell", None, None)


# ---------------------------------------------------------------------------------------------------------------
# Powell look-ahead (gpUtils._powellAhead): the points SciPy is going to ask for next are evaluated with the one it
# asks for now; SciPy must see the same values in the same order (gpUtils.py:238).
# ---------------------------------------------------------------------------------------------------------------

def _lookahead_stub(width):
    import george_oracle as go
    from approxposterior_amd import gpUtils

    class Stub(go.GP):
        """The oracle GP + the three members gpUtils._nll's memo / look-ahead use."""
        _nllMemo = None
        rounds = 0            # device rounds: single evaluations + batches
        batched = 0           # points evaluated in batches
        fit_white_noise = False

        def lookahead_width(self):
            return width

        def log_likelihood(self, y, quiet=False):
            Stub.rounds += 1
            return super().log_likelihood(y, quiet=quiet)

        def nll_batch(self, P, y):
            Stub.rounds += 1
            Stub.batched += len(P)
            saved = self.get_parameter_vector()
            out = []
            for q in P:
                self.set_parameter_vector(q)
                ll = go.GP.log_likelihood(self, y, quiet=True)
                out.append(-ll if np.isfinite(ll) else np.inf)
            self.set_parameter_vector(saved)
            return np.array(out)

        # (the oracle's recompute() goes through compute(): the memo is not dropped there, as the HIP GP's is not by
        # its own refactorisations -- only a user's compute() on a new training set does that)

    return Stub, gpUtils, go


def test_powell_lookahead_serves_scipy_the_same_values_in_fewer_device_rounds(monkeypatch):
    from scipy.optimize import minimize, rosen
    results = {}
    for width in (0, 2, 4):
        Stub, gpUtils, go = _lookahead_stub(width)
        monkeypatch.setattr(gpUtils, "george", go)           # _memoKey compares kernel types with the module's names
        rs = np.random.RandomState(3)
        X = rs.uniform(-5, 5, size=(60, 3))
        y = np.array([-rosen(x) / 100.0 for x in X])
        gp = Stub(kernel=go.ExpSquaredKernel(np.fabs(rs.randn(3)) + 0.5, ndim=3), fit_mean=True, mean=np.median(y),
                  white_noise=-12, fit_white_noise=False)
        gp.compute(X)
        seen = []

        def fn(p, *args):
            v = gpUtils._nll(p, *args)
            seen.append((np.array(p).tobytes(), v))
            return v
        x0 = [np.median(y)] + list(rs.randn(3))
        with np.errstate(all="ignore"):
            res = minimize(fn, x0, args=(gp, y, gpUtils.defaultHyperPrior), method="powell",
                           options={"maxiter": 4})
        results[width] = (res["x"].tobytes(), res["nfev"], seen, Stub.rounds, Stub.batched)
    base = results[0]
    assert base[4] == 0
    for width in (2, 4):
        got = results[width]
        assert got[0] == base[0] and got[1] == base[1]       # the same optimum, the same number of SciPy evaluations
        assert got[2] == base[2]                             # every point and every value SciPy saw, in order, bit for bit
        assert got[4] > 0
    # two line searches in three stop bracketing after the third point: width 4 saves two rounds there, width 2 one
    assert results[4][3] < results[2][3] < base[3]
    assert results[4][3] <= 0.88 * base[3], (results[4][3], base[3])


def test_powell_lookahead_abscissae_are_scipys():
    """The look-ahead recomputes SciPy's abscissae from SciPy's frames; pin them for the SciPy in this image: at f(1) of
    a line search the third bracket point is always among the guesses and most often the two evaluations after it too; at
    Brent's first step the second is one of the two guesses; a tolerance step's second guess is f(1) of the NEXT search
    whenever the search ends where the step said it would."""
    import inspect
    from scipy.optimize import minimize, _optimize
    from approxposterior_amd import gpUtils
    asked, guessed = [], []

    def g(p):        # same depth as _nll -> _powellAhead: objective -> helper
        asked.append(np.array(p))
        got = gpUtils._powellAhead(8, explain=True)
        if got is not None:
            guessed.append((len(asked) - 1, got[0], [np.array(q) for q in got[1]]))
        return float(np.sum((p - np.array([0.3, -1.2, 0.8])) ** 2) + 0.1 * np.sum(p ** 4) + np.sin(3.0 * p[0]))
    minimize(g, np.array([2.0, 1.5, -0.7]), method="powell", options={"maxiter": 6})

    def among(a, pts):
        return any(np.array_equal(a, q) for q in pts)
    starts = [(at, pts) for at, kind, pts in guessed if kind == "f(1)"]
    assert len(starts) >= 6, "no line search was recognised: SciPy's Powell has changed shape -- see gpUtils._powellAhead"
    two_more = 0
    for at, pts in starts:
        assert len(pts) == 8
        assert among(asked[at + 1], (pts[0], pts[3]))                     # -1.618034 or 2.618034
        nxt = asked[at + 2:at + 4]
        two_more += int(len(nxt) == 2 and all(among(a, pts) for a in nxt))   # the bracket closed: Brent's first two steps
    assert two_more >= len(starts) // 2, (two_more, len(starts))
    for at, kind, pts in guessed:
        if kind == "Brent's first step" and at + 1 < len(asked):
            assert len(pts) == 2 and among(asked[at + 1], pts)
    tol = [(at, pts) for at, kind, pts in guessed if kind == "tolerance step"]
    across = [(at, pts) for at, pts in tol if len(pts) == 5]             # ... + f(1), third point, two Brent steps of the next search
    hits = sum(1 for at, pts in across if any(among(a, pts[1:2]) for a in asked[at + 1:at + 4]))
    assert len(tol) >= 6 and len(across) >= 4 and hits >= len(across) // 2, (len(tol), len(across), hits)
    # the constant the cross-search look-ahead hard-codes is the one SciPy's bracket uses
    assert "_gold = %r" % gpUtils._GOLD in inspect.getsource(_optimize.bracket)


def test_more_than_max_dim_dimensions_is_an_explicit_error():
    """george has no dimension limit (gpUtils.py:150-161); this build has one (APGP_MAX_DIM = 32: the kernels keep a point's
    coordinates in registers).  A 33-dimensional training set must be refused with a ValueError that names the limit --
    before any device work -- by compute() and therefore by defaultGP."""
    from approxposterior_amd import gp as agp, gpUtils, _lib
    d = _lib.MAX_DIM + 1
    rs = np.random.RandomState(0)
    X, y = rs.uniform(-1, 1, size=(40, d)), rs.randn(40)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.ones(d), ndim=d), fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False)
    with pytest.raises(ValueError, match=r"at most 32 input dimensions.*got 33"):
        g.compute(X)
    np.random.seed(1)
    with pytest.raises(ValueError, match="APGP_MAX_DIM"):
        gpUtils.defaultGP(X, y)


def test_nelder_mead_lookahead_serves_scipy_the_same_values_in_fewer_device_rounds(monkeypatch):
    """The look-ahead for ``method="nelder-mead"`` (gpUtils._nelderMeadAhead): the initial simplex, the expanded / contracted
    points behind a reflection and the vertices of a shrink are functions of the simplex alone -- they ride along; SciPy sees
    the same points and values in the same order."""
    from scipy.optimize import minimize, rosen
    results = {}
    for width in (0, 3, 5):
        Stub, gpUtils, go = _lookahead_stub(width)
        monkeypatch.setattr(gpUtils, "george", go)
        rs = np.random.RandomState(5)
        X = rs.uniform(-5, 5, size=(50, 3))
        y = np.array([-rosen(x) / 100.0 for x in X])
        gp = Stub(kernel=go.ExpSquaredKernel(np.fabs(rs.randn(3)) + 0.5, ndim=3), fit_mean=True, mean=np.median(y),
                  white_noise=-12, fit_white_noise=False)
        gp.compute(X)
        seen = []

        def fn(p, *args):
            v = gpUtils._nll(p, *args)
            seen.append((np.array(p).tobytes(), v))
            return v
        x0 = [np.median(y)] + list(rs.randn(3))
        with np.errstate(all="ignore"):
            res = minimize(fn, x0, args=(gp, y, gpUtils.defaultHyperPrior), method="nelder-mead",
                           options={"maxiter": 60, "adaptive": True})
        results[width] = (res["x"].tobytes(), res["nfev"], seen, Stub.rounds, Stub.batched)
    base = results[0]
    for width in (3, 5):
        got = results[width]
        assert got[0] == base[0] and got[1] == base[1] and got[2] == base[2]
        assert got[4] > 0 and got[3] <= 0.8 * base[3], (got[3], base[3])
