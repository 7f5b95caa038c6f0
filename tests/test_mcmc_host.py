"""CPU: the build-side ensemble sampler and the mcmcUtils mirror (SURVEY.md
section 8f rank 1-2), tested the way the reference tests its emcee-based
counterparts: test_Burnin.py:83-90 (iburn, ithin ~ [67, 15], rtol 0.1),
test_MCSE.py:30-35 (AR(1) MCSE 0.00494), plus sampler sanity (moments of a known
Gaussian, vectorised == scalar path) and the outer loop on the injected oracle GP
(test_MAP.py:51-65, a short ApproxPosterior.run)."""
import numpy as np
import pytest

import george_oracle as go
from approxposterior_amd import approx, likelihood as lh, mcmc, mcmcUtils


def test_batch_means_mcse_ar1():
    np.random.seed(57)
    num = int(1.0e5)
    samples = np.zeros(num)
    eps = np.random.randn(num)
    for ii in range(1, num):
        samples[ii] = 0.4 * samples[ii - 1] + eps[ii]
    mcse = mcmcUtils.batchMeansMCSE(samples)
    assert np.allclose(0.00494, mcse, atol=2.5e-3)
    two = mcmcUtils.batchMeansMCSE(np.stack([samples, 2 * samples], axis=1))
    assert two.shape == (2,) and np.isclose(two[1], 2 * two[0])


def _line_problem():
    np.random.seed(42)
    mTrue, bTrue, N, obserr = -0.9594, 4.294, 50, 0.5
    x = np.sort(10 * np.random.rand(N))
    obs = mTrue * x + bTrue + obserr * np.random.randn(N)

    def logprob(theta):
        m, b = theta
        if not (-5.0 < m < 0.5 and 0.0 < b < 10.0):
            return -np.inf
        return -0.5 * np.sum((obs - (m * x + b)) ** 2 / obserr ** 2)

    def logprob_vec(thetas):
        m, b = thetas[:, 0], thetas[:, 1]
        ok = (m > -5.0) & (m < 0.5) & (b > 0.0) & (b < 10.0)
        r = obs[None, :] - (m[:, None] * x[None, :] + b[:, None])
        return np.where(ok, -0.5 * np.sum(r ** 2, axis=1) / obserr ** 2, -np.inf)
    return logprob, logprob_vec


def test_estimate_burnin_line_fit():
    """test_Burnin.py: 32 walkers x 5000 steps on the line-fit posterior."""
    logprob, logprob_vec = _line_problem()
    p0 = np.random.randn(32, 2)
    sampler = mcmc.EnsembleSampler(32, 2, logprob_vec, vectorize=True, seed=42)
    sampler.run_mcmc(p0, 5000)
    iburn, ithin = mcmcUtils.estimateBurnin(sampler, estBurnin=True, thinChains=True)
    assert np.allclose([67, 15], [iburn, ithin], rtol=1.5e-1), (iburn, ithin)
    assert 0.4 < sampler.acceptance_fraction.mean() < 0.9
    chain = sampler.get_chain(discard=iburn, thin=ithin, flat=True)
    assert np.allclose(chain.mean(axis=0), [-0.9594, 4.294], atol=0.15)
    assert mcmcUtils.estimateBurnin(sampler, estBurnin=False, thinChains=False) == (0, 1)


def test_vectorised_and_scalar_paths_agree():
    logprob, logprob_vec = _line_problem()
    p0 = np.random.RandomState(0).randn(8, 2) * 0.1 + [-1.0, 4.0]
    a = mcmc.EnsembleSampler(8, 2, logprob, seed=3)
    b = mcmc.EnsembleSampler(8, 2, logprob_vec, vectorize=True, seed=3)
    a.run_mcmc(p0, 50)
    b.run_mcmc(p0, 50)
    assert np.allclose(a.get_chain(), b.get_chain()) and np.allclose(a.get_log_prob(), b.get_log_prob())
    assert a.get_chain().shape == (50, 8, 2) and a.get_chain(flat=True, thin=5).shape == (80, 2)
    with pytest.raises(ValueError):
        mcmc.EnsembleSampler(7, 2, logprob)
    with pytest.raises(ValueError):
        mcmc.EnsembleSampler(2, 2, logprob)
    with pytest.raises(ValueError):
        a.run_mcmc(np.full((8, 2), np.nan), 1)


def test_gaussian_moments_and_blobs():
    cov = np.array([[2.0, 0.6], [0.6, 0.5]])
    icov = np.linalg.inv(cov)

    def lp(t):
        return -0.5 * np.einsum("ij,jk,ik->i", t, icov, t), np.zeros(len(t))
    s = mcmc.EnsembleSampler(40, 2, lp, vectorize=True, seed=1)
    s.run_mcmc(np.random.RandomState(2).randn(40, 2), 4000)
    flat = s.get_chain(discard=500, flat=True)
    assert np.allclose(flat.mean(axis=0), 0.0, atol=0.08)
    assert np.allclose(np.cov(flat.T), cov, rtol=0.12, atol=0.05)
    assert s.get_blobs().shape == (4000, 40)
    tau = s.get_autocorr_time(tol=0)
    assert tau.shape == (2,) and np.all(tau > 1)
    with pytest.raises(mcmc.AutocorrError):
        mcmc.integrated_time(s.get_chain()[:60], tol=50)


class _AP(object):
    ndim = 2

    def _gpll(self, t):
        return 0.0, 0.0

    @staticmethod
    def priorSample(n):
        return np.zeros((n, 2))


def test_validate_mcmc_kwargs():
    ap = _AP()
    skw, mkw = mcmcUtils.validateMCMCKwargs(ap, None, None)
    assert skw["nwalkers"] == 40 and skw["ndim"] == 2 and skw["log_prob_fn"] == ap._gpll
    assert mkw["iterations"] == 10000 and mkw["initial_state"].shape == (40, 2)
    skw, mkw = mcmcUtils.validateMCMCKwargs(ap, {"nwalkers": 10, "ndim": 7, "backend": 1,
                                                 "log_prob_fn": len}, {"iterations": 5})
    assert skw == {"nwalkers": 10, "ndim": 2, "log_prob_fn": ap._gpll}
    assert mkw["iterations"] == 5 and mkw["initial_state"].shape == (10, 2)


def _oracle_gp(theta, y, fit_amp):
    ndim = theta.shape[-1]
    k = go.ExpSquaredKernel(metric=np.fabs(np.random.randn(ndim)), ndim=ndim)
    if fit_amp:
        k = np.var(y) * k
    gp = go.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(theta)
    return gp


def test_find_map_sphere_on_oracle_gp(tmp_path, monkeypatch):
    """test_MAP.py:51-65 with the oracle GP injected (host logic only)."""
    monkeypatch.chdir(tmp_path)
    from approxposterior_amd import approx as ap_mod
    monkeypatch.setattr(ap_mod, "george", go)     # findNextPoint re-creates the GP through this name
    np.random.seed(57)
    theta = np.array(lh.sphereSample(20))
    y = np.array([lh.sphereLnlike(t) + lh.sphereLnprior(t) for t in theta])
    gp = _oracle_gp(theta, y, True)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.sphereLnprior,
                                lnlike=lh.sphereLnlike, priorSample=lh.sphereSample,
                                bounds=[(-5, 5), (-5, 5)], algorithm="jones")
    with np.errstate(all="ignore"):
        ap.optGP(seed=57, method="powell", nGPRestarts=3)
        ap.findNextPoint(numNewPoints=5, nGPRestarts=3, cache=False, verbose=False)
        testMAP, testVal = ap.findMAP(nRestarts=15)
    assert np.allclose([0.0, 0.0], testMAP, atol=1.0e-3)
    assert np.allclose(0.0, testVal, atol=1.0e-3)


def test_run_loop_on_oracle_gp(tmp_path, monkeypatch):
    """A short ApproxPosterior.run (test_APRun.py shape, fewer iterations): the
    training set grows by m per iteration, caches are written, the sampler and
    burn-in bookkeeping behave."""
    monkeypatch.chdir(tmp_path)
    from approxposterior_amd import approx as ap_mod
    monkeypatch.setattr(ap_mod, "george", go)
    np.random.seed(57)
    theta = np.array(lh.rosenbrockSample(30))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    gp = _oracle_gp(theta, y, False)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                bounds=[(-5, 5), (-5, 5)], algorithm="bape")
    with np.errstate(all="ignore"):
        ap.run(m=3, nmax=2, estBurnin=True, nGPRestarts=1, mcmcKwargs={"iterations": 300},
               samplerKwargs={"nwalkers": 10}, cache=True, verbose=False, thinChains=False,
               onlyLastMCMC=False, convergenceCheck=True, kmax=5, runName="t", timing=True)
    assert ap.theta.shape == (36, 2) and ap.y.shape == (36,)
    assert len(ap.iburns) == 2 and len(ap.ithins) == 2 and ap.ithins == [1, 1]
    assert ap.sampler.get_chain().shape == (300, 10, 2)
    assert len(ap.marginalMeans) == 2 and len(ap.trainingTime) == 2
    for f in ("tAPFModelCache.npz", "tAPGP.npz", "t0.npz", "t1.npz", "tAPTiming.npz", "tConvergenceCache.npz"):
        assert (tmp_path / f).exists(), f
    d = np.load(tmp_path / "tAPFModelCache.npz")
    assert d["theta"].shape == (36, 2)
    with pytest.raises(RuntimeError):
        ap.run(m=1, nmax=1, convergenceCheck=True, onlyLastMCMC=True, initGPOpt=False, cache=False,
               verbose=False)
