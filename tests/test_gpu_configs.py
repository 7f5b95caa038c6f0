"""GPU (``-m gpu``): the BASELINE.json configurations as written, and the reference's own
known-answer constants reproduced ON THE HIP PATH:

  C2  Rosenbrock 2-D, N_train = 1024, 1e5 candidates, BAPE           (configs[1])
  C4  D = 8, N = 4096, 1e7 candidates in 8 shards with idx_offset     (configs[3], emulated
      on one GPU: shard r = rows [r M/8, (r+1) M/8) of the one NumPy seed-1 draw)
  C5  ApproxPosterior.run, D = 8, m0 = 512, m = 64, nmax = 10, default optGPEveryN = 1,
      64 walkers x 2e4 (configs[4] as written), and the device sampler separately at N = 1152
  a6  gpUtils.optimizeGP        vs test_OptimizeGP.py:91   (reference constant) + pins.json
  a12 ApproxPosterior.findNextPoint vs test_findNewPoint.py:107 (reference constant)
  f2  the 2-D Bayesian-optimisation test of the reference (test_2DBayesOpt.py:55-73)

Every comparison is HIP (through the C ABI) vs the NumPy oracle or vs constants the
reference's test-suite holds; tolerances are stated where they are used."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = 2.2e-16


def _mods():
    import george_oracle as go
    from approxposterior_amd import gp as agp
    return go, agp


def _synthetic(n, d, seed=0):
    """X ~ U[-5,5]^(n x d) (NumPy seed), y = -rosen(X)/100."""
    from scipy.optimize import rosen
    rs = np.random.RandomState(seed)
    X = rs.uniform(-5, 5, size=(n, d))
    return X, np.array([-rosen(x) / 100.0 for x in X])


def _pair(X, y, metric, d):
    go, agp = _mods()
    def make(mod):
        gp = mod.GP(kernel=mod.ExpSquaredKernel(np.full(d, metric), ndim=d), fit_mean=True,
                    mean=np.median(y), white_noise=-12, fit_white_noise=False)
        gp.compute(X)
        return gp
    return make(go), make(agp)


def _bape(mu, var):
    with np.errstate(all="ignore"):
        return -((2 * mu + var) + np.where(var <= 0, -np.inf, var + np.log(1 - np.exp(-var))))


def _agp(mu, var):
    with np.errstate(all="ignore"):
        return -(mu + 0.5 * np.log(2 * np.pi * np.e * var))


# --------------------------------------------------------------------------------------- C2
def test_c2_as_written_bape_1e5():
    """BASELINE.json configs[1]: Rosenbrock 2-D, N_train = 1024, 1e5 candidates, BAPE, fp64.
    Training set = the reference's own sampler / likelihood with seed 57 (likelihood.py:26-64).
    Every candidate's (mu, sigma^2, u) is compared with the oracle (chunked predict), and the
    arg-min with the oracle's.  Tolerance: 200 cond eps (module docstring of test_gpu_parity)."""
    from approxposterior_amd import likelihood as lh
    go, agp = _mods()
    np.random.seed(57)
    X = np.array(lh.rosenbrockSample(1024))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in X])
    gpo, gp = _pair(X, y, 2.0, 2)
    K = gpo.kernel.get_value(gpo._x)
    K[np.diag_indices_from(K)] += np.exp(-12.0)
    cond = np.linalg.cond(K)
    tol = max(1e-13, 200 * cond * EPS)
    M = 100000
    T = np.random.RandomState(1).uniform(-5, 5, size=(M, 2))
    bi, bu, u, mu, var = gp.acquire(y, T, "bape", bounds=[(-5, 5)] * 2, return_all=True)
    mo = np.empty(M); vo = np.empty(M)
    for c0 in range(0, M, 8192):
        mo[c0:c0 + 8192], vo[c0:c0 + 8192] = gpo.predict(y, T[c0:c0 + 8192], return_var=True)
    alpha = gpo._compute_alpha(y, False)
    assert np.abs(mu - mo).max() <= tol * np.abs(alpha).sum()
    assert np.abs(var - vo).max() <= tol
    uo = _bape(mo, vo)
    # BAPE amplifies the sigma^2 rounding by 1 + 1/(e^var - 1): compare where var is resolved
    sel = vo > 1e3 * tol
    du = 2 * tol * np.abs(alpha).sum() + tol * (1 + 1 / np.expm1(vo[sel]))
    assert np.all(np.abs(u[sel] - uo[sel]) <= 4 * du + 1e-12 * np.abs(uo[sel]))
    ri = int(np.argmin(np.where(np.isfinite(uo), uo, np.inf)))
    assert bu == u[bi] and bi == int(np.argmin(np.where(np.isfinite(u), u, np.inf)))
    assert bi == ri or abs(uo[bi] - uo[ri]) <= 1e-9 * max(1.0, abs(uo[ri]))


# --------------------------------------------------------------------------------------- C4
def test_c4_emulated_eight_shards_1e7():
    """BASELINE.json configs[3] on ONE GPU: the 1e7 x 8 candidate matrix of NumPy
    RandomState(1), rank r's shard = rows [r M/8, (r+1) M/8) swept with
    idx_offset = r M/8, winners combined with dist.combine_best (what the 16 B/rank all-gather
    feeds) -- must equal the unsharded sweep's (index, u) BIT FOR BIT; plus a 4,096-row
    subsample (and the winner's neighbourhood) against the oracle."""
    import torch
    from approxposterior_amd.dist import shard_bounds, combine_best
    N, D, M = 4096, 8, 10_000_000
    X, y = _synthetic(N, D)
    gpo, gp = _pair(X, y, 8.0, D)
    cands = np.random.RandomState(1).uniform(-5.0, 5.0, size=(M, D))
    bounds = [(-5, 5)] * D
    pairs = []
    for r in range(8):
        lo, hi = shard_bounds(M, 8, r)
        assert (lo, hi) == (r * M // 8, (r + 1) * M // 8)
        T = torch.from_numpy(cands[lo:hi]).cuda()
        sbi, sbu = gp.acquire(y, T, "agp", bounds=bounds, idx_offset=lo)
        assert lo <= sbi < hi
        pairs.append((sbu, sbi))
        del T
    best = combine_best(pairs)
    T = torch.from_numpy(cands).cuda()
    whole = gp.acquire(y, T, "agp", bounds=bounds)
    del T
    assert best == whole                     # bit-identical (index, u)
    bi, bu = whole
    # oracle: a 4,096-row subsample of the 1e7 plus the rows around the winner
    rs = np.random.RandomState(7)
    rows = np.unique(np.concatenate([rs.randint(0, M, size=4096), np.arange(max(0, bi - 8), min(M, bi + 8))]))
    mo, vo = gpo.predict(y, cands[rows], return_var=True)
    _, _, u, mu, var = gp.acquire(y, cands[rows], "agp", bounds=bounds, return_all=True)
    alpha = gpo._compute_alpha(y, False)
    tol = 1e-11                              # cond(K) ~ 1.5e3 at these hyper-parameters
    assert np.abs(mu - mo).max() <= tol * np.abs(alpha).sum()
    assert np.abs(var - vo).max() <= tol
    uo = _agp(mo, vo)
    assert np.abs(u - uo).max() <= 1e-9 * np.abs(uo).max()
    j = int(np.flatnonzero(rows == bi)[0])
    assert abs(bu - uo[j]) <= 1e-9 * abs(uo[j])
    assert bu <= uo.min() + 1e-9 * abs(uo.min())      # nothing in the subsample beats the winner


# --------------------------------------------------------------------------------------- C3
@pytest.mark.parametrize("mode", ["inverse", "solve"])
def test_c3_full_size_properties(mode):
    """BASELINE.json configs[2] at FULL size (N_train = 4096, D = 8, 1e6 candidates of the seed-1
    draw, AGP) in both variance forms, through properties that do not need the oracle on all 1e6 rows:
    (i) the fused arg-min equals the arg-min of the utilities the same launch returns for every candidate;
    (ii) the candidate matrix in reversed order gives the mirrored winner with the same bits;
    (iii) sigma^2 does not depend on y and mu is affine in y (two right-hand sides, same factor);
    (iv) a 4,096-row random subsample plus the winner's neighbourhood against the oracle."""
    import torch
    N, D, M = 4096, 8, 1_000_000
    X, y = _synthetic(N, D)
    gpo, gp = _pair(X, y, 8.0, D)
    gp.variance_mode = mode
    cands = np.random.RandomState(1).uniform(-5.0, 5.0, size=(M, D))
    bounds = [(-5, 5)] * D
    T = torch.from_numpy(cands).cuda()
    bi, bu, u, mu, var = gp.acquire(y, T, "agp", bounds=bounds, return_all=True)
    assert (bi, bu) == gp.acquire(y, T, "agp", bounds=bounds)          # (run to run)
    # (i)
    assert bi == int(np.nanargmin(u)) and bu == u[bi]
    assert np.isfinite(u).all() and (var > 0).all()
    # (ii)
    Tr = torch.from_numpy(np.ascontiguousarray(cands[::-1])).cuda()
    ri, ru = gp.acquire(y, Tr, "agp", bounds=bounds)
    assert (ri, ru) == (M - 1 - bi, bu)
    del Tr
    # (iii)
    y2 = 0.5 * y + np.sin(X[:, 0])
    mu2, var2 = gp.predict(y2, cands, return_var=True)
    assert np.array_equal(var2, var)
    rows = np.unique(np.concatenate([np.random.RandomState(7).randint(0, M, size=4096),
                                     np.arange(max(0, bi - 8), min(M, bi + 8))]))
    m_o, v_o = gpo.predict(y, cands[rows], return_var=True)
    m2_o, _ = gpo.predict(y2, cands[rows], return_var=True)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    tol = 1e-11                              # cond(K) ~ 1.5e3 at these hyper-parameters
    # (iv)
    assert np.abs(mu[rows] - m_o).max() <= tol * asum
    assert np.abs(mu2[rows] - m2_o).max() <= tol * max(asum, np.abs(gpo._compute_alpha(y2, False)).sum())
    assert np.abs(var[rows] - v_o).max() <= (tol if mode == "inverse" else 10 * tol)
    uo = _agp(m_o, v_o)
    assert np.abs(u[rows] - uo).max() <= 1e-9 * np.abs(uo).max()
    assert bu <= uo.min() + 1e-9 * abs(uo.min())
    del T


# --------------------------------------------------------------------------------------- C5
def _box(D):
    lo, hi = -5.0, 5.0
    def lnprior(t):
        t = np.asarray(t, dtype=float).ravel()
        return 0.0 if np.all((t >= lo) & (t <= hi)) else -np.inf
    def sample(n):
        return np.random.uniform(lo, hi, size=(int(n), D))
    return lnprior, sample, [(lo, hi)] * D


def _oracle_twin(gp, theta):
    """Oracle GP with the hyper-parameters of the HIP GP ``gp`` on training set ``theta``."""
    go, _ = _mods()
    p = gp.get_parameter_vector()
    D = theta.shape[1]
    o = go.GP(kernel=go.ExpSquaredKernel(np.exp(p[1:]), ndim=D), fit_mean=True, mean=float(p[0]),
              white_noise=float(gp.white_noise.value), fit_white_noise=False)
    o.compute(theta)
    return o


def test_c5_run_loop_d8_as_written(tmp_path, monkeypatch):
    """BASELINE.json configs[4] AS WRITTEN, with the reference's defaults (approx.py:229-235,397-424):
    ApproxPosterior.run at D = 8, m0 = 512, m = 64, **nmax = 10** (N grows 512 -> 1152),
    **optGPEveryN = 1** (an untruncated Powell re-optimisation of the hyper-parameters after every
    appended point: 640 of them), 64 walkers x 2e4 iterations on the on-device sampler,
    nCandidates = 1e6 (the fused sweep is the point search).  About 5 minutes on one MI355X
    (profiles/r03n_c5_as_written.txt: 283 s, 21-34 s of training and 1.1-1.5 s of MCMC per outer
    iteration).  Checked against the oracle:
      * appended design points (every 16th and the last three of the 640): (mu, sigma^2, u) of the
        sweep winner at the GP state that selected it (hyper-parameters + training set at that
        moment) -- approx.py:648-691;
      * the final factor == a full oracle refit (log-likelihood and predictions) -- approx.py:693-723;
      * the device sampler's log-probabilities == oracle GP mean at the sampled coordinates;
      * the host (batched) sampler on the same surrogate: same check on its chain."""
    monkeypatch.chdir(tmp_path)
    from scipy.optimize import rosen
    from approxposterior_amd import approx, gpUtils, utility as ut
    D, m0, m, nmax = 8, 512, 64, 10
    lnprior, sample, bounds = _box(D)
    lnlike = lambda t, *a, **k: -rosen(np.asarray(t).ravel()) / 100.0   # noqa: E731
    np.random.seed(11)
    theta = sample(m0)
    y = np.array([lnlike(t) + lnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lnprior, lnlike=lnlike,
                                priorSample=sample, bounds=bounds, algorithm="agp")
    picks = []
    real_sweep = ut.sweepObjective

    def spy(fn, yy, g, cands, **kw):
        best, u = real_sweep(fn, yy, g, cands, **kw)
        mu, var = g.predict(yy, best.reshape(1, -1), return_var=True)
        picks.append((best.copy(), float(u), float(mu[0]), float(var[0]), g.get_parameter_vector().copy(),
                      np.array(g._x, copy=True), np.array(yy, copy=True)))
        return best, u
    monkeypatch.setattr(ut, "sweepObjective", spy)
    with np.errstate(all="ignore"):
        ap.run(m=m, nmax=nmax, nCandidates=1_000_000, nGPRestarts=1, cache=False,
               verbose=False, onDevice=True, estBurnin=True, thinChains=True,
               mcmcKwargs={"iterations": 20000}, samplerKwargs={"nwalkers": 64})
    assert len(picks) == nmax * m and len(ap.y) == m0 + nmax * m == 1152 and ap.gp._x.shape == (1152, D)
    go, _ = _mods()
    for best, u, mu, var, p, Xs, ys in picks[::16] + picks[-3:]:
        o = go.GP(kernel=go.ExpSquaredKernel(np.exp(p[1:]), ndim=D), fit_mean=True, mean=float(p[0]),
                  white_noise=-12.0, fit_white_noise=False)
        o.compute(Xs)
        mo, vo = o.predict(ys, best.reshape(1, -1), return_var=True)
        asum = np.abs(o._compute_alpha(ys, False)).sum()
        assert abs(mu - mo[0]) <= 1e-9 * asum and abs(var - vo[0]) <= 1e-9
        assert abs(u - _agp(mo, vo)[0]) <= 1e-7 * max(1.0, abs(u))
        assert lnprior(best) == 0.0
    # factor after the incremental appends vs a full oracle refit
    twin = _oracle_twin(ap.gp, ap.theta)
    assert np.isclose(ap.gp.log_likelihood(ap.y), twin.log_likelihood(ap.y), rtol=1e-10)
    T = np.random.RandomState(3).uniform(-5, 5, size=(512, D))
    mo, vo = twin.predict(ap.y, T, return_var=True)
    mu, var = ap.gp.predict(ap.y, T, return_var=True)
    asum = np.abs(twin._compute_alpha(ap.y, False)).sum()
    assert np.abs(mu - mo).max() <= 1e-9 * asum and np.abs(var - vo).max() <= 1e-9
    # device sampler: log-probability == oracle mean at the sampled coordinates
    chain, logp = ap.sampler.get_chain(), ap.sampler.get_log_prob()
    assert chain.shape == (20000, 64, D) and len(ap.iburns) == nmax
    for it in (0, 9999, 19999):
        assert np.abs(logp[it] - twin.predict(ap.y, chain[it], return_cov=False)).max() <= 1e-9 * asum
        assert np.all(np.abs(chain[it]) <= 5.0)
    # host sampler (one batched mean launch per half-step) on the same surrogate
    with np.errstate(all="ignore"):
        sampler, _, _ = ap.runMCMC(samplerKwargs={"nwalkers": 64}, mcmcKwargs={"iterations": 300},
                                   cache=False, estBurnin=False, thinChains=False)
    ch, lp = sampler.get_chain(), sampler.get_log_prob()
    assert np.abs(lp[-1] - twin.predict(ap.y, ch[-1], return_cov=False)).max() <= 1e-9 * asum


def test_c5_device_sampler_final_size():
    """The persistent-kernel sampler at the FINAL C5 size: N = 1152 = 512 + 10 x 64, D = 8,
    64 walkers x 2e4 iterations; log-probabilities along the chain == oracle GP mean."""
    N, D = 1152, 8
    X, y = _synthetic(N, D, seed=5)
    gpo, gp = _pair(X, y, 8.0, D)
    p0 = np.random.RandomState(2).uniform(-5, 5, size=(64, D))
    res = gp.sample_ensemble(y, p0, 20000, [(-5, 5)] * D, seed=123)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    for it in (0, 5000, 19999):
        want = gpo.predict(y, res["chain"][it], return_cov=False)
        assert np.abs(res["log_prob"][it] - want).max() <= 1e-9 * asum
    assert np.abs(res["final_log_prob"] - gpo.predict(y, res["coords"], return_cov=False)).max() <= 1e-9 * asum
    acc = res["naccept"].sum() / (20000.0 * 64)
    assert 0.05 < acc < 0.9


# ------------------------------------------------------------------- a6 / a12 on the HIP path
def _rosen_set(m0, corners=False):
    from approxposterior_amd import likelihood as lh
    theta = np.array(lh.rosenbrockSample(m0))
    if corners:
        theta = np.array(list(theta) + [[-5, 5], [5, 5]])
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    return theta, y


def test_optimizegp_hip_reproduces_reference_constant(golden_dir):
    """test_OptimizeGP.py:70-92 on the HIP-backed GP: seed 57, Rosenbrock m0 = 50,
    defaultGP(fitAmp=False), optimizeGP(seed=57, nGPRestarts=5) -> hyper-parameters
    [-1.54256578, 3.24723589] (rtol 1e-2, the reference's own tolerance) and the harness
    replay of the oracle (pins.json optgp_noamp; Powell's path may differ in the last bits of
    the likelihood, so 1e-3 on the optimum, 1e-9 on the likelihood there).  Both the
    batched-restart and the sequential form."""
    from approxposterior_amd import gpUtils
    pins = json.load(open(os.path.join(golden_dir, "pins.json")))
    want = pins["reference_test_constants"]["test_OptimizeGP.py:91"]
    replay = pins["harness_replay"]["optgp_noamp"]
    for batch in ("always", True, False):      # (lock-step forced | by size: sequential at N = 50 | sequential)
        np.random.seed(57)
        theta, y = _rosen_set(50)
        gp = gpUtils.defaultGP(theta, y, fitAmp=False)
        with np.errstate(all="ignore"):
            gp = gpUtils.optimizeGP(gp, theta, y, seed=57, nGPRestarts=5, batchRestarts=batch)
        p = gp.get_parameter_vector()
        assert np.allclose(p[1:], want, rtol=1e-2), (batch, p)
        assert np.allclose(p, replay["p"], rtol=1e-3), (batch, p)
        assert np.isclose(gp.log_likelihood(y), replay["ll"], rtol=1e-6)


def test_findnextpoint_hip_reproduces_reference_constant(golden_dir):
    """test_findNewPoint.py:86-108 on the HIP-backed GP: seed 57, m0 = 50 + two corner points,
    defaultGP(fitAmp=False), BAPE, findNextPoint(computeLnLike=False, seed=57) ->
    [0.79813416, 0.85542199] (rtol 1e-3, the reference's tolerance)."""
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    pins = json.load(open(os.path.join(golden_dir, "pins.json")))
    np.random.seed(57)
    theta, y = _rosen_set(50, corners=True)
    gp = gpUtils.defaultGP(theta, y, fitAmp=False)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior,
                                lnlike=lh.rosenbrockLnlike, priorSample=lh.rosenbrockSample,
                                bounds=((-5, 5), (-5, 5)), algorithm="bape")
    with np.errstate(all="ignore"):
        thetaT = ap.findNextPoint(computeLnLike=False, bounds=((-5, 5), (-5, 5)), seed=57)
    assert np.allclose(thetaT, pins["reference_test_constants"]["test_findNewPoint.py:107"], rtol=1e-3)
    assert np.allclose(thetaT, pins["harness_replay"]["findnext_noamp"]["thetaT"], rtol=1e-4)


# ------------------------------------------------------------------------------ f2: 2-D BO
def test_bayesopt_2d(tmp_path, monkeypatch):
    """GPU counterpart of test_2DBayesOpt.py:15-73, same set-up call for call (seed 91, the
    direct Nelder-Mead solution drawn first, m0 = 10 sphere draws, fitAmp=True, Jones utility,
    bounds [-5,5]^2, nmax = 10): thetaBest / valBest and the MAP solution within atol 1e-2 of
    the directly minimised objective, as the reference asserts."""
    monkeypatch.chdir(tmp_path)
    from scipy.optimize import minimize
    from approxposterior_amd import approx, gpUtils, likelihood as lh
    np.random.seed(91)
    fn = lambda x: -(lh.sphereLnlike(x) + lh.sphereLnprior(x))   # noqa: E731
    true = minimize(fn, lh.sphereSample(1), method="nelder-mead")
    theta = lh.sphereSample(10)
    y = np.array([lh.sphereLnlike(t) + lh.sphereLnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y, fitAmp=True)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.sphereLnprior,
                                lnlike=lh.sphereLnlike, priorSample=lh.sphereSample,
                                bounds=[[-5, 5], [-5, 5]], algorithm="jones")
    with np.errstate(all="ignore"):
        soln = ap.bayesOpt(nmax=10, tol=1.0e-3, kmax=3, seed=91, cache=False, gpMethod="powell",
                           optGPEveryN=1, nGPRestarts=3, nMinObjRestarts=5, initGPOpt=True,
                           minObjMethod="nelder-mead", verbose=False, findMAP=True,
                           gpHyperPrior=gpUtils.defaultHyperPrior)
    assert np.allclose(soln["thetaBest"], true["x"], atol=1.0e-2)
    assert np.allclose(soln["valBest"], true["fun"], atol=1.0e-2)
    assert np.allclose(soln["thetaMAPBest"], true["x"], atol=1.0e-2)
    assert np.allclose(soln["valMAPBest"], true["fun"], atol=1.0e-2)
