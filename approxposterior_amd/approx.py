# -*- coding: utf-8 -*-
"""
:py:mod:`approx.py` - ApproxPosterior: the callers of the GP-surrogate hot path
-------------------------------------------------------------------------------

Mirror of the hot-path callers of the reference's ``approxposterior/approx.py``:
``ApproxPosterior.__init__`` (:77-145), ``_gpll`` (:148-189), ``optGP``
(:192-226) and ``findNextPoint`` (:527-754), with the same signatures, defaults,
return shapes and error behaviour, running on the HIP-backed GP of
:py:mod:`approxposterior_amd.gp`.

Additions the reference does not have (all off by default so reference-style
scripts behave the same):

* ``findNextPoint(nCandidates=M)`` replaces the restarted Nelder-Mead search by
  the fused device sweep over ``M`` prior draws (optionally polished by one
  Nelder-Mead run from the winner);
* ``_gpllBatch`` evaluates the surrogate log-probability for a whole walker
  ensemble in one launch (what an ensemble sampler calls with vectorize=True).

The outer-loop glue -- ``run`` (:229-524), ``runMCMC`` (:757-859), ``findMAP``
(:862-926), ``bayesOpt`` (:929-1151) -- follows the reference's control flow on
top of the build-side ensemble sampler (:py:mod:`approxposterior_amd.mcmc`,
emcee is not installable here) and writes the same ``.npz`` caches; the emcee
HDF5 backend is replaced by a ``<runName>.npz`` chain dump (h5py is absent).
"""

import numpy as np
from scipy.optimize import minimize

import time

from . import gp as george
from . import gpUtils
from . import mcmc as emcee   # drop-in for the ``emcee`` names used below
from . import mcmcUtils
from . import utility as ut

__all__ = ["ApproxPosterior"]


class ApproxPosterior(object):
    """Approximate-posterior driver around a GP surrogate (approx.py:29-145).

    Parameters are those of the reference: the training set ``theta`` (N, D) /
    ``y`` (N,), the callables ``lnprior``, ``lnlike``, ``priorSample``, the box
    ``bounds`` (one (lo, hi) per dimension), an optional pre-built ``gp`` and
    the point-selection ``algorithm`` in {"bape", "agp", "alternate", "jones"}.
    """

    def __init__(self, theta, y, lnprior, lnlike, priorSample, bounds, gp=None,
                 algorithm="bape"):
        if theta is None or y is None:
            raise ValueError("Must supply both theta and y for initial GP training set.")
        self.theta = np.array(theta).squeeze()
        self.y = np.array(y).squeeze()
        self.ndim = 1 if self.theta.ndim <= 1 else theta.shape[-1]
        if np.any(~np.isfinite(self.theta)) or np.any(~np.isfinite(self.y)):
            print("theta, y:", theta, y)
            raise ValueError("All theta and y values must be finite!")
        if len(bounds) != self.ndim:
            raise ValueError("ERROR: bounds provided but len(bounds) != ndim.\n"
                             "ndim = %d, len(bounds) = %d" % (self.ndim, len(bounds)))
        self.bounds = bounds
        self._lnprior = lnprior
        self._lnlike = lnlike
        self.priorSample = priorSample
        self.algorithm = str(algorithm).lower()
        table = {"bape": ut.BAPEUtility, "agp": ut.AGPUtility,
                 "alternate": ut.AGPUtility, "jones": ut.JonesUtility}
        if self.algorithm not in table:
            raise ValueError("Unknown algorithm. Valid options: bape, agp, naive, or alternate.")
        self.utility = table[self.algorithm]
        self.iburns = list()
        self.ithins = list()
        self.backends = list()
        self.sampler = None
        self.gpPar = list()
        if gp is None:
            print("INFO: No GP specified. Initializing GP using ExpSquaredKernel.")
            self.gp = gpUtils.defaultGP(self.theta, self.y)
        else:
            self.gp = gp

    # ------------------------------------------------------------------ _gpll
    def _gpll(self, theta, *args, **kwargs):
        """Surrogate log-probability for one point: ``(mu(theta), lnprior)`` or
        ``(-inf, nan)`` under the reference's three guards (approx.py:148-189):
        all-non-finite theta, non-finite prior, non-finite / failed prediction."""
        if not np.any(np.isfinite(theta)):
            return -np.inf, np.nan
        lnprior = self._lnprior(theta)
        if not np.isfinite(lnprior):
            return -np.inf, np.nan
        try:
            mu = self.gp.predict(self.y, np.array(theta).reshape(1, -1),
                                 return_cov=False, return_var=False)
        except ValueError:
            return -np.inf, np.nan
        if not np.isfinite(mu):
            return -np.inf, np.nan
        return mu, lnprior

    def _gpllBatch(self, thetas):
        """Vectorised :meth:`_gpll` for a walker ensemble ``thetas`` (W, D): one
        mean-only device launch.  Returns ``(logp (W,), lnprior (W,))`` with the
        same guard semantics row by row."""
        thetas = np.asarray(thetas, dtype=float)
        if thetas.ndim == 1:
            thetas = thetas.reshape(-1, self.ndim)
        W = len(thetas)
        lp = np.full(W, -np.inf)
        blob = np.full(W, np.nan)
        pri = np.array([self._lnprior(t) if np.any(np.isfinite(t)) else -np.inf for t in thetas],
                       dtype=float)
        ok = np.isfinite(pri)
        if ok.any():
            safe = np.where(np.isfinite(thetas[ok]), thetas[ok], 0.0)
            mu = self.gp.predict(self.y, safe, return_cov=False, return_var=False)
            mu = np.where(np.all(np.isfinite(thetas[ok]), axis=1), mu, np.nan)
            good = np.isfinite(mu)
            idx = np.flatnonzero(ok)
            lp[idx[good]] = mu[good]
            blob[idx[good]] = pri[idx[good]]
        return lp, blob

    # ------------------------------------------------------------------ optGP
    def optGP(self, seed=None, method="powell", options=None, p0=None,
              nGPRestarts=1, gpHyperPrior=gpUtils.defaultHyperPrior):
        """Re-fit the GP hyper-parameters in place (approx.py:192-226)."""
        self.gp = gpUtils.optimizeGP(self.gp, self.theta, self.y, seed=seed,
                                     method=method, options=options, p0=p0,
                                     nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)

    # ---------------------------------------------------------- findNextPoint
    def findNextPoint(self, theta0=None, computeLnLike=True, seed=None,
                      cache=True, gpOptions=None, gpP0=None, verbose=True,
                      nGPRestarts=1, nMinObjRestarts=5, gpMethod="powell",
                      minObjMethod="nelder-mead", minObjOptions=None,
                      runName="apRun", numNewPoints=1, optGPEveryN=1,
                      gpHyperPrior=gpUtils.defaultHyperPrior, args=None,
                      nCandidates=None, polish=False, **kwargs):
        """Select ``numNewPoints`` design points by minimising the (negative)
        utility, optionally evaluate the forward model there, append to the
        training set, re-factorise and periodically re-fit the GP
        (approx.py:527-754; same return shapes :745-753).

        With ``nCandidates`` the point search is the fused device sweep over
        that many ``priorSample`` draws (box ``bounds`` fused as the prior)
        instead of ``nMinObjRestarts`` Nelder-Mead runs; ``polish`` then refines
        the winner with one Nelder-Mead run of the scalar utility.
        """
        assert isinstance(numNewPoints, int) and numNewPoints >= 1
        assert isinstance(optGPEveryN, int) and optGPEveryN >= 1
        if verbose and numNewPoints < optGPEveryN:
            print("WARNING: numNewPoints < optGPEveryN."
                  "GP hyperparameters will not be re-optimized. Set "
                  "numNewPoints < optGPEveryN to fix this, if important (it probably is).")
        if args is None:
            args = ()
        newTheta = list()
        newY = list()
        for ii in range(numNewPoints):
            if self.algorithm == "alternate":      # AGP on even, BAPE on odd (approx.py:656-661)
                self.utility = ut.AGPUtility if ii % 2 == 0 else ut.BAPEUtility
            if nCandidates is None:
                thetaT, uT = ut.minimizeObjective(self.utility, self.y, self.gp,
                                                  sampleFn=self.priorSample,
                                                  priorFn=self._lnprior,
                                                  nRestarts=nMinObjRestarts,
                                                  method=minObjMethod,
                                                  options=minObjOptions,
                                                  bounds=self.bounds, theta0=theta0,
                                                  args=(self.y, self.gp, self._lnprior))
            else:
                cands = np.asarray(self.priorSample(int(nCandidates)), dtype=float)
                cands = cands.reshape(int(nCandidates), -1)
                thetaT, uT = ut.sweepObjective(self.utility, self.y, self.gp, cands,
                                               bounds=self.bounds)
                if polish:
                    thetaT, uT = ut.minimizeObjective(self.utility, self.y, self.gp,
                                                      sampleFn=self.priorSample,
                                                      priorFn=self._lnprior, nRestarts=1,
                                                      method=minObjMethod, options=minObjOptions,
                                                      bounds=self.bounds, theta0=thetaT,
                                                      args=(self.y, self.gp, self._lnprior))
            newTheta.append(thetaT)
            if computeLnLike:
                loglikeT = self._lnlike(thetaT, *args, **kwargs)
                if hasattr(loglikeT, "__iter__"):
                    yT = np.array([loglikeT[0] + self._lnprior(thetaT)])
                else:
                    yT = np.array([loglikeT + self._lnprior(thetaT)])
                newY.append(yT)
                if self.theta.ndim > 1:
                    self.theta = np.vstack([self.theta, np.array(thetaT)])
                else:
                    self.theta = np.hstack([self.theta, thetaT])
                self.y = np.hstack([self.y, yT])
                try:
                    currentHype = self.gp.get_parameter_vector()
                    if verbose:
                        print('hyperparameters', currentHype)
                    if cache:
                        self.gpPar.append(currentHype)
                    # same kernel / mean / white-noise objects, new training set
                    oldGP = self.gp
                    self.gp = george.GP(kernel=self.gp.kernel, fit_mean=True,
                                        mean=self.gp.mean,
                                        white_noise=self.gp.white_noise,
                                        fit_white_noise=False)
                    self.gp.set_parameter_vector(currentHype)
                    if hasattr(self.gp, "_try_extend"):
                        # same hyper-parameters, one more row: O(N^2) factor extension
                        self.gp.compute(self.theta, previous=oldGP)
                    else:
                        self.gp.compute(self.theta)
                    if ii % optGPEveryN == 0:
                        self.optGP(seed=seed, method=gpMethod, options=gpOptions,
                                   p0=gpP0, nGPRestarts=nGPRestarts,
                                   gpHyperPrior=gpHyperPrior)
                except ValueError:
                    print("theta:", self.theta)
                    print("y:", self.y)
                    print("gp parameters names:", self.gp.get_parameter_names())
                    print("gp parameters:", self.gp.get_parameter_vector())
                    raise ValueError("GP couldn't optimize!")
                if cache:
                    np.savez(str(runName) + "APFModelCache.npz", theta=self.theta, y=self.y)
        if numNewPoints == 1:
            newTheta = newTheta[0]
            if computeLnLike:
                newY = newY[0]
        if computeLnLike:
            return np.asarray(newTheta), np.asarray(newY)
        return np.asarray(newTheta)

    # ---------------------------------------------------------------- runMCMC
    def runMCMC(self, samplerKwargs=None, mcmcKwargs=None, runName="apRun",
                cache=True, estBurnin=True, thinChains=True, verbose=False,
                args=None, batched=True, onDevice=False, **kwargs):
        """Sample the GP surrogate posterior with the stretch-move ensemble
        sampler and estimate burn-in / thinning (approx.py:757-859).  Returns
        ``(sampler, iburn, ithin)``.

        ``batched=True`` (default) evaluates each half-ensemble with ONE
        mean-only GP launch (``_gpllBatch``); ``batched=False`` calls the scalar
        ``_gpll`` once per walker exactly as emcee does for the reference;
        ``onDevice=True`` runs the entire chain as one persistent kernel
        (``GP.sample_ensemble``) -- only valid when ``lnprior`` is the box prior
        ``self.bounds`` (constant inside, -inf outside).
        With ``cache=True`` the finished chain is written to ``<runName>.npz``
        (keys chain, log_prob, blobs) where the reference writes ``<runName>.h5``.
        """
        samplerKwargs, mcmcKwargs = mcmcUtils.validateMCMCKwargs(self, samplerKwargs,
                                                                 mcmcKwargs, verbose)
        if onDevice:
            # the whole loop as one persistent kernel; valid when lnprior is the box
            # prior self.bounds (constant inside, -inf outside), which the caller asserts
            res = self.gp.sample_ensemble(self.y, mcmcKwargs["initial_state"],
                                          mcmcKwargs["iterations"], self.bounds,
                                          seed=np.random.randint(0, 2 ** 31 - 1))
            self.sampler = emcee.DeviceChain(res)
            if cache:
                bname = str(runName) + ".npz"
                self.backends.append(bname)
                np.savez(bname, chain=self.sampler.get_chain(), log_prob=self.sampler.get_log_prob(),
                         blobs=np.array([]))
            iburn, ithin = mcmcUtils.estimateBurnin(self.sampler, estBurnin=estBurnin,
                                                    thinChains=thinChains, verbose=verbose)
            return self.sampler, iburn, ithin
        skw = dict(samplerKwargs)
        if batched:
            skw["log_prob_fn"] = lambda thetas, *a, **k: self._gpllBatch(thetas)
            skw["vectorize"] = True
        self.sampler = emcee.EnsembleSampler(**skw, backend=None, args=args, kwargs=kwargs,
                                             blobs_dtype=[("lnprior", float)])
        for _ in self.sampler.sample(**mcmcKwargs):
            pass
        if verbose:
            print("mcmc finished")
        if cache:
            bname = str(runName) + ".npz"
            self.backends.append(bname)
            blobs = self.sampler.get_blobs()
            np.savez(bname, chain=self.sampler.get_chain(), log_prob=self.sampler.get_log_prob(),
                     blobs=np.array([]) if blobs is None else blobs)
        iburn, ithin = mcmcUtils.estimateBurnin(self.sampler, estBurnin=estBurnin,
                                                thinChains=thinChains, verbose=verbose)
        return self.sampler, iburn, ithin

    # -------------------------------------------------------------------- run
    def run(self, m=10, nmax=2, seed=None, timing=False, verbose=True,
            mcmcKwargs=None, samplerKwargs=None, estBurnin=False,
            thinChains=False, runName="apRun", cache=True, gpMethod="powell",
            gpOptions=None, gpP0=None, optGPEveryN=1, nGPRestarts=1,
            nMinObjRestarts=5, onlyLastMCMC=False, initGPOpt=True, kmax=3,
            gpHyperPrior=gpUtils.defaultHyperPrior, eps=1.0, convergenceCheck=False,
            minObjMethod="nelder-mead", minObjOptions=None, args=None,
            nCandidates=None, **kwargs):
        """BAPE / AGP outer loop (approx.py:229-524): for ``nmax`` iterations find
        ``m`` new design points (re-fitting the GP every ``optGPEveryN``), run the
        surrogate MCMC, record burn-in / thinning and optionally stop when the
        marginal posterior means move by less than ``eps`` previous standard
        deviations for ``kmax`` consecutive iterations.  (The reference only
        honours that stop rule when ``verbose`` is set -- quirk Q4; here it does
        not depend on verbosity.)  ``nCandidates`` switches the point search to
        the fused device sweep."""
        if cache:
            np.savez(str(runName) + "APFModelCache.npz", theta=self.theta, y=self.y)
            self.gpPar = list()
        if seed is not None:
            np.random.seed(seed)
        if timing:
            self.trainingTime = list()
            self.mcmcTime = list()
        if convergenceCheck:
            self.marginalMeans = list()
            self.marginalStds = list()
            self.marginalZScores = list()
        if initGPOpt:
            self.optGP(seed=seed, method=gpMethod, options=gpOptions, p0=gpP0,
                       nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)
        kk = 0
        if convergenceCheck and onlyLastMCMC:
            raise RuntimeError("If convergenceCheck is True, must run an MCMC each iteration.\n"
                               "convergenceCheck = %d onlyLastMCMC = %d" % (convergenceCheck, onlyLastMCMC))
        for nn in range(nmax):
            if verbose:
                print("Iteration: %d" % nn)
            start = time.time()
            _, _ = self.findNextPoint(computeLnLike=True, seed=seed, cache=cache,
                                      gpMethod=gpMethod, gpOptions=gpOptions,
                                      nGPRestarts=nGPRestarts, nMinObjRestarts=nMinObjRestarts,
                                      optGPEveryN=optGPEveryN, numNewPoints=m,
                                      gpHyperPrior=gpHyperPrior, minObjMethod=minObjMethod,
                                      minObjOptions=minObjOptions, runName=runName,
                                      theta0=None, args=args, verbose=verbose,
                                      nCandidates=nCandidates, **kwargs)
            if timing:
                self.trainingTime.append(time.time() - start)
            if cache:
                np.savez(str(runName) + "APGP.npz",
                         gpParamNames=self.gp.get_parameter_names(),
                         gpParamValues=self.gpPar)
            if onlyLastMCMC and nn != (nmax - 1):
                self.sampler = None
                continue
            start = time.time()
            self.sampler, iburn, ithin = self.runMCMC(samplerKwargs=samplerKwargs,
                                                      mcmcKwargs=mcmcKwargs,
                                                      runName=str(runName) + str(nn),
                                                      cache=cache, estBurnin=estBurnin,
                                                      thinChains=thinChains, verbose=verbose,
                                                      args=args, **kwargs)
            self.iburns.append(iburn)
            self.ithins.append(ithin)
            if timing:
                self.mcmcTime.append(time.time() - start)
                if cache:
                    np.savez(str(runName) + "APTiming.npz", trainingTime=self.trainingTime,
                             mcmcTime=self.mcmcTime)
            if convergenceCheck:
                samples = self.sampler.get_chain(discard=self.iburns[-1], flat=True,
                                                 thin=self.ithins[-1])
                meanNN = np.mean(samples, axis=0)
                stdNN = np.std(samples, axis=0)
                self.marginalMeans.append(meanNN)
                self.marginalStds.append(stdNN)
                if nn > 0:
                    zScore = np.fabs((meanNN - meanPrev) / stdPrev)
                    self.marginalZScores.append(zScore)
                    kk = kk + 1 if np.all(zScore < eps) else 0
                meanPrev, stdPrev = meanNN, stdNN
                if cache:
                    np.savez(str(runName) + "ConvergenceCache.npz", means=self.marginalMeans,
                             stds=self.marginalStds, zscores=self.marginalZScores,
                             eps=eps, kmax=kmax, finalIteration=kk)
                if kk >= kmax:
                    if verbose:
                        print("Approximate marginal posterior distributions converged.")
                        print("Delta zScore threshold, eps: %e" % eps)
                        print("kk, kmax: %d, %d" % (kk, kmax))
                        print("Final abs(zScore):", zScore)
                    break

    # ---------------------------------------------------------------- findMAP
    def findMAP(self, theta0=None, method="nelder-mead", options=None, nRestarts=15):
        """Maximum a posteriori estimate of the function the GP has learned:
        minimise minus the GP mean from ``nRestarts`` starts around the best
        training point (approx.py:862-926).  Returns ``(MAP, MAPVal)``."""
        if theta0 is not None:
            theta0 = np.array(theta0).reshape(1, self.theta.shape[-1])
        else:
            theta0 = self.theta[np.argmax(self.y)]
        if str(method).lower() == "nelder-mead" and options is None:
            options = {"adaptive": True}

        def fn(x):
            if not np.isfinite(self._lnprior(x)):
                return np.inf
            return -(self._gpll(x)[0])

        MAP, MAPVal = ut.minimizeObjective(fn, self.y, self.gp, self.priorSample,
                                           self._lnprior, nRestarts=nRestarts, args=None,
                                           method=method, options=options,
                                           bounds=self.bounds, theta0=theta0)
        return MAP, -MAPVal

    # --------------------------------------------------------------- bayesOpt
    def bayesOpt(self, nmax, theta0=None, tol=1.0e-3, kmax=3, seed=None,
                 verbose=True, runName="apRun", cache=True, gpMethod="powell",
                 gpOptions=None, gpP0=None, optGPEveryN=1, nGPRestarts=1,
                 nMinObjRestarts=5, initGPOpt=True, minObjMethod="nelder-mead",
                 gpHyperPrior=gpUtils.defaultHyperPrior, minObjOptions=None,
                 findMAP=True, args=None, nCandidates=None, **kwargs):
        """Bayesian optimisation loop (approx.py:929-1151): one new design point
        per iteration by the object's utility (use algorithm="jones"), optional
        MAP of the GP mean each iteration, stop after ``kmax`` consecutive
        iterations whose best value changes by less than ``tol``.  Returns the
        reference's solution dictionary."""
        thetas, vals = list(), list()
        thetasMAP, valsMAP = list(), list()
        if cache:
            np.savez(str(runName) + "APFModelCache.npz", theta=self.theta, y=self.y)
        if seed is not None:
            np.random.seed(seed)
        if initGPOpt:
            self.optGP(seed=seed, method=gpMethod, options=gpOptions, p0=gpP0,
                       nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)
        kk = 0
        nn = -1
        for nn in range(nmax):
            if verbose:
                print("Iteration: %d" % nn)
            optN = 1 if nn % optGPEveryN == 0 else 99999999
            thetaT, yT = self.findNextPoint(computeLnLike=True, seed=seed, cache=cache,
                                            gpMethod=gpMethod, gpOptions=gpOptions,
                                            nGPRestarts=nGPRestarts,
                                            nMinObjRestarts=nMinObjRestarts,
                                            optGPEveryN=optN, numNewPoints=1,
                                            gpHyperPrior=gpHyperPrior,
                                            minObjMethod=minObjMethod,
                                            minObjOptions=minObjOptions, runName=runName,
                                            args=args, verbose=verbose,
                                            nCandidates=nCandidates, **kwargs)
            if verbose:
                print("Forward model evaluation at: ", thetaT, ", function value: ", yT)
            if cache:
                np.savez(str(runName) + "APGP.npz",
                         gpParamNames=self.gp.get_parameter_names(),
                         gpParamValues=self.gp.get_parameter_vector())
            thetas.append(self.theta[np.argmax(self.y)])
            vals.append(self.y[np.argmax(self.y)])
            if findMAP:
                thetaN, valN = self.findMAP(theta0=theta0, method=minObjMethod,
                                            options=minObjOptions, nRestarts=nMinObjRestarts)
                if verbose:
                    print("Current MAP solution: ", thetaN, valN)
                thetasMAP.append(thetaN)
                valsMAP.append(valN)
            if nn > 0:
                kk = kk + 1 if np.fabs(vals[-1] - vals[-2]) < tol else 0
                if kk >= kmax:
                    break
        soln = {"thetaBest": thetas[-1], "valBest": vals[-1],
                "thetas": np.asarray(thetas).squeeze(),
                "vals": np.asarray(vals).squeeze(), "nev": nn + 1}
        if findMAP:
            soln["thetasMAP"] = np.asarray(thetasMAP).squeeze()
            soln["valsMAP"] = np.asarray(valsMAP).squeeze()
            soln["thetaMAPBest"] = soln["thetasMAP"][np.argmax(soln["valsMAP"])]
            soln["valMAPBest"] = soln["valsMAP"][np.argmax(soln["valsMAP"])]
        return soln
