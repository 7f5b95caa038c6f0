# -*- coding: utf-8 -*-
"""
:py:mod:`mcmcUtils.py` - MCMC keyword hygiene and chain diagnostics
-------------------------------------------------------------------

Mirror of the reference's ``approxposterior/mcmcUtils.py``:
``validateMCMCKwargs`` (:15-100), ``batchMeansMCSE`` (:103-161) and
``estimateBurnin`` (:164-227), same names, arguments and return values.  These
sit on the caller side of the GP hot path ("next" row 2)
and are plain NumPy.  One deliberate difference: with ``samplerKwargs=None`` the
reference reads the non-existent key ``"dim"`` and raises ``KeyError``
(mcmcUtils.py:47, quirk Q3); here the documented default of 20 walkers per
dimension is applied.
"""

import numpy as np

__all__ = ["validateMCMCKwargs", "batchMeansMCSE", "estimateBurnin"]


def validateMCMCKwargs(ap, samplerKwargs, mcmcKwargs, verbose=False):
    """Sanitise the sampler / sampling keyword dictionaries for an
    :class:`ApproxPosterior` ``ap`` (mcmcUtils.py:15-100): force ``ndim`` and
    ``log_prob_fn`` (the GP surrogate), drop user backends, default to 20 walkers
    per dimension, 10,000 iterations and prior draws as the initial state."""
    if samplerKwargs is None:
        samplerKwargs = dict()
        samplerKwargs["ndim"] = ap.ndim
        samplerKwargs["nwalkers"] = 20 * samplerKwargs["ndim"]
        samplerKwargs["log_prob_fn"] = ap._gpll
    else:
        samplerKwargs.pop("ndim", None)
        samplerKwargs["ndim"] = ap.ndim
        if "nwalkers" not in samplerKwargs:
            print("WARNING: samplerKwargs provided but nwalkers not in samplerKwargs")
            print("Defaulting to nwalkers = 20 per dimension.")
            samplerKwargs["nwalkers"] = 20 * samplerKwargs["ndim"]
        if "backend" in samplerKwargs.keys():
            print("WARNING: backend in samplerKwargs. approxposterior creates its own!")
            print("with filename = apRun.h5. Disregarding user-supplied backend.")
        samplerKwargs.pop("log_prob_fn", None)
        samplerKwargs.pop("backend", None)
        samplerKwargs["log_prob_fn"] = ap._gpll

    if mcmcKwargs is None:
        mcmcKwargs = dict()
        mcmcKwargs["iterations"] = 10000
        mcmcKwargs["initial_state"] = ap.priorSample(samplerKwargs["nwalkers"])
    else:
        if "iterations" not in mcmcKwargs:
            mcmcKwargs["iterations"] = 10000
            if verbose:
                print("WARNING: mcmcKwargs provided, but iterations not in mcmcKwargs.")
                print("Defaulting to iterations = 10000.")
        if "initial_state" not in mcmcKwargs:
            mcmcKwargs["initial_state"] = ap.priorSample(samplerKwargs["nwalkers"])
            if verbose:
                print("WARNING: mcmcKwargs provided, but initial_state not in mcmcKwargs.")
                print("Defaulting to nwalkers samples from priorSample.")
    return samplerKwargs, mcmcKwargs


def batchMeansMCSE(samples, bins=None, fn=None):
    """Monte Carlo standard error by non-overlapping batch means (Flegal, Haran &
    Jones 2008), per dimension (mcmcUtils.py:103-161)."""
    if fn is None:
        fn = lambda x: x   # noqa: E731
    if bins is None:
        bins = max(int(np.sqrt(len(samples))), 2)
    assert isinstance(bins, int), "num must be an interger"
    samples = np.asarray(samples)
    b = int(len(samples) / bins)
    if samples.ndim > 1:
        y = np.zeros((bins, samples.shape[-1]))
    else:
        y = np.zeros(bins)
    mu = np.mean(fn(samples), axis=0)
    for ii in range(bins):
        y[ii] = np.sum(fn(samples[ii * b:(ii + 1) * b]), axis=0) / b
    mcse = b / (bins - 1) * np.sum((y - mu) ** 2, axis=0)
    return np.sqrt(mcse / len(samples))


def estimateBurnin(sampler, estBurnin=True, thinChains=True, verbose=False):
    """Burn-in (2 max tau) and thinning (max(tau_min / 2, 1)) estimates from the
    integrated autocorrelation time of a finished sampler (mcmcUtils.py:164-227)."""
    tau = sampler.get_autocorr_time(tol=0)
    if np.any(~np.isfinite(tau)):
        tau = tau[np.isfinite(np.array(tau))]
        if len(tau) < 1:
            if verbose:
                print("Failed to compute integrated autocorrelation length, tau.")
                print("Setting tau = 1")
            tau = 1
    iburn = int(2.0 * np.max(tau)) if estBurnin else 0
    ithin = np.max((int(0.5 * np.min(tau)), 1)) if thinChains else 1
    if verbose:
        print("burn-in estimate: %d" % iburn)
        print("thin estimate: %d" % ithin)
    return iburn, ithin
