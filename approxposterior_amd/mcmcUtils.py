# -*- coding: utf-8 -*-
"""
:py:mod:`mcmcUtils.py` - sampler keyword hygiene and chain diagnostics
----------------------------------------------------------------------

Same public names, arguments and return values as the reference module
(``validateMCMCKwargs`` mcmcUtils.py:15-100, ``batchMeansMCSE`` :103-161,
``estimateBurnin`` :164-227); independent implementation.  These functions sit
on the caller side of the GP hot path ("next" row 2) and
are plain NumPy.  One deliberate difference: when ``samplerKwargs`` is ``None``
the reference looks up a key that does not exist (``"dim"``, quirk Q3) and
raises ``KeyError``; here the documented default of 20 walkers per dimension is
what happens.
"""

import numpy as np

__all__ = ["validateMCMCKwargs", "batchMeansMCSE", "estimateBurnin"]

_DEFAULT_ITERATIONS = 10000
_WALKERS_PER_DIM = 20


def _say(verbose, *lines):
    if verbose:
        for line in lines:
            print(line)


def validateMCMCKwargs(ap, samplerKwargs, mcmcKwargs, verbose=False):
    """Return ``(samplerKwargs, mcmcKwargs)`` made safe for sampling the GP
    surrogate of the :class:`ApproxPosterior` ``ap``.

    Whatever the user passed, ``ndim`` is the object's dimensionality and
    ``log_prob_fn`` is ``ap._gpll``; a user ``backend`` is discarded (the driver
    owns the chain cache); missing ``nwalkers`` becomes 20 per dimension, missing
    ``iterations`` 10,000 and a missing ``initial_state`` is drawn from
    ``ap.priorSample``.
    """
    skw = {} if samplerKwargs is None else samplerKwargs
    if "backend" in skw:
        print("WARNING: a sampler backend was supplied; approxposterior manages its own "
              "chain cache, the supplied backend is ignored.")
    for key in ("backend", "log_prob_fn", "ndim"):
        skw.pop(key, None)
    if "nwalkers" not in skw:
        if samplerKwargs is not None:
            print("WARNING: samplerKwargs given without nwalkers; using %d walkers per "
                  "dimension." % _WALKERS_PER_DIM)
        skw["nwalkers"] = _WALKERS_PER_DIM * ap.ndim
    skw["ndim"] = ap.ndim
    skw["log_prob_fn"] = ap._gpll

    mkw = {} if mcmcKwargs is None else mcmcKwargs
    if "iterations" not in mkw:
        if mcmcKwargs is not None:
            _say(verbose, "WARNING: mcmcKwargs given without iterations; using %d." % _DEFAULT_ITERATIONS)
        mkw["iterations"] = _DEFAULT_ITERATIONS
    if "initial_state" not in mkw:
        if mcmcKwargs is not None:
            _say(verbose, "WARNING: mcmcKwargs given without initial_state; drawing nwalkers "
                          "points from priorSample.")
        mkw["initial_state"] = ap.priorSample(skw["nwalkers"])
    return skw, mkw


def batchMeansMCSE(samples, bins=None, fn=None):
    """Monte Carlo standard error of the mean of ``fn(samples)`` by the method of
    non-overlapping batch means (Flegal, Haran & Jones 2008).

    ``samples`` is (nsamples,) or (nsamples, ndim); ``bins`` defaults to
    ``max(int(sqrt(nsamples)), 2)``; ``fn`` defaults to the identity.  Returns one
    value per dimension.
    """
    values = np.asarray(samples) if fn is None else np.asarray(fn(np.asarray(samples)))
    total = len(values)
    if bins is None:
        bins = max(int(np.sqrt(total)), 2)
    assert isinstance(bins, int), "bins must be an integer"
    width = total // bins                          # samples per batch; the remainder is unused
    batches = values[: bins * width].reshape((bins, width) + values.shape[1:])
    batch_means = batches.mean(axis=1)
    grand_mean = values.mean(axis=0)
    spread = np.sum((batch_means - grand_mean) ** 2, axis=0)
    return np.sqrt(width / (bins - 1.0) * spread / total)


def estimateBurnin(sampler, estBurnin=True, thinChains=True, verbose=False):
    """Burn-in and thinning suggestions from the integrated autocorrelation time
    ``tau`` of a finished sampler: ``iburn = int(2 max(tau))`` and
    ``ithin = max(int(min(tau) / 2), 1)``; 0 and 1 when the respective estimate is
    switched off.  Non-finite components of ``tau`` are dropped; if none is left
    ``tau = 1`` is assumed."""
    tau = np.atleast_1d(np.asarray(sampler.get_autocorr_time(tol=0), dtype=float))
    tau = tau[np.isfinite(tau)]
    if tau.size == 0:
        _say(verbose, "Could not estimate the integrated autocorrelation time; assuming tau = 1.")
        tau = np.ones(1)
    iburn = int(2.0 * tau.max()) if estBurnin else 0
    ithin = max(int(0.5 * tau.min()), 1) if thinChains else 1
    _say(verbose, "burn-in estimate: %d" % iburn, "thin estimate: %d" % ithin)
    return iburn, ithin
