# -*- coding: utf-8 -*-
"""
:py:mod:`mcmc.py` - affine-invariant ensemble sampler driving the batched GP
---------------------------------------------------------------------------

The reference samples its GP surrogate with ``emcee.EnsembleSampler``
(approx.py:839-846); emcee is a third-party dependency that is neither vendored
in the reference nor installed here, so this module restates the published
algorithm it relies on -- the Goodman & Weare (2010) stretch move as emcee >= 3
implements it (red/blue split, a = 2) -- behind the
subset of emcee's interface the reference touches:

    EnsembleSampler(nwalkers, ndim, log_prob_fn, args=, kwargs=, backend=, blobs_dtype=)
    sampler.sample(initial_state, iterations=...)   (generator, approx.py:846)
    sampler.run_mcmc(initial_state, nsteps)
    sampler.get_chain(discard=, thin=, flat=)       (approx.py:479)
    sampler.get_log_prob(...), sampler.get_blobs(...)
    sampler.get_autocorr_time(tol=0)                (mcmcUtils.py:198)
    sampler.acceptance_fraction, sampler.iteration

What is new relative to the reference's usage: with ``vectorize=True`` the
log-probability function receives the whole half-ensemble (S, D) in ONE call,
which ``ApproxPosterior._gpllBatch`` turns into one mean-only GP launch on the
MI355X instead of S scalar ``predict`` calls per half-step.  The chain is a
sequential process (each half-step depends on the previous one), so only that
per-half-step batch is parallel; independent ensembles shard across GPUs as
whole replicas.
"""

import numpy as np

__all__ = ["EnsembleSampler", "integrated_time", "AutocorrError"]


class AutocorrError(Exception):
    """Chain too short for a reliable autocorrelation time (emcee semantics)."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super(AutocorrError, self).__init__(*args, **kwargs)


def _next_pow_two(n):
    i = 1
    while i < n:
        i <<= 1
    return i


def _autocorr_1d(x):
    """Normalised autocorrelation function of a 1-D series via FFT."""
    x = np.atleast_1d(x)
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    acf /= acf[0]
    return acf


def _auto_window(taus, c):
    m = np.arange(len(taus)) < c * taus
    if np.any(m):
        return int(np.argmin(m))
    return len(taus) - 1


def integrated_time(x, c=5, tol=50, quiet=False):
    """Integrated autocorrelation time per dimension of a chain ``x`` of shape
    (nsteps, nwalkers, ndim), Sokal's automated windowing with window constant
    ``c`` (the estimator behind emcee's ``get_autocorr_time``; with ``tol=0`` it
    always returns an estimate, which is how mcmcUtils.py:198 calls it)."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, np.newaxis, np.newaxis]
    if x.ndim == 2:
        x = x[:, :, np.newaxis]
    if x.ndim != 3:
        raise ValueError("invalid dimensions")
    n_t, n_w, n_d = x.shape
    tau_est = np.empty(n_d)
    windows = np.empty(n_d, dtype=int)
    # real transforms (half the work of emcee's complex ones, same values to rounding), the series spread over a few host
    # threads (pocketfft releases the GIL): 0.6 s per call at 2e4 x 64 x 8 with the per-series complex FFTs -- 6 s of
    # BASELINE config 5's ten MCMC post-processing steps
    import os
    from concurrent.futures import ThreadPoolExecutor
    n2 = 2 * _next_pow_two(n_t)

    def acf_of(series):
        spec = np.fft.rfft(series - np.mean(series), n=n2)
        acf = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, n=n2)[:n_t]
        return acf / acf[0]
    cols = [np.ascontiguousarray(x[:, k, d]) for d in range(n_d) for k in range(n_w)]
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) // 2))) as pool:
        acfs = list(pool.map(acf_of, cols))
    for d in range(n_d):
        f = np.zeros(n_t)
        for k in range(n_w):
            f += acfs[d * n_w + k]
        f /= n_w
        taus = 2.0 * np.cumsum(f) - 1.0
        windows[d] = _auto_window(taus, c)
        tau_est[d] = taus[windows[d]]
    flag = tol * tau_est > n_t
    if np.any(flag) and tol > 0:
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} "
               "parameter(s). Use this estimate with caution and run a longer chain!\n"
               "N/{0} = {2:.0f};\ntau: {3}").format(tol, np.sum(flag), n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
    return tau_est


class EnsembleSampler(object):
    """Goodman-Weare stretch-move ensemble sampler (emcee-3 semantics)."""

    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, backend=None,
                 blobs_dtype=None, vectorize=False, a=2.0, seed=None, moves=None, pool=None):
        if nwalkers < 2 * ndim or nwalkers % 2 != 0:
            # emcee's own constraints for the red/blue move
            if nwalkers % 2 != 0:
                raise ValueError("The number of walkers must be even.")
            raise ValueError("The number of walkers needs to be at least twice the dimension.")
        if moves is not None or pool is not None:
            raise NotImplementedError("only the default stretch move, no pool")
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        self.log_prob_fn = log_prob_fn
        self.args = () if args is None else tuple(args)
        self.kwargs = {} if kwargs is None else dict(kwargs)
        self.vectorize = bool(vectorize)
        self.a = float(a)
        self.blobs_dtype = blobs_dtype
        self.backend = backend
        # own RandomState, like emcee: independent of numpy's global stream unless seeded
        self._random = np.random.RandomState(seed)
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self):
        self.iteration = 0
        self._chain = []
        self._log_prob = []
        self._blobs = []
        self._naccepted = np.zeros(self.nwalkers)
        self._coords = None
        self._lp = None
        self._bl = None

    @property
    def acceptance_fraction(self):
        return self._naccepted / max(self.iteration, 1)

    # -------------------------------------------------------------- log-prob
    def compute_log_prob(self, coords):
        """log-probability (and first blob, if any) for every row of ``coords``."""
        coords = np.atleast_2d(coords)
        if not np.isfinite(coords).all():                 # (one pass in the usual case; the two messages as before)
            if np.isinf(coords).any():
                raise ValueError("At least one parameter value was infinite")
            raise ValueError("At least one parameter value was NaN")
        if self.vectorize:
            res = self.log_prob_fn(coords, *self.args, **self.kwargs)
            if isinstance(res, tuple):
                lp = np.asarray(res[0], dtype=float).reshape(len(coords))
                blob = np.asarray(res[1], dtype=float).reshape(len(coords)) if len(res) > 1 else None
            else:
                lp = np.asarray(res, dtype=float).reshape(len(coords))
                blob = None
        else:
            lp = np.empty(len(coords))
            blob = None
            for i, c in enumerate(coords):
                r = self.log_prob_fn(c, *self.args, **self.kwargs)
                if isinstance(r, (tuple, list)):
                    lp[i] = float(np.ravel(r[0])[0])
                    if len(r) > 1:
                        if blob is None:
                            blob = np.full(len(coords), np.nan)
                        blob[i] = float(np.ravel(r[1])[0])
                else:
                    lp[i] = float(np.ravel(r)[0])
        if np.isnan(lp).any():
            raise ValueError("Probability function returned NaN")
        return lp, blob

    # ---------------------------------------------------------------- sample
    def sample(self, initial_state, iterations=1, store=True, progress=False, **unused):
        """Advance the ensemble ``iterations`` steps; yields (coords, log_prob,
        blobs) after each step (the reference just drains it, approx.py:846)."""
        p = np.array(initial_state, dtype=float, copy=True)
        if p.ndim == 1:
            p = p.reshape(self.nwalkers, self.ndim)
        if p.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions")
        if self._coords is None or initial_state is not None:
            self._coords = p
            self._lp, self._bl = self.compute_log_prob(p)
            if np.any(np.isneginf(self._lp)) and np.all(np.isneginf(self._lp)):
                raise ValueError("Initial state has a zero probability for every walker")
        nw, nd, a = self.nwalkers, self.ndim, self.a
        for _ in range(int(iterations)):
            inds = np.arange(nw) % 2
            self._random.shuffle(inds)
            halves = (np.flatnonzero(inds == 0), np.flatnonzero(inds == 1))
            for split in range(2):
                S, C = halves[split], halves[1 - split]
                s, c = self._coords[S], self._coords[C]
                zz = ((a - 1.0) * self._random.rand(len(S)) + 1.0) ** 2.0 / a
                factors = (nd - 1.0) * np.log(zz)
                rint = self._random.randint(len(C), size=(len(S),))
                q = c[rint] - (c[rint] - s) * zz[:, None]
                new_lp, new_bl = self.compute_log_prob(q)
                with np.errstate(invalid="ignore"):      # -inf - -inf for walkers outside the prior
                    lnpdiff = factors + new_lp - self._lp[S]
                accepted = np.log(self._random.rand(len(S))) < lnpdiff
                idx = S[accepted]
                self._coords[idx] = q[accepted]
                self._lp[idx] = new_lp[accepted]
                if new_bl is not None:
                    if self._bl is None:
                        self._bl = np.full(nw, np.nan)
                    self._bl[idx] = new_bl[accepted]
                self._naccepted[idx] += 1
            self.iteration += 1
            if store:
                self._chain.append(self._coords.copy())
                self._log_prob.append(self._lp.copy())
                if self._bl is not None:
                    self._blobs.append(self._bl.copy())
            yield self._coords, self._lp, self._bl

    def run_mcmc(self, initial_state, nsteps, **kwargs):
        out = None
        for out in self.sample(initial_state, iterations=nsteps, **kwargs):
            pass
        return out

    # ---------------------------------------------------------------- access
    def _get(self, store, discard, thin, flat):
        if len(store) == 0:
            raise AttributeError("you must run the sampler before accessing the results")
        v = np.asarray(store)[discard + thin - 1::thin]
        if flat:
            v = v.reshape((-1,) + v.shape[2:])
        return v

    def get_chain(self, discard=0, thin=1, flat=False):
        return self._get(self._chain, int(discard), int(thin), flat)

    def get_log_prob(self, discard=0, thin=1, flat=False):
        return self._get(self._log_prob, int(discard), int(thin), flat)

    def get_blobs(self, discard=0, thin=1, flat=False):
        if len(self._blobs) == 0:
            return None
        return self._get(self._blobs, int(discard), int(thin), flat)

    def get_autocorr_time(self, discard=0, thin=1, **kwargs):
        return thin * integrated_time(self.get_chain(discard=discard, thin=thin), **kwargs)


class DeviceChain(EnsembleSampler):
    """Result of an on-device run (``GP.sample_ensemble``) behind the same accessors
    as :class:`EnsembleSampler` (``get_chain``, ``get_log_prob``,
    ``get_autocorr_time``, ``acceptance_fraction``), so burn-in estimation and the
    reference-style post-processing work unchanged."""

    def __init__(self, result):
        chain = result["chain"]
        self.nwalkers = chain.shape[1]
        self.ndim = chain.shape[2]
        self.log_prob_fn = None
        self.args, self.kwargs = (), {}
        self.vectorize = True
        self.a = 2.0
        self.blobs_dtype = None
        self.backend = None
        self._random = None
        self.iteration = chain.shape[0]
        self._chain = chain
        self._log_prob = result["log_prob"]
        blobs = result.get("blobs")
        self._blobs = [] if blobs is None else blobs      # (iterations, walkers) lnprior values of a gathered host chain
        self._naccepted = np.asarray(result["naccept"], dtype=float)
        self._coords = result["coords"]
        self._lp = result["final_log_prob"]
        self._bl = None

    def sample(self, *args, **kwargs):
        raise NotImplementedError("a finished on-device chain cannot be advanced from the host")
