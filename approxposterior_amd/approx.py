# -*- coding: utf-8 -*-
"""
:py:mod:`approx.py` - ApproxPosterior: the callers of the GP-surrogate hot path
-------------------------------------------------------------------------------

The class the reference's users hold (``approxposterior/approx.py``): same public
method names, argument names, defaults, return shapes, random-number draw order
and cache files --

    ApproxPosterior(theta, y, lnprior, lnlike, priorSample, bounds, gp, algorithm)  :77-145
    _gpll(theta)                          surrogate log-probability        :148-189
    optGP(...)                            hyper-parameter re-fit           :192-226
    run(m, nmax, ...)                     BAPE / AGP outer loop            :229-524
    findNextPoint(...)                    design-point selection           :527-754
    runMCMC(...)                          surrogate MCMC                   :757-859
    findMAP(...)                          MAP of the GP mean               :862-926
    bayesOpt(nmax, ...)                   Bayesian optimisation            :929-1151

-- organised around the device path instead of the reference's inline loops: a
design point is found by :meth:`_selectPoint` (restarted Nelder-Mead as in the
reference, or the fused HIP sweep over ``nCandidates`` prior draws), absorbed by
:meth:`_absorbPoint` (training set + O(N^2) extension of the device-resident
Cholesky factor) and the stopping rules live in two small monitor classes.  The
supported way to run the reference's *own* loop on the MI355X is INTEGRATION.md
level 1 (hand the reference's ``ApproxPosterior`` a ``gp=`` from this package);
this class is for scripts that import ``approxposterior_amd`` directly.

Additions the reference does not have (all off by default):

* ``nCandidates=M`` (``findNextPoint`` / ``run`` / ``bayesOpt``): point search by the
  fused sweep over ``M`` ``priorSample`` draws, optionally ``polish``-ed by one
  Nelder-Mead run from the winner;
* ``_gpllBatch``: the surrogate log-probability of a whole walker ensemble in one
  launch (what the ensemble sampler calls with ``vectorize=True``);
* ``runMCMC(onDevice=True)`` / ``run(onDevice=True)``: the whole chain as one
  persistent kernel (box prior only);
* several GPUs: launched as ``python -m torch.distributed.run --nproc-per-node N script.py`` with a
  process group initialised before the object is used (``tools/run_c5_dist.py``), every rank runs the
  same outer loop on an identical training set -- the ``nCandidates`` sweep is sharded by rank with one
  16-byte-per-rank all-gather (:func:`dist.sharded_acquire`), each rank samples its own replica
  ensemble and the chains are gathered once (:func:`dist.replicated_ensembles`), optimiser restarts are
  spread over the ranks (:func:`dist.spread_restarts`), the forward model runs on rank 0 only and its
  value is broadcast, and only rank 0 prints and writes the caches (BASELINE config 5 "on 8 GPUs").

The emcee HDF5 backend is replaced by a ``<runName>.npz`` chain dump (h5py and emcee
are not installable here; :py:mod:`approxposterior_amd.mcmc` restates the sampler).
"""

import time

import numpy as np

from . import dist as apdist
from . import gp as george
from . import gpUtils
from . import mcmc as emcee   # drop-in for the ``emcee`` names used below
from . import mcmcUtils
from . import utility as ut

__all__ = ["ApproxPosterior"]

_UTILITIES = {"bape": ut.BAPEUtility, "agp": ut.AGPUtility,
              "alternate": ut.AGPUtility, "jones": ut.JonesUtility}


def _cacheName(runName, suffix):
    return "%s%s.npz" % (runName, suffix)


class _MarginalMonitor(object):
    """Stop rule of ``run`` (approx.py:476-523): the marginal posterior means of
    successive iterations, in units of the previous iteration's marginal standard
    deviations, must stay below ``eps`` for ``kmax`` consecutive iterations."""

    def __init__(self, eps, kmax):
        self.eps, self.kmax = eps, kmax
        self.means, self.stds, self.zscores = [], [], []
        self.streak = 0

    def update(self, samples):
        mean, std = np.mean(samples, axis=0), np.std(samples, axis=0)
        if self.means:
            z = np.fabs((mean - self.means[-1]) / self.stds[-1])
            self.zscores.append(z)
            self.streak = self.streak + 1 if np.all(z < self.eps) else 0
        self.means.append(mean)
        self.stds.append(std)
        return self.streak >= self.kmax

    def save(self, path):
        np.savez(path, means=self.means, stds=self.stds, zscores=self.zscores,
                 eps=self.eps, kmax=self.kmax, finalIteration=self.streak)


class _PlateauMonitor(object):
    """Stop rule of ``bayesOpt`` (approx.py:1118-1128): ``kmax`` consecutive
    iterations whose best value moved by less than ``tol``."""

    def __init__(self, tol, kmax):
        self.tol, self.kmax = tol, kmax
        self.last = None
        self.streak = 0

    def update(self, value):
        if self.last is not None:
            self.streak = self.streak + 1 if np.fabs(value - self.last) < self.tol else 0
        self.last = value
        return self.streak >= self.kmax


class ApproxPosterior(object):
    """Approximate-posterior driver around a GP surrogate (approx.py:29-145).

    ``theta`` (N, D) / ``y`` (N,) is the initial training set, ``lnprior``,
    ``lnlike`` and ``priorSample`` the user's callables, ``bounds`` one (lo, hi)
    pair per dimension, ``gp`` an optional pre-built GP and ``algorithm`` one of
    "bape", "agp", "alternate", "jones".  ``distributed`` (None: whenever a
    ``torch.distributed`` process group is initialised; False: never) and ``group``
    select the multi-GPU paths; the training set handed to every rank must be the same.
    """

    def __init__(self, theta, y, lnprior, lnlike, priorSample, bounds, gp=None,
                 algorithm="bape", distributed=None, group=None):
        self.distributed, self.group = distributed, group
        self.deviceCandidates = False      # nCandidates drawn on the device, uniform in ``bounds`` (see findNextPoint)
        if theta is None or y is None:
            raise ValueError("Must supply both theta and y for initial GP training set.")
        self.theta = np.array(theta).squeeze()
        self.y = np.array(y).squeeze()
        self.ndim = theta.shape[-1] if self.theta.ndim > 1 else 1
        if not (np.all(np.isfinite(self.theta)) and np.all(np.isfinite(self.y))):
            raise ValueError("All theta and y values must be finite!")
        if len(bounds) != self.ndim:
            raise ValueError("ERROR: bounds provided but len(bounds) != ndim.\n"
                             "ndim = %d, len(bounds) = %d" % (self.ndim, len(bounds)))
        self.bounds = bounds
        self._lnprior, self._lnlike, self.priorSample = lnprior, lnlike, priorSample
        self.algorithm = str(algorithm).lower()
        if self.algorithm not in _UTILITIES:
            raise ValueError("Unknown algorithm. Valid options: bape, agp, naive, or alternate.")
        self.utility = _UTILITIES[self.algorithm]
        self.iburns, self.ithins, self.backends, self.gpPar = [], [], [], []
        self.sampler = None
        if gp is None:
            print("INFO: No GP specified. Initializing GP using ExpSquaredKernel.")
            gp = gpUtils.defaultGP(self.theta, self.y)
        self.gp = gp

    # ---------------------------------------------------------------- several GPUs
    def _ranks(self):
        """``(rank, world)`` under an initialised process group (unless ``distributed=False``), else None."""
        return apdist.context(self.group, self.distributed)

    def _chief(self):
        """True on the rank that prints, writes caches and calls the forward model."""
        ranks = self._ranks()
        return ranks is None or ranks[0] == 0

    def _agree(self, *values):
        """Rank 0's float64 arrays on every rank (a no-op without a process group)."""
        if self._ranks() is None:
            return values if len(values) > 1 else values[0]
        out = tuple(apdist.broadcast_bytes(np.asarray(v, dtype=np.float64), 0, self.group, enabled=self.distributed)
                    for v in values)
        return out if len(out) > 1 else out[0]

    # ------------------------------------------------------- surrogate log-probability
    def _gpll(self, theta, *args, **kwargs):
        """``(mu(theta), lnprior(theta))``, or ``(-inf, nan)`` when theta has no finite
        coordinate, the prior excludes it, or the prediction fails / is not finite
        (the three guards of approx.py:148-189)."""
        rejected = (-np.inf, np.nan)
        if not np.any(np.isfinite(theta)):
            return rejected
        prior = self._lnprior(theta)
        if not np.isfinite(prior):
            return rejected
        try:
            mean = self.gp.predict(self.y, np.array(theta).reshape(1, -1),
                                   return_cov=False, return_var=False)
        except ValueError:
            return rejected
        return (mean, prior) if np.isfinite(mean) else rejected

    def _gpllBatch(self, thetas):
        """:meth:`_gpll` for a whole walker ensemble ``thetas`` (W, D) with one
        mean-only device launch: ``(logp (W,), lnprior (W,))``, guarded row by row."""
        pts = np.asarray(thetas, dtype=float)
        if pts.ndim == 1:
            pts = pts.reshape(-1, self.ndim)
        fin = np.isfinite(pts)
        batch = getattr(self._lnprior, "batch", None)
        if batch is not None and fin.all():
            # the usual half-step of the sampler, without the row-by-row bookkeeping below: every coordinate finite, the
            # prior's vectorised twin (likelihood.py) in one call, one mean-only launch for the walkers inside the prior
            prior = np.asarray(batch(pts), dtype=float).reshape(len(pts))
            inside = np.isfinite(prior)
            if inside.all():
                mean = self.gp.predict(self.y, pts, return_cov=False, return_var=False)
                ok = np.isfinite(mean)
                if ok.all():
                    return mean, prior
                return np.where(ok, mean, -np.inf), np.where(ok, prior, np.nan)
            logp = np.full(len(pts), -np.inf)
            blob = np.full(len(pts), np.nan)
            rows = np.flatnonzero(inside)
            if rows.size:
                mean = self.gp.predict(self.y, pts[rows], return_cov=False, return_var=False)
                ok = np.isfinite(mean)
                keep = rows[ok]
                logp[keep] = mean[ok]
                blob[keep] = prior[keep]
            return logp, blob
        logp = np.full(len(pts), -np.inf)
        blob = np.full(len(pts), np.nan)
        some = fin.any(axis=1)                      # (a walker without a finite coordinate is rejected before the prior)
        if batch is not None and some.all():
            # the prior's vectorised twin (likelihood.py): one call per ensemble instead of one per walker
            prior = np.asarray(batch(pts), dtype=float).reshape(len(pts))
        else:
            prior = np.array([self._lnprior(p) if ok else -np.inf for p, ok in zip(pts, some)], dtype=float)
        rows = np.flatnonzero(np.isfinite(prior))
        if rows.size:
            if rows.size == len(pts) and fin.all():
                finite, sel = np.ones(len(pts), dtype=bool), pts                      # (the usual half-step: nothing to mask)
            else:
                finite = np.all(fin[rows], axis=1)
                sel = np.where(fin[rows], pts[rows], 0.0)
            mean = self.gp.predict(self.y, sel, return_cov=False, return_var=False)
            ok = finite & np.isfinite(mean)
            keep = rows[ok]
            logp[keep] = mean[ok]
            blob[keep] = prior[keep]
        return logp, blob

    # ------------------------------------------------------------------ GP re-fit
    def optGP(self, seed=None, method="powell", options=None, p0=None,
              nGPRestarts=1, gpHyperPrior=gpUtils.defaultHyperPrior):
        """Re-fit the GP hyper-parameters in place (approx.py:192-226).  Under a process group the
        restarts are spread over the ranks and every rank ends with the same optimum."""
        if self._ranks() is not None:
            apdist.sync_random_state(0, self.group, enabled=self.distributed)      # the restart start points are one global draw
        self.gp = gpUtils.optimizeGP(self.gp, self.theta, self.y, seed=seed,
                                     method=method, options=options, p0=p0,
                                     nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior,
                                     distributed=self.distributed, group=self.group)

    # ------------------------------------------------------- design-point selection
    def _selectPoint(self, utility, theta0, nRestarts, method, options, nCandidates, polish):
        """One design point: the minimiser of ``utility`` over the prior.  Under a process group the
        ``nCandidates`` draw is ONE global matrix (same random state on every rank), each rank sweeps its
        contiguous rows and the winners meet in a 16-byte-per-rank all-gather; the Nelder-Mead search is
        replicated and rank 0's answer kept."""
        scalarArgs = (self.y, self.gp, self._lnprior)
        ranks = self._ranks()
        if nCandidates is None:
            point, value = ut.minimizeObjective(utility, self.y, self.gp, sampleFn=self.priorSample,
                                                priorFn=self._lnprior, nRestarts=nRestarts, method=method,
                                                options=options, bounds=self.bounds, theta0=theta0,
                                                args=scalarArgs)
            return self._agree(point, value) if ranks is not None else (point, value)
        total = int(nCandidates)
        kind = ut.utilityKind(utility) if (ranks is not None or self.deviceCandidates) else None
        lo, hi = apdist.shard_bounds(total, ranks[1], ranks[0]) if ranks is not None else (0, total)
        if self.deviceCandidates:
            # the global matrix is a function of (seed, row) alone: this rank generates its rows in HBM, no host draw,
            # no H2D copy; the box ``bounds`` IS the prior here (priorSample is not consulted)
            seed = int(np.random.randint(0, 2 ** 31 - 1))
            mine = self.gp.box_candidates(hi - lo, self.bounds, seed, idx_offset=lo)
            row = lambda g: self.gp.box_candidates(1, self.bounds, seed, idx_offset=g).cpu().numpy()[0]   # noqa: E731
        else:
            draws = np.asarray(self.priorSample(total), dtype=float).reshape(total, -1)
            mine = draws if ranks is None else np.ascontiguousarray(draws[lo:hi])
            row = lambda g: np.array(draws[g])                                                              # noqa: E731
        if ranks is None and not self.deviceCandidates:
            point, value = ut.sweepObjective(utility, self.y, self.gp, mine, bounds=self.bounds)
        else:
            best, value = apdist.sharded_acquire(
                lambda offset: self.gp.acquire(self.y, mine, kind, bounds=self.bounds, idx_offset=offset,
                                               device_record=True),
                lo, group=self.group if ranks is not None else None,
                enabled=self.distributed if ranks is not None else False)     # (no ranks: this process's own record, no gather)
            if best < 0:
                raise RuntimeError("ERROR: Cannot find a valid solution: no candidate is allowed by the prior")
            point = row(best)
        if polish:
            point, value = ut.minimizeObjective(utility, self.y, self.gp,
                                                sampleFn=self.priorSample, priorFn=self._lnprior,
                                                nRestarts=1, method=method, options=options,
                                                bounds=self.bounds, theta0=point, args=scalarArgs)
            if ranks is not None:
                point, value = self._agree(point, value)
        return point, value

    def _absorbPoint(self, point, value):
        """Append (point, value) to the training set and move the GP onto it: same
        kernel / mean / white-noise objects and hyper-parameters (approx.py:693-717); the
        device GP extends its factor by one row instead of refactorising."""
        stack = np.vstack if self.theta.ndim > 1 else np.hstack
        self.theta = stack([self.theta, np.array(point)])
        self.y = np.hstack([self.y, value])
        hyper = self.gp.get_parameter_vector()
        grown = george.GP(kernel=self.gp.kernel, fit_mean=True, mean=self.gp.mean,
                          white_noise=self.gp.white_noise, fit_white_noise=False)
        grown.set_parameter_vector(hyper)
        if hasattr(grown, "_try_extend"):
            grown.compute(self.theta, previous=self.gp)
        else:
            grown.compute(self.theta)
        self.gp = grown
        return hyper

    def findNextPoint(self, theta0=None, computeLnLike=True, seed=None,
                      cache=True, gpOptions=None, gpP0=None, verbose=True,
                      nGPRestarts=1, nMinObjRestarts=5, gpMethod="powell",
                      minObjMethod="nelder-mead", minObjOptions=None,
                      runName="apRun", numNewPoints=1, optGPEveryN=1,
                      gpHyperPrior=gpUtils.defaultHyperPrior, args=None,
                      nCandidates=None, polish=False, deviceCandidates=None, **kwargs):
        """Select ``numNewPoints`` design points by minimising the (negative) utility;
        with ``computeLnLike`` evaluate the forward model at each, absorb it into the
        training set / GP and re-fit the hyper-parameters every ``optGPEveryN`` points
        (approx.py:527-754).  Returns ``theta`` or ``(theta, y)`` -- a single point /
        value when ``numNewPoints == 1``, arrays otherwise (approx.py:745-753).

        ``nCandidates`` replaces the restarted Nelder-Mead search by the fused device
        sweep over that many ``priorSample`` draws (box ``bounds`` as the prior);
        ``polish`` refines the sweep winner with one Nelder-Mead run;
        ``deviceCandidates=True`` draws the candidates on the device, uniformly in ``bounds``
        (counter-based Philox keyed by one integer from NumPy's global stream) instead of calling
        ``priorSample`` -- valid when the prior IS that box.
        """
        if deviceCandidates is not None:
            self.deviceCandidates = bool(deviceCandidates)
        assert isinstance(numNewPoints, int) and numNewPoints >= 1
        assert isinstance(optGPEveryN, int) and optGPEveryN >= 1
        if verbose and numNewPoints < optGPEveryN:
            print("WARNING: numNewPoints < optGPEveryN: the GP hyperparameters will not be "
                  "re-optimized during this call.")
        fwdArgs = () if args is None else args
        points, values = [], []
        ranks = self._ranks()
        chief = self._chief()
        verbose, cache = verbose and chief, cache and chief      # one rank talks and writes
        for count in range(numNewPoints):
            if ranks is not None:
                # one global random stream: the candidate matrix / restart draws below are the same on
                # every rank whatever the forward model (rank 0 only) did to rank 0's stream
                apdist.sync_random_state(0, self.group, enabled=self.distributed)
            if self.algorithm == "alternate":      # AGP, BAPE, AGP, ... (approx.py:656-661)
                self.utility = (ut.AGPUtility, ut.BAPEUtility)[count % 2]
            point, _ = self._selectPoint(self.utility, theta0, nMinObjRestarts, minObjMethod,
                                         minObjOptions, nCandidates, polish)
            points.append(point)
            if not computeLnLike:
                continue
            failure = None
            value = np.zeros(1)
            if chief:
                try:
                    like = self._lnlike(point, *fwdArgs, **kwargs)
                    like = like[0] if hasattr(like, "__iter__") else like      # (lnlike, blobs...) allowed
                    value = np.array([like + self._lnprior(point)], dtype=np.float64).reshape(1)
                except Exception as err:       # noqa: BLE001 -- raised on every rank just below
                    if ranks is None:
                        raise
                    failure = err
            if ranks is not None:
                # only rank 0 ran the forward model: if it raised there, every rank raises here instead of waiting
                # for a broadcast that never comes
                apdist.raise_together(failure, "the forward model (lnlike)", group=self.group, enabled=self.distributed)
            value = self._agree(value)             # the forward model ran once, on rank 0
            values.append(value)
            try:
                hyper = self._absorbPoint(point, value)
                if verbose:
                    print("hyperparameters", hyper)
                if cache:
                    self.gpPar.append(hyper)
                if count % optGPEveryN == 0:
                    self.optGP(seed=seed, method=gpMethod, options=gpOptions, p0=gpP0,
                               nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)
            except ValueError:
                raise ValueError("GP couldn't optimize! names %s, parameters %s, %d training points"
                                 % (self.gp.get_parameter_names(), self.gp.get_parameter_vector(),
                                    len(self.y)))
            if cache:
                np.savez(_cacheName(runName, "APFModelCache"), theta=self.theta, y=self.y)
        if numNewPoints == 1:
            points = points[0]
            values = values[0] if computeLnLike else values
        if computeLnLike:
            return np.asarray(points), np.asarray(values)
        return np.asarray(points)

    # ------------------------------------------------------------------ surrogate MCMC
    def _dumpChain(self, runName):
        path = _cacheName(runName, "")
        self.backends.append(path)
        blobs = self.sampler.get_blobs()
        np.savez(path, chain=self.sampler.get_chain(), log_prob=self.sampler.get_log_prob(),
                 blobs=np.array([]) if blobs is None else blobs)

    def _requireBoxPrior(self):
        """The on-device sampler (``GP.sample_ensemble``) knows ONE prior: constant inside ``self.bounds``,
        -inf outside.  ``_gpll`` returns ``mu(theta) + lnprior(theta)`` (approx.py:167-188), so any other
        ``lnprior`` -- a Gaussian, a tilted box -- would make the device chain sample a different posterior
        without a word.  ``lnprior`` is probed (own RandomState: the caller's NumPy stream is untouched) at
        the centre, at seeded points inside, just inside every corner and just outside every face: it must be
        one finite constant inside and non-finite outside, else ``ValueError``."""
        lo = np.array([b[0] for b in self.bounds], dtype=np.float64)
        hi = np.array([b[1] for b in self.bounds], dtype=np.float64)
        span = hi - lo
        rs = np.random.RandomState(20260304)
        inside = [0.5 * (lo + hi)] + list(lo + span * rs.uniform(0.0, 1.0, size=(8, len(lo))))
        for corner in range(min(2 ** len(lo), 16)):
            pick = np.array([(corner >> d) & 1 for d in range(len(lo))], dtype=np.float64)
            inside.append(lo + span * (1e-6 + pick * (1.0 - 2e-6)))
        vals = np.array([float(np.asarray(self._lnprior(p)).ravel()[0]) for p in inside])
        ok = np.all(np.isfinite(vals)) and np.all(np.abs(vals - vals[0]) <= 1e-12 * max(1.0, abs(vals[0])))
        outside = []
        for d in range(len(lo)):
            for sign, edge in ((-1.0, lo), (1.0, hi)):
                p = 0.5 * (lo + hi)
                p[d] = edge[d] + sign * 1e-2 * span[d]
                outside.append(p)
        with np.errstate(all="ignore"):
            out_vals = np.array([float(np.asarray(self._lnprior(p)).ravel()[0]) for p in outside])
        ok = ok and not np.any(np.isfinite(out_vals))
        if not ok:
            raise ValueError("runMCMC(onDevice=True) samples mu(theta) under the box prior self.bounds only; lnprior is not "
                             "constant inside / -inf outside these bounds (probed %d + %d points). Use onDevice=False "
                             "(batched=True keeps the GP on the device)." % (len(inside), len(outside)))

    def runMCMC(self, samplerKwargs=None, mcmcKwargs=None, runName="apRun",
                cache=True, estBurnin=True, thinChains=True, verbose=False,
                args=None, batched=True, onDevice=False, **kwargs):
        """Sample the GP-surrogate posterior with the stretch-move ensemble sampler and
        estimate burn-in / thinning (approx.py:757-859): ``(sampler, iburn, ithin)``.

        ``batched`` (default) evaluates each half-ensemble with ONE mean-only GP launch
        (:meth:`_gpllBatch`); ``batched=False`` calls :meth:`_gpll` once per walker as
        emcee does for the reference; ``onDevice=True`` runs the entire chain as one
        persistent kernel (``GP.sample_ensemble``) -- valid when ``lnprior`` is the box
        prior ``self.bounds`` (constant inside, -inf outside): checked (:meth:`_requireBoxPrior`,
        ``ValueError`` otherwise).
        With ``cache`` the chain goes to ``<runName>.npz`` (keys chain, log_prob, blobs)
        where the reference writes ``<runName>.h5``.
        """
        ranks = self._ranks()
        if ranks is not None:
            apdist.sync_random_state(0, self.group, enabled=self.distributed)
            verbose, cache = verbose and ranks[0] == 0, cache and ranks[0] == 0
        samplerKwargs, mcmcKwargs = mcmcUtils.validateMCMCKwargs(self, samplerKwargs,
                                                                 mcmcKwargs, verbose)
        if onDevice:
            self._requireBoxPrior()
        if onDevice or ranks is not None:
            # one ensemble per rank, seeded base + rank, gathered once along the walker axis
            # (no ranks -- no process group, or distributed=False inside somebody else's: the local chain, seed
            # ``base``, and no collective whatever group the process has open)
            base = np.random.randint(0, 2 ** 31 - 1)
            merged = apdist.replicated_ensembles(
                lambda seed: self._sampleReplica(seed, samplerKwargs, mcmcKwargs, args, kwargs,
                                                 batched, onDevice),
                seed=base, group=self.group if ranks is not None else None,
                enabled=self.distributed if ranks is not None else False)
            chain, logp, naccept = merged[0], merged[1], merged[2]
            self.sampler = emcee.DeviceChain({"chain": chain, "log_prob": logp, "naccept": naccept,
                                              "coords": chain[-1], "final_log_prob": logp[-1],
                                              "blobs": merged[3] if len(merged) > 3 else None})
        else:
            self.sampler = self._hostSampler(samplerKwargs, args, kwargs, batched)
            for _ in self.sampler.sample(**mcmcKwargs):
                pass
        if verbose:
            print("mcmc finished")
        if cache:
            self._dumpChain(runName)
        iburn, ithin = mcmcUtils.estimateBurnin(self.sampler, estBurnin=estBurnin,
                                                thinChains=thinChains, verbose=verbose)
        return self.sampler, iburn, ithin

    def _hostSampler(self, samplerKwargs, args, kwargs, batched, seed=None):
        setup = dict(samplerKwargs)
        if batched:
            setup["log_prob_fn"] = lambda pts, *a, **k: self._gpllBatch(pts)
            setup["vectorize"] = True
        if seed is not None:
            setup["seed"] = seed
        return emcee.EnsembleSampler(**setup, backend=None, args=args, kwargs=kwargs,
                                     blobs_dtype=[("lnprior", float)])

    def _sampleReplica(self, seed, samplerKwargs, mcmcKwargs, args, kwargs, batched, onDevice):
        """This rank's ensemble: ``(chain, log_prob, naccept[, blobs])`` for :func:`dist.replicated_ensembles`."""
        if onDevice:
            res = self.gp.sample_ensemble(self.y, mcmcKwargs["initial_state"], mcmcKwargs["iterations"],
                                          self.bounds, seed=seed)
            return res["chain"], res["log_prob"], res["naccept"]
        sampler = self._hostSampler(samplerKwargs, args, kwargs, batched, seed=seed)
        for _ in sampler.sample(**mcmcKwargs):
            pass
        blobs = sampler.get_blobs()
        out = (sampler.get_chain(), sampler.get_log_prob(), sampler._naccepted)
        return out if blobs is None else out + (blobs,)

    # ---------------------------------------------------------------------- outer loop
    def run(self, m=10, nmax=2, seed=None, timing=False, verbose=True,
            mcmcKwargs=None, samplerKwargs=None, estBurnin=False,
            thinChains=False, runName="apRun", cache=True, gpMethod="powell",
            gpOptions=None, gpP0=None, optGPEveryN=1, nGPRestarts=1,
            nMinObjRestarts=5, onlyLastMCMC=False, initGPOpt=True, kmax=3,
            gpHyperPrior=gpUtils.defaultHyperPrior, eps=1.0, convergenceCheck=False,
            minObjMethod="nelder-mead", minObjOptions=None, args=None,
            nCandidates=None, onDevice=False, batched=True, deviceCandidates=None, **kwargs):
        """BAPE / AGP outer loop (approx.py:229-524): ``nmax`` times, find ``m`` design
        points (re-fitting the GP every ``optGPEveryN``), sample the surrogate posterior,
        record burn-in / thinning, and -- with ``convergenceCheck`` -- stop once the
        marginal means have moved by less than ``eps`` previous standard deviations for
        ``kmax`` consecutive iterations.  (The reference honours that rule only when
        ``verbose`` is set, quirk Q4; here it does not depend on verbosity.)
        ``nCandidates`` switches the point search to the fused device sweep
        (``deviceCandidates``: drawn on the device, see :meth:`findNextPoint`);
        ``onDevice`` / ``batched`` are passed to :meth:`runMCMC`."""
        if convergenceCheck and onlyLastMCMC:
            raise RuntimeError("If convergenceCheck is True, must run an MCMC each iteration.\n"
                               "convergenceCheck = %d onlyLastMCMC = %d" % (convergenceCheck, onlyLastMCMC))
        verbose, cache = verbose and self._chief(), cache and self._chief()
        if cache:
            np.savez(_cacheName(runName, "APFModelCache"), theta=self.theta, y=self.y)
            self.gpPar = []
        if seed is not None:
            np.random.seed(seed)
        if timing:
            self.trainingTime, self.mcmcTime = [], []
        monitor = _MarginalMonitor(eps, kmax) if convergenceCheck else None
        if monitor is not None:
            self.marginalMeans, self.marginalStds = monitor.means, monitor.stds
            self.marginalZScores = monitor.zscores
        fit = dict(seed=seed, nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)
        if initGPOpt:
            self.optGP(method=gpMethod, options=gpOptions, p0=gpP0, **fit)
        for iteration in range(nmax):
            if verbose:
                print("Iteration: %d" % iteration)
            clock = time.time()
            self.findNextPoint(computeLnLike=True, cache=cache, gpMethod=gpMethod,
                               gpOptions=gpOptions, nMinObjRestarts=nMinObjRestarts,
                               optGPEveryN=optGPEveryN, numNewPoints=m,
                               minObjMethod=minObjMethod, minObjOptions=minObjOptions,
                               runName=runName, theta0=None, args=args, verbose=verbose,
                               nCandidates=nCandidates, deviceCandidates=deviceCandidates, **fit, **kwargs)
            if timing:
                self.trainingTime.append(time.time() - clock)
            if cache:
                np.savez(_cacheName(runName, "APGP"), gpParamNames=self.gp.get_parameter_names(),
                         gpParamValues=self.gpPar)
            if onlyLastMCMC and iteration != nmax - 1:
                self.sampler = None
                continue
            clock = time.time()
            _, iburn, ithin = self.runMCMC(samplerKwargs=samplerKwargs, mcmcKwargs=mcmcKwargs,
                                           runName="%s%d" % (runName, iteration), cache=cache,
                                           estBurnin=estBurnin, thinChains=thinChains,
                                           verbose=verbose, args=args, onDevice=onDevice,
                                           batched=batched, **kwargs)
            self.iburns.append(iburn)
            self.ithins.append(ithin)
            if timing:
                self.mcmcTime.append(time.time() - clock)
                if cache:
                    np.savez(_cacheName(runName, "APTiming"), trainingTime=self.trainingTime,
                             mcmcTime=self.mcmcTime)
            if monitor is None:
                continue
            converged = monitor.update(self.sampler.get_chain(discard=iburn, flat=True, thin=ithin))
            if cache:
                monitor.save(_cacheName(runName, "ConvergenceCache"))
            if converged:
                if verbose:
                    print("Approximate marginal posterior distributions converged: |z| = %s < eps = %e "
                          "for %d iterations" % (monitor.zscores[-1], eps, kmax))
                break

    # ----------------------------------------------------------------------------- MAP
    def findMAP(self, theta0=None, method="nelder-mead", options=None, nRestarts=15):
        """Maximum of the function the GP has learned: minimise minus the GP mean from
        ``nRestarts`` starts around ``theta0`` (default: the best training point)
        (approx.py:862-926).  Returns ``(MAP, MAPVal)``."""
        if theta0 is None:
            start = self.theta[np.argmax(self.y)]
        else:
            start = np.array(theta0).reshape(1, self.theta.shape[-1])
        if options is None and str(method).lower() == "nelder-mead":
            options = {"adaptive": True}

        def minusMean(x):
            return -(self._gpll(x)[0]) if np.isfinite(self._lnprior(x)) else np.inf

        if self._ranks() is not None:
            apdist.sync_random_state(0, self.group, enabled=self.distributed)
        best, value = ut.minimizeObjective(minusMean, self.y, self.gp, self.priorSample,
                                           self._lnprior, nRestarts=nRestarts, args=None,
                                           method=method, options=options, bounds=self.bounds,
                                           theta0=start)
        if self._ranks() is not None:
            best, value = self._agree(best, value)
        return best, -value

    # ---------------------------------------------------------- Bayesian optimisation
    def bayesOpt(self, nmax, theta0=None, tol=1.0e-3, kmax=3, seed=None,
                 verbose=True, runName="apRun", cache=True, gpMethod="powell",
                 gpOptions=None, gpP0=None, optGPEveryN=1, nGPRestarts=1,
                 nMinObjRestarts=5, initGPOpt=True, minObjMethod="nelder-mead",
                 gpHyperPrior=gpUtils.defaultHyperPrior, minObjOptions=None,
                 findMAP=True, args=None, nCandidates=None, deviceCandidates=None, **kwargs):
        """Bayesian optimisation (approx.py:929-1151): one design point per iteration by
        the object's utility (use algorithm="jones"), optionally the MAP of the GP mean
        after each, stop after ``kmax`` consecutive iterations whose best value changed
        by less than ``tol``.  Returns the reference's solution dictionary (thetaBest,
        valBest, thetas, vals, nev [, thetasMAP, valsMAP, thetaMAPBest, valMAPBest])."""
        verbose, cache = verbose and self._chief(), cache and self._chief()
        if cache:
            np.savez(_cacheName(runName, "APFModelCache"), theta=self.theta, y=self.y)
        if seed is not None:
            np.random.seed(seed)
        fit = dict(seed=seed, nGPRestarts=nGPRestarts, gpHyperPrior=gpHyperPrior)
        if initGPOpt:
            self.optGP(method=gpMethod, options=gpOptions, p0=gpP0, **fit)
        plateau = _PlateauMonitor(tol, kmax)
        history = {"thetas": [], "vals": [], "thetasMAP": [], "valsMAP": []}
        evaluations = 0
        for iteration in range(nmax):
            if verbose:
                print("Iteration: %d" % iteration)
            # re-fit on every optGPEveryN-th iteration: a period no single call can reach otherwise
            period = 1 if iteration % optGPEveryN == 0 else 99999999
            point, value = self.findNextPoint(computeLnLike=True, cache=cache, gpMethod=gpMethod,
                                              gpOptions=gpOptions, nMinObjRestarts=nMinObjRestarts,
                                              optGPEveryN=period, numNewPoints=1,
                                              minObjMethod=minObjMethod, minObjOptions=minObjOptions,
                                              runName=runName, args=args, verbose=verbose,
                                              nCandidates=nCandidates, deviceCandidates=deviceCandidates,
                                              **fit, **kwargs)
            evaluations = iteration + 1
            if verbose:
                print("Forward model evaluation at: ", point, ", function value: ", value)
            if cache:
                np.savez(_cacheName(runName, "APGP"), gpParamNames=self.gp.get_parameter_names(),
                         gpParamValues=self.gp.get_parameter_vector())
            top = int(np.argmax(self.y))
            history["thetas"].append(self.theta[top])
            history["vals"].append(self.y[top])
            if findMAP:
                mapPoint, mapValue = self.findMAP(theta0=theta0, method=minObjMethod,
                                                  options=minObjOptions, nRestarts=nMinObjRestarts)
                if verbose:
                    print("Current MAP solution: ", mapPoint, mapValue)
                history["thetasMAP"].append(mapPoint)
                history["valsMAP"].append(mapValue)
            if plateau.update(history["vals"][-1]):
                break
        soln = {"thetaBest": history["thetas"][-1], "valBest": history["vals"][-1],
                "thetas": np.asarray(history["thetas"]).squeeze(),
                "vals": np.asarray(history["vals"]).squeeze(), "nev": evaluations}
        if findMAP:
            soln["thetasMAP"] = np.asarray(history["thetasMAP"]).squeeze()
            soln["valsMAP"] = np.asarray(history["valsMAP"]).squeeze()
            top = int(np.argmax(soln["valsMAP"]))
            soln["thetaMAPBest"] = soln["thetasMAP"][top]
            soln["valMAPBest"] = soln["valsMAP"][top]
        return soln
