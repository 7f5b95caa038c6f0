# -*- coding: utf-8 -*-
"""
:py:mod:`gpUtils.py` - GP construction and hyper-parameter fitting on MI355X
----------------------------------------------------------------------------

Host-side mirror of the reference's ``approxposterior/gpUtils.py`` with the same
public names and argument meaning (``defaultHyperPrior`` :22-43, ``_nll`` :46-80,
``_grad_nll`` :83-111, ``defaultGP`` :114-181, ``optimizeGP`` :184-257).  The
``george`` objects the reference builds are replaced by the HIP-backed ones of
:py:mod:`approxposterior_amd.gp`; every objective evaluation SciPy requests is
one Gram + Cholesky + solve on the GPU through the C ABI.

Two things sit between SciPy and the device, neither of which changes a value SciPy sees: a memo of exact repeats
(Powell re-asks f at the head of every line search) and, for ``method="powell"``, a look-ahead -- the points of a
line search whose abscissae depend on no function value are evaluated together with the one asked for, side by side
on the idle compute units, and answered from the memo when SciPy gets to them (:func:`_powellAhead`).
"""

import sys
from collections import OrderedDict

import numpy as np
from scipy.optimize import minimize

from . import gp as george   # drop-in for the ``george`` names used below

__all__ = ["defaultHyperPrior", "defaultGP", "optimizeGP"]


def defaultHyperPrior(p):
    """Flat prior keeping every log hyper-parameter (all but the mean) within
    [-20, 20]; returns 0.0 inside, -inf outside (gpUtils.py:22-43)."""
    if (np.fabs(p)[1:] > 20).any():
        return -np.inf
    return 0.0


_MEMO_SIZE = 64
_LOCKSTEP_MIN_N = 512     # optimizeGP's restarts run in lock-step (batched device evaluations) from this many training points on


def _memoTable(gp, y):
    """The GP's table of already-evaluated ``_nll`` values for THIS training set and y, or None when the
    object cannot carry one.  SciPy's Powell asks for f at the current point again at the head of every
    line search (one in ten of all evaluations in BASELINE config 5); an exact repeat -- same bytes of
    ``p``, same fixed mean / white noise / yerr, same x and y -- is answered from here.  ``GP.compute``
    drops the table; objects without the ``_nllMemo`` slot (a real ``george.GP``) are never memoised."""
    if not hasattr(gp, "_nllMemo"):       # only a GP that drops the table in compute() carries one
        return None
    memo = gp._nllMemo
    yb = y.tobytes() if type(y) is np.ndarray and y.dtype == np.float64 else np.asarray(y, dtype=np.float64).tobytes()
    if memo is None or memo["y"] != yb:
        memo = gp._nllMemo = {"y": yb, "table": OrderedDict()}
    return memo["table"]


def _memoKey(p, gp):
    """``p`` fixes every fitted parameter; what it does not fix must be in the key too: a frozen mean / white noise,
    yerr, and the kernel OBJECT -- a caller may swap ``gp.kernel`` for another of the same parameter count, or edit a
    frozen attribute (a LinearKernel's ``order``), without a ``compute()`` in between.  For the two shapes
    ``defaultGP`` builds without a linear term the objects' identities say it all (they have no frozen attribute);
    anything else pays for the bytes of the evaluated kernel struct (``GP._factor_key``)."""
    yerr2 = gp._yerr2
    frozen = (None if gp.fit_mean else float(gp.mean.value),
              None if gp.fit_white_noise else float(gp.white_noise.value),
              yerr2 if type(yerr2) is float else np.asarray(yerr2, dtype=np.float64).tobytes())
    k = gp.kernel
    tk = type(k)
    if tk is george.ExpSquaredKernel:
        sig = id(k)
    elif tk is george.Product and type(k.k1) is george.ConstantKernel and type(k.k2) is george.ExpSquaredKernel:
        sig = (id(k), id(k.k1), id(k.k2))
    else:
        sig = gp._factor_key()
    return np.asarray(p, dtype=np.float64).tobytes(), frozen, sig


def _brentGolden(brent, x, a, b):
    """Abscissa of a golden-section step of ``scipy.optimize._optimize.Brent.optimize`` from the state (x; a, b), with
    its own expressions; None when its convergence test would stop the loop instead."""
    tol1 = brent.tol * np.abs(x) + brent._mintol
    tol2 = 2.0 * tol1
    xmid = 0.5 * (a + b)
    if np.abs(x - xmid) < (tol2 - 0.5 * (b - a)):
        return None
    deltax = (a - x) if x >= xmid else (b - x)
    rat = brent._cg * deltax
    if np.abs(rat) < tol1:
        return x + tol1 if rat >= 0 else x - tol1
    return x + rat


def _brentAhead(brent, xa, xb, xc):
    """The first TWO abscissae of ``Brent.optimize`` on the bracket (xa, xb, xc) -- ``(u1, u2 if f(u1) > f(xb), u2 if
    f(u1) <= f(xb))`` -- none of which depends on a function value: the loop starts from x = w = v = xb with deltax = 0,
    so its first step is a golden-section one; after it either v == x (f(u1) > fx: u1 becomes w, an end of the interval
    moves to u1) or w == v (f(u1) <= fx: u1 becomes x), and with two of the three points equal the parabola's numerator
    and denominator are both exactly 0, the fit is rejected and the second step is a golden-section one too, from a state
    that depends only on which of the two it was.  (The third step interpolates three distinct points: values needed.)"""
    x = xb
    a, b = (xa, xc) if xa < xc else (xc, xa)
    u1 = _brentGolden(brent, x, a, b)
    if u1 is None:
        return None, None, None
    ag, bg = (u1, b) if u1 < x else (a, u1)            # f(u1) > fx: the interval's end on u1's side moves to u1
    al, bl = (x, b) if u1 >= x else (a, x)             # f(u1) <= fx: x moves to u1, the old x closes the other side
    return u1, _brentGolden(brent, x, ag, bg), _brentGolden(brent, u1, al, bl)


_GOLD = 1.618034          # scipy.optimize._optimize.bracket's ``_gold`` (read from its frame where one is at hand; pinned by a test)


_AHEAD_CACHE = {}


def _firstAbscissae(brent, gold):
    """The value-independent abscissae of a line search that brackets from (0, 1) -- third point, Brent's first step, its second
    for either outcome; for the swapped bracket (f(0) < f(1)) and the unswapped one -- as functions of SciPy's constants
    only: computed once per (gold, tol, _cg, _mintol)."""
    key = (float(gold), float(brent.tol), float(brent._cg), float(brent._mintol))
    got = _AHEAD_CACHE.get(key)
    if got is None:
        xa, xb = np.asarray([0.0, 1.0])
        cB, cA = xa + gold * (xa - xb), xb + gold * (xb - xa)
        got = _AHEAD_CACHE[key] = (xb, cB, _brentAhead(brent, xb, xa, cB), cA, _brentAhead(brent, xa, xb, cA))
        if len(_AHEAD_CACHE) > 64:
            _AHEAD_CACHE.pop(next(iter(_AHEAD_CACHE)))
    return got


def _nextSearchAhead(site, brent, base, xi, x):
    """The first points of the NEXT line search of ``_minimize_powell``, assuming the current one ends at Brent's current
    best abscissa ``x`` -- which is what a tolerance step x +- tol1 announces (nine line searches in ten end with one, and
    its value is rarely better than f(x)).  SciPy returns ``p + x * xi`` as the new point and goes on along ``direc[i + 1]``:
    f(0) there is f(x) (a memo hit), then f(1), the third bracket point and Brent's first two steps -- the abscissae of
    :func:`_powellAhead`, none of which depends on the point.  After the last direction of a sweep the next evaluation is
    the extrapolated point ``x + (x - x1)``.  Only for the searches of the direction loop (``xi`` is a row of ``direc``);
    anything else: nothing."""
    fr = site
    for name in ("_minimize_scalar_brent", "_recover_from_bracket_error", "_linesearch_powell", "_minimize_powell"):
        fr = fr.f_back
        if fr is None or fr.f_code.co_name != name:
            return []
    loc = fr.f_locals
    direc, i, n = loc["direc"], loc["i"], loc["N"]
    if loc.get("lower_bound") is not None or loc.get("upper_bound") is not None:
        return []
    if not (0 <= i < n and np.shares_memory(xi, direc) and np.array_equal(direc[i], xi)):
        return []
    new = base + x * xi                                   # _linesearch_powell: xi = alpha_min * xi; return fret, p + xi, xi
    if i + 1 == n:
        direc1 = new - loc["x1"]                          # the extrapolated point (lmax = 1 without bounds)
        return [new + 1 * direc1]
    nxt = direc[i + 1]
    if not np.any(nxt):
        return []
    xb, cB, uB, _, _ = _firstAbscissae(brent, _GOLD)
    return [new + a_ * nxt for a_ in (xb, cB, uB[0], uB[1]) if a_ is not None]


def _powellAhead(width, explain=False):
    """The points SciPy's Powell line search is going to ask for NEXT, known before any value is (gpUtils.py:238:
    ``minimize(_nll, method="powell")``).  ``_linesearch_powell`` minimises ``myfunc(alpha) = f(p + alpha xi)`` with
    Brent's method, which first brackets from (0, 1): f(0) is the current point (a memo hit), then f(1), then
    f(-1.618034) if f(0) < f(1) (four line searches in five: a unit step is usually too long) else f(2.618034); when that
    third value closes the bracket (nine times in ten), Brent's first two steps are golden-section steps whose abscissae
    depend only on which bracket it is and on whether the first step improved (:func:`_brentAhead`).  So when f(1) is
    asked for, the abscissae of up to three further evaluations are known up to a handful of cases; an evaluation
    occupies a quarter of the chip, so the likely ones ride along with f(1) in ONE batched device call and the next
    answers come from the memo -- SciPy sees the same values in the same order, bit for bit.  The same at the two later
    points where a miss can still look ahead: the third bracket point (its Brent steps), Brent's first step (its second) and a
    tolerance step x +- tol1 near the end of the search (the same step to the other side, and the first points of the NEXT
    line search from the point this one is about to return: :func:`_nextSearchAhead`).

    The abscissae are recomputed with SciPy's own constants and expressions from the frames of the caller (``bracket`` <-
    ``Brent.get_bracket_info`` <- ``Brent.optimize``: ``_gold``, ``tol``, ``_cg``, ``_mintol``) and the points as
    ``p + alpha * xi`` from ``myfunc``'s closure, exactly as ``myfunc`` forms them.  Anything unexpected in those
    frames (another SciPy, another method, another call site): no look-ahead, nothing else changes.  A wrong guess
    costs idle-CU work only: a point that is never asked for is never used.

    Returns up to ``width`` points (most likely first), or None; with ``explain`` (tests) ``(call site, points)``, the call
    site one of "f(1)", "third bracket point", "Brent's first step", "tolerance step"."""
    try:
        line = sys._getframe(2)                        # _nll's caller: SciPy's function wrapper <- myfunc(alpha)
        for _ in range(4):                             # (a caller's own thin wrapper around _nll may sit in between)
            if line is None or line.f_code.co_name == "myfunc":
                break
            line = line.f_back
        if line is None or line.f_code.co_name != "myfunc":
            return None
        site = line.f_back
        if site is None:
            return None
        loc = line.f_locals
        alpha, base, xi = loc["alpha"], loc["p"], loc["xi"]
        where = site.f_code.co_name
        abscissae, extra, kind = [], [], None
        if where == "bracket":
            bl = site.f_locals
            info = site.f_back
            brent = info.f_locals.get("self") if info is not None and info.f_code.co_name == "get_bracket_info" else None
            if "fa" not in bl or "fc" in bl or "w" in bl:
                return None
            if "fb" not in bl:
                # fb = func(xb) of a (0, 1) start
                if not (alpha == 1.0 and bl["xa"] == 0.0 and bl["xb"] == 1.0):
                    return None
                gold, xa, xb = bl["_gold"], bl["xa"], bl["xb"]
                if gold != _GOLD:
                    return None                        # (another SciPy: _nextSearchAhead's constant would be wrong too)
                # f(0) < f(1): SciPy swaps (xa, xb) and goes on to -1.618034 -- that case first; else no swap, 2.618034
                if brent is not None:
                    _, cB, uB, cA, uA = _firstAbscissae(brent, gold)
                else:
                    cB, cA = xa + gold * (xa - xb), xb + gold * (xb - xa)
                    uB = uA = (None, None, None)
                abscissae = [cB, uB[0], uB[1], cA, uA[0], uB[2], uA[1], uA[2]]
                kind = "f(1)"
            else:
                # fc = func(xc): if it closes the bracket, Brent's first two steps on (xa, xb, xc)
                if brent is None or alpha != bl["xc"]:
                    return None
                abscissae = list(_brentAhead(brent, bl["xa"], bl["xb"], bl["xc"]))
                kind = "third bracket point"
        elif where == "optimize":
            ol = site.f_locals
            brent = ol.get("self")
            if brent is None:
                return None
            if ol.get("iter") == 0:
                if not (ol["x"] == ol["w"] == ol["v"]):
                    return None
                u1, ugt, ule = _brentAhead(brent, ol["xa"], ol["xb"], ol["xc"])
                if u1 is None or u1 != alpha:
                    return None
                abscissae = [ugt, ule]                 # Brent's first step: its second, for either outcome
                kind = "Brent's first step"
            else:
                # a TOLERANCE step: the interpolated step was shorter than tol1 (or too close to an end of the interval), so
                # SciPy evaluates x + tol1 or x - tol1 -- the search is closing in on x.  If the value there is no better
                # (the usual case) the interval's end moves to it and the next step is, more often than not, the same step
                # to the other side (after which the convergence test stops the loop): one more abscissa that is known now
                x, u, tol1 = ol["x"], ol["u"], ol["tol1"]
                if u != alpha:
                    return None
                if u == x + tol1:
                    abscissae = [x - tol1]
                elif u == x - tol1:
                    abscissae = [x + tol1]
                else:
                    return None
                extra = _nextSearchAhead(site, brent, base, xi, x)
                kind = "tolerance step"
        else:
            return None
        points = ([base + a_ * xi for a_ in abscissae if a_ is not None] + extra)[:width]
        return (kind, points) if explain else points
    except (KeyError, AttributeError, ValueError, TypeError):
        return None


def _nelderMeadAhead(width):
    """The same idea for ``method="nelder-mead"`` (the reference's other derivative-free choice, gpUtils.py:233-236): within
    one iteration of ``_minimize_neldermead`` the reflected point is followed by the expanded, the outside-contracted or
    the inside-contracted point -- or by none -- depending on f(reflected), but all three are functions of the simplex
    alone; so are the N + 1 vertices of the initial simplex and the N vertices of a shrink.  Recomputed with SciPy's own
    expressions from its frame; unbounded searches only.  Returns up to ``width`` points or None."""
    try:
        fr = sys._getframe(2)                          # _nll's caller: SciPy's function wrapper <- _minimize_neldermead
        for _ in range(4):
            if fr is None or fr.f_code.co_name == "_minimize_neldermead":
                break
            fr = fr.f_back
        if fr is None or fr.f_code.co_name != "_minimize_neldermead":
            return None
        loc = fr.f_locals
        if loc.get("bounds") is not None:
            return None
        asked = sys._getframe(1).f_locals["p"]
        sim = loc["sim"]
        if "xbar" not in loc:
            # the initial simplex: fsim[k] = func(sim[k]) for k = 0 .. N
            k = loc.get("k")
            if k is None or not np.array_equal(sim[k], asked):
                return None
            return [sim[j] for j in range(k + 1, min(len(sim), k + 1 + width))]
        if np.array_equal(loc["xr"], asked) and not loc.get("doshrink"):
            xbar, rho, chi, psi = loc["xbar"], loc["rho"], loc["chi"], loc["psi"]
            worst = sim[-1]
            if not np.array_equal((1 + rho) * xbar - rho * worst, asked):
                return None                            # (a stale xr of an earlier iteration)
            return [(1 + rho * chi) * xbar - rho * chi * worst,          # expansion
                    (1 - psi) * xbar + psi * worst,                      # inside contraction
                    (1 + psi * rho) * xbar - psi * rho * worst][:width]  # outside contraction
        if loc.get("doshrink") and "j" in loc and np.array_equal(sim[loc["j"]], asked):
            sigma, j = loc["sigma"], loc["j"]
            return [sim[0] + sigma * (sim[m] - sim[0]) for m in range(j + 1, min(len(sim), j + 1 + width))]
        return None
    except (KeyError, AttributeError, ValueError, TypeError, IndexError):
        return None


def _nll(p, gp, y, priorFn=None):
    """Negative marginal log-likelihood at hyper-parameters ``p``; +inf where the
    prior forbids ``p`` or the Gram matrix is not positive definite
    (gpUtils.py:46-80)."""
    if priorFn is not None and not np.isfinite(priorFn(p)):
        return np.inf
    try:
        gp.set_parameter_vector(p)
    except np.linalg.LinAlgError:
        return np.inf
    table = _memoTable(gp, y)
    if table is not None:
        key = _memoKey(p, gp)      # (after set_parameter_vector: a fitted mean / white noise is part of p)
        hit = table.get(key)
        if hit is not None:
            table.move_to_end(key)
            return hit
        width = gp.lookahead_width() if hasattr(gp, "lookahead_width") else 0
        if width > 0:
            ahead = _powellAhead(width)
            if ahead is None:
                ahead = _nelderMeadAhead(width)
            if ahead and priorFn is not None:
                if priorFn is defaultHyperPrior:                       # (its own expression, on all guesses at once)
                    keep = ~(np.fabs(np.array(ahead))[:, 1:] > 20).any(axis=1)
                    if not keep.all():
                        ahead = [q for q, k in zip(ahead, keep) if k]
                else:
                    ahead = [q for q in ahead if np.isfinite(priorFn(q))]
            if ahead:
                vals = gp.nll_batch(np.array([p] + ahead), y)      # entry b: what _nll(P[b]) returns, bit for bit
                for q, v in zip(ahead, vals[1:]):
                    table[_memoKey(q, gp)] = float(v)
                val = table[key] = float(vals[0])
                while len(table) > _MEMO_SIZE:
                    table.popitem(last=False)
                return val
    ll = gp.log_likelihood(y, quiet=True)
    val = -ll if np.isfinite(ll) else np.inf
    if table is not None:
        table[key] = val
        if len(table) > _MEMO_SIZE:
            table.popitem(last=False)
    return val


def _grad_nll(p, gp, y, priorFn=None):
    """Gradient of :func:`_nll` (gpUtils.py:83-111).  As in the reference it does
    NOT set ``p`` itself: SciPy always evaluates ``_nll(p)`` first."""
    if priorFn is not None and not np.isfinite(priorFn(p)):
        return np.full_like(p, np.inf)
    return -gp.grad_log_likelihood(y, quiet=True)


def defaultGP(theta, y, order=None, white_noise=-12, fitAmp=False):
    """Squared-exponential GP with a seeded-random initial metric, optional
    amplitude ``var(y)``, constant mean ``median(y)`` and fixed white noise,
    factorised on the GPU (gpUtils.py:114-181).

    ``order`` adds ``(var(y)/10) * LinearKernel(log_gamma2=initialMetric[0], order)``
    as the reference does (gpUtils.py:167-173); integer orders only on the device.
    """
    theta = np.asarray(theta).squeeze()
    y = np.asarray(y).squeeze()
    ndim = 1 if theta.ndim <= 1 else theta.shape[-1]

    # same RNG call as the reference: the goldens depend on the draw order
    initialMetric = np.fabs(np.random.randn(ndim))
    kernel = george.kernels.ExpSquaredKernel(metric=initialMetric, ndim=ndim)
    if fitAmp:
        kernel = np.var(y) * kernel
    if order is not None:
        kernel = kernel + (np.var(y) / 10.0) * george.kernels.LinearKernel(
            log_gamma2=initialMetric[0], order=order, bounds=None, ndim=ndim)
    gp = george.GP(kernel=kernel, fit_mean=True, mean=np.median(y),
                   white_noise=white_noise, fit_white_noise=False)
    gp.compute(theta)
    return gp


class _LockStep(object):
    """Rendezvous that turns the objective calls of several concurrently running SciPy
    optimisers into batched device calls: a worker posts its point and blocks; when every
    still-running worker has posted, the last arrival evaluates the whole batch with
    ``batchFn`` and wakes the others.  Each optimiser sees exactly the values it would see
    running alone, so the restarts' trajectories do not depend on the batching."""

    def __init__(self, nWorkers, batchFn):
        import threading
        self._cv = threading.Condition()
        self._active = nWorkers
        self._batchFn = batchFn
        self._pending = {}
        self._results = {}

    def _flush(self):
        ids = sorted(self._pending)
        pts = [self._pending[i] for i in ids]
        self._pending = {}
        try:
            vals = self._batchFn(pts)
        except BaseException as err:          # hand the failure to every waiting worker
            vals = [err] * len(ids)
        for i, v in zip(ids, vals):
            self._results[i] = v
        self._cv.notify_all()

    def evaluate(self, wid, p):
        with self._cv:
            self._pending[wid] = np.array(p, dtype=np.float64, copy=True)
            if len(self._pending) >= self._active:
                self._flush()
            while wid not in self._results:
                self._cv.wait()
            v = self._results.pop(wid)
        if isinstance(v, BaseException):
            raise v
        return v

    def retire(self, wid):
        with self._cv:
            self._active -= 1
            if self._pending and len(self._pending) >= self._active:
                self._flush()


def _minimizeLockStep(gp, y, x0s, method, options, priorFn):
    """The restarts of :func:`optimizeGP` as concurrent SciPy runs (one thread each) whose
    ``_nll`` evaluations are served in lock-step by ``gp.nll_batch`` -- one batched
    Gram + Cholesky per round instead of one per restart."""
    import threading
    step = _LockStep(len(x0s), lambda pts: gp.nll_batch(np.array(pts), y))
    sols, errs = [None] * len(x0s), [None] * len(x0s)

    def work(k):
        seen = OrderedDict()      # this restart's exact repeats (Powell's line-search heads), as _nll's table

        def fn(p):
            # the prior gate of _nll (gpUtils.py:68-70) needs no device work
            if priorFn is not None and not np.isfinite(priorFn(p)):
                return np.inf
            key = np.asarray(p, dtype=np.float64).tobytes()
            val = seen.get(key)
            if val is None:
                val = seen[key] = float(step.evaluate(k, p))
                if len(seen) > _MEMO_SIZE:
                    seen.popitem(last=False)
            return val
        try:
            sols[k] = minimize(fn, x0s[k], method=method, jac=None, bounds=None, options=options)["x"]
        except BaseException as err:
            errs[k] = err
        finally:
            step.retire(k)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(x0s))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for err in errs:
        if err is not None:
            raise err
    return sols


def _startPoint(gp, y, p0):
    """Start point of one restart, drawn exactly as the reference does (gpUtils.py:224-230)."""
    if p0 is None:
        return [np.median(y)] + [np.random.randn() for _ in range(len(gp.get_parameter_vector()) - 1)]
    return np.array(p0) + np.min(p0) * 1.0e-3 * np.random.randn(len(p0))


def _runRestarts(gp, y, x0s, method, options, gpHyperPrior, batchRestarts):
    """Minimise ``_nll`` from every start point of ``x0s``: ``(solutions, mll)``.  Derivative-free methods
    with several starts run concurrently with batched device evaluations; everything else is the
    reference's sequential loop body (gpUtils.py:232-247)."""
    derivativeFree = method in ["nelder-mead", "powell", "cg"]
    # (below _LOCKSTEP_MIN_N training points one evaluation is a single fused launch of 30-160 us: the rendezvous of the
    # lock-step threads costs more than the batch saves -- tools/optgp_restarts_time.py, three Powell restarts: 16.7 against
    # 32.8 ms at N = 50, 54 against 60 at N = 400, 76 against 60 at N = 600, 114 against 75 at N = 1152.  Same solutions
    # either way: every restart sees the values it would see alone.)
    if (batchRestarts and len(x0s) > 1 and derivativeFree and hasattr(gp, "nll_batch")
            and (batchRestarts == "always" or len(y) >= _LOCKSTEP_MIN_N)):
        res = _minimizeLockStep(gp, y, x0s, method, options, gpHyperPrior)
        # marginal likelihood at each solution: one more batch (the sequential loop's
        # set_parameter_vector + recompute + log_likelihood, gpUtils.py:243-247)
        return res, -gp.nll_batch(np.array(res), y)
    res, mll = [], []
    for x0 in x0s:
        jac = None if derivativeFree else _grad_nll
        sol = minimize(_nll, x0, args=(gp, y, gpHyperPrior), method=method,
                       jac=jac, bounds=None, options=options)["x"]
        res.append(sol)
        gp.set_parameter_vector(sol)
        gp.recompute()
        mll.append(gp.log_likelihood(y, quiet=True))
    return res, np.array(mll, dtype=np.float64)


def optimizeGP(gp, theta, y, seed=None, nGPRestarts=1, method="powell",
               options=None, p0=None, gpHyperPrior=defaultHyperPrior, batchRestarts=True,
               distributed=None, group=None):
    """Maximise the marginal log-likelihood over the GP hyper-parameters with
    ``nGPRestarts`` SciPy runs and keep the best (gpUtils.py:184-257).  ``seed``
    and ``theta`` are accepted and unused, as in the reference (quirk Q6).

    With ``batchRestarts`` (default) and a derivative-free ``method`` the restarts run
    concurrently and their ``_nll`` evaluations are batched on the device -- from 512 training
    points on, where a batch saves more than the threads' rendezvous costs (``"always"``:
    whatever the size); start points, per-restart trajectories and the selected optimum are
    those of the sequential loop (the optimisers draw no random numbers, and a batched
    evaluation is bit-identical to a single one).  ``batchRestarts=False`` runs the
    reference's sequential loop.

    Under an initialised ``torch.distributed`` group (``distributed`` None / True; one process per
    GPU, NumPy's global random state identical on every rank -- ``dist.sync_random_state``) every rank
    draws ALL start points, runs restarts ``rank, rank + world, ...`` on its own GPU, and one all-gather
    of ``(mll, p)`` per restart gives every rank the same optimum (``dist.spread_restarts``)."""
    from . import dist as apdist
    # all start points first, in the reference's draw order: its loop interleaves the draws with the
    # minimisations, but SciPy's optimisers draw no random numbers
    x0s = [_startPoint(gp, y, p0) for _ in range(nGPRestarts)]
    if apdist.context(group, distributed) is None:
        res, mll = _runRestarts(gp, y, x0s, method, options, gpHyperPrior, batchRestarts)
    else:
        def runMine(indices):
            sols, vals = _runRestarts(gp, y, [x0s[i] for i in indices], method, options,
                                      gpHyperPrior, batchRestarts)
            return list(zip(vals, sols))
        mll, res = apdist.spread_restarts(nGPRestarts, runMine, len(gp.get_parameter_vector()), group,
                                          enabled=distributed)
    best = int(np.argmax(mll))
    gp.set_parameter_vector(res[best])
    gp.recompute()
    return gp
