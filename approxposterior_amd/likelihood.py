# -*- coding: utf-8 -*-
"""
:py:mod:`likelihood.py` - test objectives used as fixtures
----------------------------------------------------------

Own restatements of the reference's small test functions
(approxposterior/likelihood.py): the Wang & Li (2017) Rosenbrock posterior
(:26-110), the 1-D Bayesian-optimisation function (:120-170) and the 2-D sphere
(:180-240).  They are trivial scalar functions kept only because BASELINE
configs C1/C2 and the reference's known-answer tests are built from them; names
and call signatures match the reference so scripts can switch imports.
All sampling goes through NumPy's legacy global RandomState exactly as the
reference does (its goldens depend on the draw order).
"""

import numpy as np

__all__ = ["rosenbrockLnlike", "rosenbrockLnprior", "rosenbrockSample",
           "rosenbrockLnprob", "testBOFn", "testBOFnSample", "testBOFnLnPrior",
           "sphereLnlike", "sphereSample", "sphereLnprior"]


def _rosen(x):
    """sum_i 100 (x_{i+1} - x_i^2)^2 + (1 - x_i)^2 (scipy.optimize.rosen)."""
    x = np.asarray(x, dtype=float)
    return np.sum(100.0 * (x[1:] - x[:-1] ** 2.0) ** 2.0 + (1.0 - x[:-1]) ** 2.0, axis=0)


def rosenbrockLnlike(theta):
    """ln L = -rosen(theta) / 100 (likelihood.py:26-41)."""
    return -_rosen(theta) / 100.0


def rosenbrockLnprior(theta):
    """Uniform prior on [-5, 5]^D: 0 inside, -inf outside (likelihood.py:44-64)."""
    return -np.inf if np.any(np.fabs(theta) > 5) else 0.0


# A prior may carry a vectorised twin as its ``batch`` attribute -- f.batch(T) == [f(t) for t in T] for T of shape (W, D),
# same values -- which the walker-ensemble log-probability (ApproxPosterior._gpllBatch) calls once per half-step instead
# of f once per walker (the README example: 4e5 scalar calls, 1 s of its 6.6).  User priors without one work as before.
rosenbrockLnprior.batch = lambda T: np.where(np.any(np.fabs(T) > 5, axis=1), -np.inf, 0.0)


def rosenbrockSample(n=1, dim=2):
    """n draws from U[-5, 5]^dim, squeezed (likelihood.py:67-85)."""
    return np.random.uniform(low=-5, high=5, size=(n, dim)).squeeze()


def rosenbrockLnprob(theta):
    """ln prior + ln likelihood, -inf outside the prior (likelihood.py:88-110)."""
    lp = rosenbrockLnprior(theta)
    if not np.isfinite(lp):
        return -np.inf
    return lp + rosenbrockLnlike(theta)


def testBOFn(theta):
    """-sin(3 t) - t^2 + 0.7 t (likelihood.py:120-128)."""
    theta = np.asarray(theta)
    return -np.sin(3 * theta) - theta ** 2 + 0.7 * theta


def testBOFnSample(n=1):
    """n draws from U[-1, 2] (the code's range; the reference docstring says
    [-2, 1]) (likelihood.py:131-147)."""
    return np.random.uniform(low=-1, high=2, size=(n, 1)).squeeze()


def testBOFnLnPrior(theta):
    """Uniform prior on [-1, 2] (likelihood.py:150-170)."""
    return -np.inf if (np.any(theta < -1) or np.any(theta > 2)) else 0.0


testBOFnLnPrior.batch = lambda T: np.where(np.any(T < -1, axis=1) | np.any(T > 2, axis=1), -np.inf, 0.0)


def sphereLnlike(theta):
    """-sum theta^2 (likelihood.py:180-199)."""
    theta = np.asarray(theta)
    return -np.sum(theta ** 2)


def sphereSample(n=1):
    """n draws from U[-2, 2]^2 (likelihood.py:202-219)."""
    return np.random.uniform(low=-2, high=2, size=(n, 2)).squeeze()


def sphereLnprior(theta):
    """Uniform prior on [-2, 2]^D (likelihood.py:222-240)."""
    return -np.inf if np.any(np.fabs(theta) > 2) else 0.0


sphereLnprior.batch = lambda T: np.where(np.any(np.fabs(T) > 2, axis=1), -np.inf, 0.0)
