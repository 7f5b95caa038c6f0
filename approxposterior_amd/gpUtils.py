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
"""

import numpy as np
from scipy.optimize import minimize

from . import gp as george   # drop-in for the ``george`` names used below

__all__ = ["defaultHyperPrior", "defaultGP", "optimizeGP"]


def defaultHyperPrior(p):
    """Flat prior keeping every log hyper-parameter (all but the mean) within
    [-20, 20]; returns 0.0 inside, -inf outside (gpUtils.py:22-43)."""
    if np.any(np.fabs(p)[1:] > 20):
        return -np.inf
    return 0.0


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
    ll = gp.log_likelihood(y, quiet=True)
    return -ll if np.isfinite(ll) else np.inf


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


def optimizeGP(gp, theta, y, seed=None, nGPRestarts=1, method="powell",
               options=None, p0=None, gpHyperPrior=defaultHyperPrior):
    """Maximise the marginal log-likelihood over the GP hyper-parameters with
    ``nGPRestarts`` SciPy runs and keep the best (gpUtils.py:184-257).  ``seed``
    and ``theta`` are accepted and unused, as in the reference (quirk Q6)."""
    res, mll = [], []
    for _ in range(nGPRestarts):
        if p0 is None:
            x0 = [np.median(y)] + [np.random.randn() for _ in range(len(gp.get_parameter_vector()) - 1)]
        else:
            x0 = np.array(p0) + np.min(p0) * 1.0e-3 * np.random.randn(len(p0))
        jac = None if method in ["nelder-mead", "powell", "cg"] else _grad_nll
        sol = minimize(_nll, x0, args=(gp, y, gpHyperPrior), method=method,
                       jac=jac, bounds=None, options=options)["x"]
        res.append(sol)
        gp.set_parameter_vector(sol)
        gp.recompute()
        mll.append(gp.log_likelihood(y, quiet=True))
    best = int(np.argmax(mll))
    gp.set_parameter_vector(res[best])
    gp.recompute()
    return gp
