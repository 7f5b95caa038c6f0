# -*- coding: utf-8 -*-
"""
:py:mod:`george_extras.py` - members of ``george.GP`` that approxposterior never calls
-------------------------------------------------------------------------------------

OFF the hot path (none of these is a row): ``apply_inverse``, ``get_matrix``, the covariance
form of ``predict`` (george's default ``return_cov=True``; the reference only ever asks for the mean or
``return_var=True``: approx.py:178, utility.py:131,178,224) and the ``nll`` / ``lnlikelihood`` aliases.  Kept for
scripts written against george that hold the GP object directly; mixed into :class:`approxposterior_amd.gp.GP`.
The kernel matrices come from the HIP cross-kernel of the row append (``apgp_kernel_cross``); the dense solves and the
(M, N) x (N, M) product are plain library calls through torch (rocBLAS) -- acceptable here, never on the path.
"""

import ctypes

import numpy as np

from . import _lib


class GeorgeExtras(object):
    def apply_inverse(self, y):
        """K^-1 y for a vector (N,) or a matrix (N, k) -- george ``GP.apply_inverse`` (cho_solve against the
        resident factor; a library triangular solve pair on the device)."""
        self.recompute()
        torch, dev, lib = self._rt()
        b = np.ascontiguousarray(y, dtype=np.float64)
        n = len(self._x)
        if b.shape[0] != n:
            raise ValueError("Dimension mismatch")
        with self._on(torch, dev):
            bd = torch.from_numpy(b.reshape(n, -1)).to(dev)
            out = torch.cholesky_solve(bd, torch.tril(self._L[:n, :n]), upper=False)   # (bytes above the diagonal are undefined)
            return out.cpu().numpy().reshape(b.shape)

    def get_matrix(self, x1, x2=None):
        """The kernel matrix k(x1, x2) (x2 = None: k(x1, x1)), without the white-noise diagonal -- george
        ``GP.get_matrix``; evaluated by the HIP cross-kernel."""
        torch, dev, lib = self._rt()
        a1 = np.ascontiguousarray(self.parse_samples(x1), dtype=np.float64)
        a2 = a1 if x2 is None else np.ascontiguousarray(self.parse_samples(x2), dtype=np.float64)
        with self._on(torch, dev):
            st = self._stream(torch)
            ks = self._kernel_struct()
            d1 = torch.from_numpy(a1).to(dev)
            d2 = d1 if x2 is None else torch.from_numpy(a2).to(dev)
            out = torch.empty((len(a1), len(a2)), dtype=torch.float64, device=dev)
            _lib.check(lib.apgp_kernel_cross(d1.data_ptr(), len(a1), d2.data_ptr(), len(a2), ctypes.byref(ks),
                                             out.data_ptr(), len(a2), st), "apgp_kernel_cross")
            return out.cpu().numpy()

    def nll(self, vector, y, quiet=True):
        """george ``GP.nll``: -log_likelihood at ``vector`` (the parameters stay set, as in george)."""
        self.set_parameter_vector(vector)
        if not quiet:
            return -self.log_likelihood(y, quiet=False)
        ll = self.log_likelihood(y, quiet=True)
        return -ll if np.isfinite(ll) else np.inf

    def grad_nll(self, vector, y, quiet=True):
        """george ``GP.grad_nll``: -grad_log_likelihood at ``vector``."""
        self.set_parameter_vector(vector)
        return -self.grad_log_likelihood(y, quiet=quiet)

    def lnlikelihood(self, y, quiet=False):
        return self.log_likelihood(y, quiet=quiet)

    def grad_lnlikelihood(self, y, quiet=False):
        return self.grad_log_likelihood(y, quiet=quiet)

    def _predict_cov(self, y, xs):
        """(mu, cov) of george's ``GP.predict`` defaults (return_cov=True): cov = k(t, t) - V^T V with
        V = L^-1 k(X, t).  Not on approxposterior's path (it only ever asks for the mean or
        return_var=True: approx.py:178, utility.py:131,178,224) -- served for callers that use the
        george default: the two cross-kernel matrices by the HIP kernel of the row append
        (apgp_kernel_cross), the triangular solve against the resident factor and the (M, N) x (N, M)
        product as plain library calls (rocBLAS trsm / gemm through torch), all on the device.
        Memory: (2 N + M) M doubles."""
        torch, dev, lib = self._rt()
        mu, = self._sweep(y, xs, kind=None, want=("mu",))
        n, m = len(self._x), len(xs)
        if m == 0:
            return mu, np.empty((0, 0), dtype=np.float64)
        with self._on(torch, dev):
            st = self._stream(torch)
            ks = self._kernel_struct()
            if getattr(self, "_x_d", None) is None or self._x_d.shape[0] != n:
                self._x_d = torch.from_numpy(self._x).to(dev)
            t_d = torch.from_numpy(np.ascontiguousarray(xs, dtype=np.float64)).to(dev)
            kxt = torch.empty((m, n), dtype=torch.float64, device=dev)
            cov = torch.empty((m, m), dtype=torch.float64, device=dev)
            _lib.check(lib.apgp_kernel_cross(t_d.data_ptr(), m, self._x_d.data_ptr(), n, ctypes.byref(ks),
                                             kxt.data_ptr(), n, st), "apgp_kernel_cross")
            _lib.check(lib.apgp_kernel_cross(t_d.data_ptr(), m, t_d.data_ptr(), m, ctypes.byref(ks),
                                             cov.data_ptr(), m, st), "apgp_kernel_cross")
            v = torch.linalg.solve_triangular(torch.tril(self._L[:n, :n]), kxt.T, upper=False)   # (bytes above the diagonal are undefined)
            cov -= v.T @ v
            return mu, cov.cpu().numpy()
