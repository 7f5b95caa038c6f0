# -*- coding: utf-8 -*-
"""
:py:mod:`gp.py` - MI355X-backed drop-in for the ``george.GP`` members that
approxposterior touches
--------------------------------------------------------------------------------

The reference (dflemin3/approxposterior) has no plugin/FFI interface: its seam is
the duck-typed ``george.GP`` instance passed as ``gp=`` (approx.py:77,140-144)
and re-created at approx.py:712-715.  This module provides objects exposing
exactly the members the reference calls:

  ``kernels.ExpSquaredKernel(metric, ndim)``, ``float * kernel``   gpUtils.py:160-165
  ``GP(kernel, fit_mean, mean, white_noise, fit_white_noise)``     gpUtils.py:176-177
  ``gp.compute(x)`` / ``gp.recompute()``                           gpUtils.py:178,244,254
  ``set/get_parameter_vector``, ``get_parameter_names``, ``len``   gpUtils.py:74,227,243
  ``gp.log_likelihood(y, quiet=True)``                             gpUtils.py:78,247
  ``gp.grad_log_likelihood(y, quiet=True)``                        gpUtils.py:110
  ``gp.predict(y, t, return_var=True)`` / mean only                utility.py:131; approx.py:178
  ``gp.computed``, ``gp.kernel``, ``gp.mean``, ``gp.white_noise``  utility.py:130; approx.py:712-714

All arithmetic -- including the blocked Cholesky factorisation (csrc/potrf.hip)
-- runs in the hand-written HIP kernels of ``libapgp.so`` (include/apgp.h) on one
MI355X.  PyTorch is only the device allocator / stream provider.  There is no
CPU fallback: without a GPU or without the built extension every compute entry
point raises.

Beyond george's API the GP also offers the batched counterparts the reference
lacks: ``acquire`` (fused predict + utility + arg-min over a candidate matrix)
and array-valued ``predict``.
"""

import ctypes

import numpy as np
from numpy.linalg import LinAlgError

from . import _lib
from .george_extras import GeorgeExtras

__all__ = ["GP", "ExpSquaredKernel", "ConstantKernel", "Product", "ConstantModel",
           "kernels", "UTILITY_KINDS"]

UTILITY_KINDS = {"agp": _lib.UTIL_AGP, "bape": _lib.UTIL_BAPE, "jones": _lib.UTIL_JONES}


class _NullContext(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CONTEXT = _NullContext()

# Above this condition estimate ((max L_ii / min L_ii)^2) the explicit L^-1
# contraction is no longer trusted for the predictive variance (SURVEY.md
# section 7, "Conditioning vs. formulation") and the solve-based sweep
# (apgp_acquire_solve) is used instead.  ``GP.variance_mode`` = "solve" | "inverse"
# overrides the choice for one object (tests); no environment variable is read.
COND_SOLVE = 1.0e10
# from this size on a missing L^-1 is formed before the solves WHEN ALPHA IS WANTED (a sweep, a gradient or the
# sampler follows -- they need W anyway): apgp_trtri_pack + two matrix-vector products, 0.11 + 0.02 ms at N = 512
# against 0.08 + 0.09 ms for the two triangular solves, 0.82 + 0.05 ms against 2 x 0.72 ms at N = 4096
# (profiles/r03n_fit_timing.txt).  A bare log_likelihood on a new y keeps the single forward solve.
W_FIRST_MIN_N = 512


# ---------------------------------------------------------------------------
# Host-side parameter objects (no arithmetic lives here)
# ---------------------------------------------------------------------------

class ConstantModel(object):
    """Mirror of george.modeling.ConstantModel (``gp.mean``, ``gp.white_noise``)."""

    def __init__(self, value):
        self.value = float(value)

    def get_value(self, x):
        return self.value + np.zeros(len(x))

    def __len__(self):
        return 1

    def __repr__(self):
        return "ConstantModel(value=%r)" % self.value


def _as_model(obj, default):
    if obj is None:
        return ConstantModel(default)
    if isinstance(obj, ConstantModel) or (hasattr(obj, "value") and hasattr(obj, "get_value")):
        return obj
    return ConstantModel(float(obj))


class Kernel(object):
    is_kernel = True
    ndim = 1

    def __rmul__(self, b):
        # george: ``c * kernel`` -> Product(ConstantKernel(log(c/ndim)), kernel);
        # the constant kernel evaluates to ndim*exp(log_constant) = c
        # (gpUtils.py:165; pinned by test_InitGP.py:43 + test_GPUtil.py:50-62).
        if hasattr(b, "is_kernel"):
            return Product(b, self)
        return Product(ConstantKernel(log_constant=np.log(float(b) / self.ndim),
                                      ndim=self.ndim), self)

    __mul__ = __rmul__

    def __add__(self, other):
        # george: ``kernel + other`` -> Sum(kernel, other) (gpUtils.py:170, the optional
        # linear-regression term of defaultGP(order=...))
        if not hasattr(other, "is_kernel"):
            raise NotImplementedError("only kernel + kernel sums are on the MI355X hot path")
        return Sum(self, other)

    def __radd__(self, other):
        if not hasattr(other, "is_kernel"):
            raise NotImplementedError("only kernel + kernel sums are on the MI355X hot path")
        return Sum(other, self)

    def __len__(self):
        return len(self.get_parameter_vector())


class ConstantKernel(Kernel):
    def __init__(self, log_constant, ndim=1):
        self.log_constant = float(log_constant)
        self.ndim = int(ndim)
        self.dirty = True

    def get_parameter_names(self):
        return ("log_constant",)

    def get_parameter_vector(self):
        return np.array([self.log_constant])

    def set_parameter_vector(self, v):
        self.log_constant = float(v[0])
        self.dirty = True

    def __len__(self):
        return 1


class ExpSquaredKernel(Kernel):
    """k(x,x') = exp(-0.5 sum_d (x_d-x'_d)^2 / M_d); parameters are log M_d."""

    def __init__(self, metric, ndim=1):
        self.ndim = int(ndim)
        metric = np.atleast_1d(np.asarray(metric, dtype=np.float64))
        if metric.size == 1 and self.ndim > 1:
            metric = np.full(self.ndim, float(metric[0]))
        if metric.size != self.ndim:
            raise ValueError("Dimension mismatch")
        self.log_M = np.log(metric)
        self.dirty = True

    def get_parameter_names(self):
        return tuple("metric:log_M_%d_%d" % (d, d) for d in range(self.ndim))

    def get_parameter_vector(self):
        return np.array(self.log_M)

    def set_parameter_vector(self, v):
        self.log_M = np.array(v, dtype=np.float64)
        self.dirty = True

    def __len__(self):
        return len(self.log_M)


class LinearKernel(Kernel):
    """george.kernels.LinearKernel(log_gamma2, order, ndim) (gpUtils.py:170-173):
    k(x,x') = sum_d (x_d x'_d)^P / gamma^2 with the per-axis sum george uses for its
    non-stationary kernels; ``order`` P is a constant, ``log_gamma2`` the
    only parameter.  No reference test pins it ("parity unpinned"); P must be an integer
    >= 0 here."""

    def __init__(self, log_gamma2=None, order=None, bounds=None, ndim=1, axes=None):
        if log_gamma2 is None or order is None:
            raise ValueError("log_gamma2 and order are required")
        if axes is not None:
            raise NotImplementedError("axes subsets are not on the MI355X hot path")
        if int(order) != order or order < 0 or order > 16:
            raise NotImplementedError("LinearKernel order must be an integer in [0, 16] on the device path")
        self.log_gamma2 = float(log_gamma2)
        self.order = int(order)
        self.ndim = int(ndim)
        self.dirty = True

    def get_parameter_names(self):
        return ("log_gamma2",)

    def get_parameter_vector(self):
        return np.array([self.log_gamma2])

    def set_parameter_vector(self, v):
        self.log_gamma2 = float(v[0])
        self.dirty = True

    def __len__(self):
        return 1


class Product(Kernel):
    def __init__(self, k1, k2):
        self.k1 = k1
        self.k2 = k2
        self.ndim = k2.ndim

    @property
    def dirty(self):
        return self.k1.dirty or self.k2.dirty

    @dirty.setter
    def dirty(self, v):
        self.k1.dirty = v
        self.k2.dirty = v

    def get_parameter_names(self):
        return tuple(["k1:" + n for n in self.k1.get_parameter_names()] +
                     ["k2:" + n for n in self.k2.get_parameter_names()])

    def get_parameter_vector(self):
        return np.concatenate([self.k1.get_parameter_vector(),
                               self.k2.get_parameter_vector()])

    def set_parameter_vector(self, v):
        n1 = len(self.k1)
        self.k1.set_parameter_vector(v[:n1])
        self.k2.set_parameter_vector(v[n1:])

    def __len__(self):
        return len(self.k1) + len(self.k2)


class Sum(Product):
    """george.kernels.Sum: same parameter protocol as Product (k1:..., k2:...)."""


class _KernelsNamespace(object):
    """Stands in for ``george.kernels``."""
    ExpSquaredKernel = ExpSquaredKernel
    ConstantKernel = ConstantKernel
    LinearKernel = LinearKernel
    Product = Product
    Sum = Sum


kernels = _KernelsNamespace()


def _flatten_term(kernel):
    """One additive term: (constant factor, ExpSquaredKernel or None, LinearKernel or None)."""
    amp, se, lin = 1.0, None, None
    stack = [kernel]
    while stack:
        k = stack.pop()
        if isinstance(k, Sum):
            raise NotImplementedError("nested kernel sums are not on the MI355X hot path")
        if isinstance(k, Product):
            stack += [k.k1, k.k2]
        elif isinstance(k, ConstantKernel):
            amp *= k.ndim * np.exp(k.log_constant)
        elif isinstance(k, ExpSquaredKernel):
            if se is not None or lin is not None:
                raise NotImplementedError("product of two non-constant kernels")
            se = k
        elif isinstance(k, LinearKernel):
            if se is not None or lin is not None:
                raise NotImplementedError("product of two non-constant kernels")
            lin = k
        else:
            raise NotImplementedError("kernel type %r is not on the MI355X hot path" % type(k))
    return float(amp), se, lin


def _flatten_kernel(kernel, with_linear=False):
    """(amp, log_M) of an ExpSquared kernel, optionally times constants; with
    ``with_linear`` also (lin_coef, lin_order) of an added [constant *] LinearKernel
    (``kernel + c * LinearKernel``, gpUtils.py:170-173), (0.0, 0) if there is none."""
    # the two shapes gpUtils.defaultGP builds without a linear term (gpUtils.py:160-165), without the generic walk:
    # this runs once per objective evaluation of an optimiser loop
    tk = type(kernel)
    if tk is ExpSquaredKernel:
        return (1.0, kernel.log_M, 0.0, 0) if with_linear else (1.0, kernel.log_M)
    if tk is Product and type(kernel.k1) is ConstantKernel and type(kernel.k2) is ExpSquaredKernel:
        amp = float(kernel.k1.ndim * np.exp(kernel.k1.log_constant))
        return (amp, kernel.k2.log_M, 0.0, 0) if with_linear else (amp, kernel.k2.log_M)
    terms = [kernel.k1, kernel.k2] if isinstance(kernel, Sum) else [kernel]
    amp = log_M = None
    lin_coef, lin_order = 0.0, 0
    for term in terms:
        c, se, lin = _flatten_term(term)
        if se is not None:
            if amp is not None:
                raise NotImplementedError("sum of two ExpSquaredKernel terms")
            amp, log_M = c, np.asarray(se.log_M, dtype=np.float64)
        elif lin is not None:
            if lin_coef != 0.0:
                raise NotImplementedError("sum of two LinearKernel terms")
            lin_coef, lin_order = c * float(np.exp(-lin.log_gamma2)), lin.order
        else:
            raise NotImplementedError("a purely constant kernel term is not on the MI355X hot path")
    if amp is None:
        raise NotImplementedError("an ExpSquaredKernel term is required")
    if with_linear:
        return amp, log_M, float(lin_coef), int(lin_order)
    return amp, log_M


class _NllPlan(object):
    """The arguments of one ``apgp_nll_eval`` call, kept between the evaluations of an optimiser loop
    (``GP._factor_again``): device buffers (referenced, so their addresses stay valid), the kernel struct that is
    refilled in place, the host record, and what has to be unchanged for the plan to apply."""
    __slots__ = ("x", "ybytes", "n", "nlog2pi", "stream", "dev_index", "current_device", "raw_stream", "ks", "fn", "args",
                 "o", "K", "z", "keep")

    def __init__(self, gp, torch, dev, stream, ks, yv, x_d, y_d, K, z, scr, o, n):
        self.x = gp._x
        self.ybytes = yv.tobytes()
        self.n = n
        self.nlog2pi = n * np.log(2.0 * np.pi)
        self.stream = stream
        self.dev_index = dev.index
        self.current_device = torch.cuda.current_device
        self.raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self.ks = ks
        self.fn = gp._rt()[2].apgp_nll_eval
        self.o = o
        self.K, self.z = K, z
        self.keep = (x_d, y_d, scr)
        self.args = [x_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), 0.0, K.data_ptr(), z.data_ptr(),
                     scr[0].data_ptr(), scr[1].data_ptr(), o.ctypes.data, ctypes.c_void_p(stream)]


class _OnePlan(object):
    """The arguments of one ``apgp_predict1_host`` call (one candidate, mean + variance), kept for the next."""
    __slots__ = ("xs", "p1", "factor", "solve", "ybytes", "n", "ndim", "mean", "ld", "stream", "dev_index", "current_device",
                 "raw_stream", "ks", "ks_ref", "fn", "xs_ptr", "p1_ptr", "factor_ptr", "stream_arg")

    def __init__(self, gp, torch, dev, stream, ks, yv, n, solve):
        self.xs, self.p1 = gp._xs, gp._p1_work
        self.solve = bool(solve)
        self.factor = gp._L if solve else gp._work          # substitution against L, or the resident dense L^-1
        self.ld = int(gp._ld) if solve else (n + 63) // 64 * 64
        self.ybytes = yv.tobytes()
        self.n, self.ndim = n, int(ks.ndim)
        self.mean = float(gp.mean.value)
        self.stream, self.dev_index = stream, dev.index
        self.current_device = torch.cuda.current_device
        self.raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self.ks, self.ks_ref = ks, ctypes.byref(ks)
        self.fn = gp._rt()[2].apgp_predict1_host
        self.xs_ptr, self.p1_ptr, self.factor_ptr = gp._xs.data_ptr(), gp._p1_work.data_ptr(), self.factor.data_ptr()
        self.stream_arg = ctypes.c_void_p(stream)


class _MeanPlan(object):
    """The arguments of one small ``apgp_predict_mean_host`` call, kept for the next (``GP._predict_mean_again``)."""
    __slots__ = ("xs", "work", "ybytes", "n", "ndim", "mean", "max_m", "stream", "dev_index", "current_device", "raw_stream",
                 "ks", "ks_ref", "fn", "xs_ptr", "work_ptr", "stream_arg")

    def __init__(self, gp, torch, dev, stream, ks, yv, n):
        self.xs, self.work = gp._xs, gp._mean_work
        self.ybytes = yv.tobytes()
        self.n, self.ndim = n, int(ks.ndim)
        self.mean = float(gp.mean.value)
        self.max_m = min(4096, gp._mean_work.numel() // (self.ndim + 1))
        self.stream, self.dev_index = stream, dev.index
        self.current_device = torch.cuda.current_device
        self.raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self.ks, self.ks_ref = ks, ctypes.byref(ks)
        self.fn = gp._rt()[2].apgp_predict_mean_host
        self.xs_ptr, self.work_ptr = gp._xs.data_ptr(), gp._mean_work.data_ptr()
        self.stream_arg = ctypes.c_void_p(stream)


# ---------------------------------------------------------------------------
# GP
# ---------------------------------------------------------------------------

class GP(GeorgeExtras):
    """``george.GP``-shaped object whose arithmetic runs on one MI355X.  (The members of ``george.GP`` the reference
    never calls -- ``apply_inverse``, ``get_matrix``, ``predict``'s covariance form, the ``nll`` aliases -- live in
    :py:mod:`approxposterior_amd.george_extras`, off the hot path.)"""

    def __init__(self, kernel=None, fit_kernel=True, mean=None, fit_mean=None,
                 white_noise=None, fit_white_noise=None, solver=None, device=None,
                 **kwargs):
        if kernel is None:
            raise ValueError("a kernel is required")
        self.kernel = kernel
        self.mean = _as_model(mean, 0.0)
        self.white_noise = _as_model(white_noise, np.log(1.25e-12))
        self.fit_mean = bool(fit_mean)
        self.fit_white_noise = bool(fit_white_noise)
        self._device_arg = device
        self.variance_mode = None     # None: by condition estimate; "solve" / "inverse": forced
        self.extend_max_rows = None   # rows compute(x, previous=...) may append before it refactorises (None: by cost)
        self._computed = False
        self._x = None
        self._yerr2 = 0.0
        self._nllMemo = None          # gpUtils._nll: values already evaluated on this training set
        self._batch_cache = None      # nll_batch: work buffers and the device copy of y, kept between the rounds of a fit
        self._reset_device_state()

    # -- device plumbing -------------------------------------------------------
    def _reset_device_state(self):
        self._L = None            # (N,N) lower Cholesky factor (device); a view of _L_store["buf"] after appends
        self._L_store = None      # growable store shared along a chain of appended GPs (compute(previous=))
        self._ld = None           # leading dimension of _L in memory
        self._nll_owned = False   # _L / _z are private buffers of an _nll evaluation (reusable by the next)
        self._alpha_y = None      # host copy of the y alpha/z were computed for
        self._alpha_mean = None
        self._y_d = None
        self._ztz_host = None
        self._z = None
        self._alpha = None
        self._packed = None       # packed L^-1 tiles
        self._packed_solve = None  # packed tiles of the substitution form (apgp_pack_lsolve)
        self._work = None         # trtri work (dense L^-1 in first panel)
        self._xs = None           # packed training stream (depends on alpha)
        self._xs_key = None
        self._mean_work = getattr(self, "_mean_work", None)   # scratch survives refits
        self._p1_work = getattr(self, "_p1_work", None)
        self._nll_scratch = getattr(self, "_nll_scratch", None)
        self._nll_plan = None     # the arguments of the last _nll evaluation, ready for the next (_factor_again)
        self._mean_plan = None    # ... of the last small mean-only prediction (_predict_mean_again)
        self._one_plan = None     # ... of the last single-candidate prediction with variance (_predict_one_again)
        self.cond_estimate = None
        self.log_determinant = None

    def _rt(self):
        """(torch, device, lib) -- fails loudly without GPU / extension."""
        rt = getattr(self, "_rt_cache", None)
        if rt is not None:
            return rt
        import torch
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.ApgpError("no MI355X visible: approxposterior_amd has no CPU fallback")
        dev = self._device_arg
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        elif not isinstance(dev, torch.device):
            dev = torch.device(dev)
        self._rt_cache = (torch, dev, lib)
        return self._rt_cache

    @staticmethod
    def _on(torch, dev):
        """Context that makes ``dev`` current -- a no-op object when it already is (the context manager of
        torch.cuda.device costs ~4 us, on the per-evaluation path of gpUtils._nll)."""
        if torch.cuda.current_device() == dev.index:
            return _NULL_CONTEXT
        return torch.cuda.device(dev)

    @staticmethod
    def _stream(torch):
        # the raw-handle query is ~30x cheaper than torch.cuda.current_stream() and this
        # sits on the per-call path of the sampler's log-probability
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return ctypes.c_void_p(raw(torch.cuda.current_device()))
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _kernel_struct(self, ks=None):
        """The kernel's hyper-parameters as the C ABI's struct (``ks``: refilled in place)."""
        amp, log_M, lin_coef, lin_order = _flatten_kernel(self.kernel, with_linear=True)
        if ks is None:
            ks = _lib.KernelStruct()
        elif ks.ndim != len(log_M):
            raise ValueError("dimension mismatch")
        ks.ndim = len(log_M)
        ks.amp = amp
        ks.lin_coef = lin_coef
        ks.lin_order = lin_order
        ks.diag_add = float(self._yerr2) + float(np.exp(self.white_noise.value))
        ks.inv_metric[:len(log_M)] = np.exp(-np.asarray(log_M, dtype=np.float64)).tolist()   # (the rest stays 0)
        return ks

    # -- parameter-vector protocol (george) -------------
    def get_parameter_names(self):
        names = []
        if self.fit_mean:
            names.append("mean:value")
        if self.fit_white_noise:
            names.append("white_noise:value")
        names += ["kernel:" + n for n in self.kernel.get_parameter_names()]
        return tuple(names)

    def get_parameter_vector(self):
        v = []
        if self.fit_mean:
            v.append(self.mean.value)
        if self.fit_white_noise:
            v.append(self.white_noise.value)
        return np.concatenate([np.array(v, dtype=np.float64),
                               self.kernel.get_parameter_vector()])

    def set_parameter_vector(self, p):
        p = np.asarray(p, dtype=np.float64).ravel()
        if len(p) != len(self):
            raise ValueError("dimension mismatch")
        n = 0
        if self.fit_mean:
            self.mean.value = float(p[n]); n += 1
        if self.fit_white_noise:
            self.white_noise.value = float(p[n]); n += 1
        self.kernel.set_parameter_vector(p[n:])
        self.kernel.dirty = True   # marks dirty; no compute (george semantics)

    def __len__(self):
        return int(self.fit_mean) + int(self.fit_white_noise) + len(self.kernel)

    @property
    def computed(self):
        return self._computed and not self.kernel.dirty

    # -- helpers -----------------------------------------------------------------
    def parse_samples(self, t):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        if t.ndim == 1:
            t = t[:, None]
        if t.ndim != 2 or t.shape[1] != self.kernel.ndim:
            raise ValueError("Dimension mismatch")
        return np.ascontiguousarray(t)

    def _check_dimensions(self, y):
        y = np.atleast_1d(np.asarray(y, dtype=np.float64))
        if self._x is None or y.shape[0] != len(self._x):
            raise ValueError("Dimension mismatch")
        return np.ascontiguousarray(y)

    # -- compute / recompute: K1 gram + blocked Cholesky (+ fused forward solve) + K2 ------
    def compute(self, x, yerr=0.0, previous=None, **kwargs):
        """george GP.compute.  ``previous`` (extension): a GP factorised with the
        same hyper-parameters on a training set that is a PREFIX of ``x`` -- what
        ApproxPosterior.findNextPoint has at hand when it appends a design point
        (approx.py:693-717).  The factor is then extended row by row in O(N^2) per
        new point instead of refactorised in O(N^3)."""
        x_in = x
        x = self.parse_samples(x)
        if x.shape[1] > _lib.MAX_DIM:
            raise ValueError("approxposterior_amd supports at most %d input dimensions (APGP_MAX_DIM in include/apgp.h: the "
                             "kernels keep a point's coordinates in registers); got %d" % (_lib.MAX_DIM, x.shape[1]))
        if x is x_in or (isinstance(x_in, np.ndarray) and np.shares_memory(x, x_in)):
            x = x.copy()              # the object owns its training set: a caller who edits x in place afterwards is not seen
        same_x = self._x is not None and self._x.shape == x.shape and np.array_equal(self._x, x)
        self._x = x
        self._nllMemo = None          # gpUtils._nll's table of evaluated points belongs to the old training set
        self._yerr2 = float(yerr) ** 2
        if previous is not None and self._try_extend(previous):
            return
        self._factor(None, upload_x=not same_x)

    def _trust_inverse(self):
        """True when the explicit L^-1 (sweep contraction, alpha by matrix-vector products, row
        appends) may stand in for triangular solves: forced by ``variance_mode``, otherwise by the
        condition estimate against COND_SOLVE."""
        mode = (self.variance_mode or "").lower()
        if mode == "inverse":
            return True
        if mode == "solve":
            return False
        return self.cond_estimate is not None and self.cond_estimate <= COND_SOLVE

    def _factor_key(self):
        """What the factor depends on: the bytes of the C ABI's kernel struct (amplitude, inverse metric, linear term,
        white noise + yerr^2) -- the mean is not part of it."""
        return bytes(self._kernel_struct())

    def _try_extend(self, prev):
        """Extend ``prev``'s Cholesky factor by the rows of self._x it does not cover:
        l = L^-1 k(x_new, X_old), d = sqrt(k(x_new,x_new) + diag_add - l.l)."""
        if getattr(prev, "_L", None) is None or getattr(prev, "_x", None) is None:
            return False
        if getattr(prev, "_factored_key", None) != self._factor_key():
            return False
        n0, n1 = len(prev._x), len(self._x)
        # one appended row costs a triangular solve (0.40 us per row of N) and the whole Cholesky
        # 0.47 us per row of N + 30 us: beyond one row (two below N = 512) the
        # refactorisation is the faster way.  ``extend_max_rows`` overrides the rule (up to 64).
        limit = self.extend_max_rows if self.extend_max_rows is not None else (1 if n1 >= 512 else 2)
        if not (0 < n0 < n1 and n1 - n0 <= min(64, int(limit)) and prev._x.shape[1] == self._x.shape[1]
                and np.array_equal(prev._x, self._x[:n0])):
            return False
        torch, dev, lib = self._rt()
        ks = self._kernel_struct()
        w_prev = None
        if getattr(prev, "_work", None) is not None and prev._trust_inverse():
            w_prev = prev._work
        prev_L, prev_store = prev._L, getattr(prev, "_L_store", None)
        self._reset_device_state()
        self._computed = False
        with torch.cuda.device(dev):
            st = self._stream(torch)
            self._x_d = torch.from_numpy(self._x).to(dev)
            # The factor lives in a growable store shared along the chain of GP objects findNextPoint
            # creates (approx.py:712-717: a new GP per appended point): rows are appended IN PLACE while
            # the store has room and nobody else has appended to it (``used`` == the previous size: a
            # second child of the same parent gets its own copy); otherwise a 1.5x larger zeroed store
            # is allocated and the old factor copied once -- not an n x n allocation + copy per point.
            if prev_store is not None and prev_store["used"] == n0 and prev_store["buf"].shape[0] >= n1:
                store = prev_store
            else:
                cap = ((max(n1 + 63, int(1.5 * n1)) + 63) // 64) * 64
                store = {"buf": torch.zeros((cap, cap), dtype=torch.float64, device=dev), "used": n0}
                store["buf"][:n0, :n0].copy_(prev_L[:n0, :n0])
            buf = store["buf"]
            ld = buf.shape[1]
            L = buf[:n1, :n1]
            row = torch.empty(n1, dtype=torch.float64, device=dev)
            ss = torch.empty(1, dtype=torch.float64, device=dev)
            info = torch.zeros(1, dtype=torch.int32, device=dev)
            for j in range(n0, n1):
                # row j: l = L^-1 k(x_j, X[:j]), pivot sqrt(k(x_j,x_j) + diag_add - l.l) -- all enqueued,
                # no host round trip per row (a failed pivot is reported through `info`)
                _lib.check(lib.apgp_kernel_cross(self._x_d[j:].data_ptr(), 1, self._x_d.data_ptr(), j,
                                                 ctypes.byref(ks), row.data_ptr(), n1, st), "apgp_kernel_cross")
                if j == n0 and w_prev is not None:
                    # the previous fit's dense L^-1 is resident (a sweep ran on it): l = W k is an
                    # HBM-rate matrix-vector product (18 us at N = 4096; the solve: 0.7 ms)
                    _lib.check(lib.apgp_winv_apply(w_prev.data_ptr(), (n0 + 63) // 64 * 64, n0, row.data_ptr(), 0.0,
                                                   0, L[j].data_ptr(), ss.data_ptr(), None, st),
                               "apgp_winv_apply(append)")
                else:
                    _lib.check(lib.apgp_trsv(L.data_ptr(), j, ld, row.data_ptr(), 0.0, 0, L[j].data_ptr(),
                                             ss.data_ptr(), st), "apgp_trsv(append)")
                kxx = ks.amp            # k(x_new, x_new): amplitude + the linear term's sum_d (x_d^2)^P
                if ks.lin_coef != 0.0:
                    kxx += ks.lin_coef * float(np.sum((self._x[j] * self._x[j]) ** ks.lin_order))
                _lib.check(lib.apgp_append_diag(L[j, j:].data_ptr(), ss.data_ptr(), kxx + ks.diag_add,
                                                info.data_ptr(), j + 1, st), "apgp_append_diag")
            store["used"] = n1
            out5 = torch.empty(5, dtype=torch.float64, device=dev)
            _lib.check(lib.apgp_fit_summary(L.data_ptr(), n1, ld, None, info.data_ptr(), out5.data_ptr(), st),
                       "apgp_fit_summary")
            o = out5.cpu().numpy()          # the only synchronisation of the extension
        if int(o[4]) != 0 or not np.isfinite(o[0]):
            store["used"] = -1            # (the failed rows stay in the store: nobody may append to it again)
            self._reset_device_state()
            raise LinAlgError("%d-th leading minor of the array is not positive definite" % int(o[4]))
        self._L = L
        self._L_store = store
        self._ld = ld
        self.log_determinant = float(o[0])
        self.cond_estimate = float((o[2] / o[1]) ** 2)
        self._const = -0.5 * (n1 * np.log(2.0 * np.pi) + self.log_determinant)
        self._computed = True
        self.kernel.dirty = False
        self._factored_key = self._factor_key()
        return True

    def _factor(self, y, upload_x=False):
        """Gram + Cholesky (+ z = L^-1 (y - mean) carried through the factorisation)
        + log-determinant / diagonal range / z.z / info, fetched with ONE 40-byte
        device-to-host copy.  This is one gpUtils._nll evaluation."""
        if y is not None and not upload_x and self._nll_plan is not None and self._factor_again(y):
            return
        torch, dev, lib = self._rt()
        x = self._x
        n = len(x)
        yv = None if y is None else self._check_dimensions(y)
        keep_x = None if upload_x else getattr(self, "_x_d", None)
        keep_y = self._y_d if (yv is not None and self._alpha_y is not None
                               and np.array_equal(self._alpha_y, yv)) else None
        keep_nll = (self._L, self._z) if (self._L is not None and self._z is not None and self._L_store is None
                                          and self._nll_owned) else None
        keep_stream = getattr(self, "_nll_stream", None)
        self._reset_device_state()
        self._computed = False
        ks = self._kernel_struct()
        with self._on(torch, dev):
            st = self._stream(torch)
            self._x_d = keep_x if keep_x is not None else torch.from_numpy(x).to(dev)
            # zeroed: the library reads and writes the lower triangle only (Gram, factor in place -- LAPACK's
            # dpotrf contract), so what is kept as the factor is a clean lower-triangular L.
            # (n <= 64 with y: the fused kernel writes the whole n x n itself.)
            # An optimiser's evaluations (set_parameter_vector + log_likelihood, over and over) refactorise in
            # the SAME buffers: the previous factor is dead once its hyper-parameters are, and nothing else
            # references an exactly-sized private factor (appended chains share a store: never reused here).
            # ... on the SAME stream only: the in-place refactorisation (memset + Gram) is ordered behind the previous
            # evaluation's readers by stream order, not by the caching allocator -- a caller that switched the current
            # stream between two evaluations gets fresh buffers
            reuse = keep_nll if (keep_nll is not None and yv is not None and keep_nll[0].shape[0] == n
                                 and keep_stream == st.value) else None
            if reuse is not None:
                # (no memset: the tiles on and below the diagonal are rewritten in full by the Gram kernel, the strict
                # upper tiles are still the zeros of the allocation -- nothing in the library writes them; the memset
                # was a fourth dependent launch per evaluation, 14 us on average over N = 512 .. 4096)
                K, z = reuse
            else:
                K = (torch.empty if (yv is not None and n <= 64) else torch.zeros)((n, n), dtype=torch.float64, device=dev)
                z = None
            # the whole evaluation as ONE library call and one synchronisation: Gram + Cholesky on the persistent /
            # hybrid plan (+ z = L^-1 (y - mean) riding along when y is given: a gpUtils._nll evaluation; without y:
            # compute() / recompute(), the same plan -- round 4 still sent these through a launch per 64-column step)
            y_d = None
            if yv is not None:
                y_d = keep_y if keep_y is not None else torch.from_numpy(yv).to(dev)
                if z is None:
                    z = torch.empty(n, dtype=torch.float64, device=dev)
            scr = self._nll_scratch
            if scr is None:
                scr = self._nll_scratch = (torch.empty(1, dtype=torch.int32, device=dev),
                                           torch.empty(5, dtype=torch.float64, device=dev))
            o = np.empty(5, dtype=np.float64)
            _lib.check(lib.apgp_nll_eval(self._x_d.data_ptr(), n, ctypes.byref(ks),
                                         y_d.data_ptr() if y_d is not None else None,
                                         float(self.mean.value), K.data_ptr(),
                                         z.data_ptr() if yv is not None else None,
                                         scr[0].data_ptr(), scr[1].data_ptr(), o.ctypes.data, st),
                       "apgp_nll_eval")
            if yv is None:
                z = None
            L = K
        if int(o[4]) != 0:
            # same failure mode as scipy.linalg.cholesky inside george
            raise LinAlgError("%d-th leading minor of the array is not positive definite" % int(o[4]))
        if not np.isfinite(o[0]):
            raise LinAlgError("non-finite log-determinant")
        self._L = L
        self._ld = n
        self.log_determinant = float(o[0])
        self.cond_estimate = float((o[2] / o[1]) ** 2)
        self._const = -0.5 * (n * np.log(2.0 * np.pi) + self.log_determinant)
        self._computed = True
        self.kernel.dirty = False
        self._factored_key = bytes(ks)
        if z is not None:
            self._z = z
            self._ztz_host = float(o[3])
            self._y_d = y_d
            self._alpha_y = np.array(yv, copy=True)
            self._alpha_mean = self.mean.value
            self._nll_owned = True    # (K, z) came from an _nll evaluation: the next one may refactorise in place
            self._nll_stream = st.value
            # everything the NEXT evaluation on this training set, y and stream needs, ready to go (_factor_again)
            self._nll_plan = _NllPlan(self, torch, dev, st.value or 0, ks, yv, self._x_d, y_d, K, z, scr, o, n)

    def _factor_again(self, y):
        """One more gpUtils._nll evaluation (gpUtils.py:46-80) in the buffers of the last one: what ``_factor`` does when
        it finds them reusable, minus everything that cannot have changed -- the optimiser loop's path (SciPy asks for
        ~6e5 evaluations in BASELINE config 5; the generic path's 17 us of Python between two device evaluations were 7 %
        of each).  False: something differs (training set, y, stream, device, buffers) -- the caller takes the generic path."""
        plan = self._nll_plan
        if (plan.x is not self._x or plan.K is not self._L or plan.z is not self._z or self._L_store is not None
                or not self._nll_owned or type(y) is not np.ndarray or y.dtype != np.float64):
            return False
        cur = plan.current_device()
        if cur != plan.dev_index or plan.raw_stream is None or plan.raw_stream(cur) != plan.stream:
            return False
        if y.size != plan.n or not y.flags.c_contiguous or y.tobytes() != plan.ybytes:
            return False
        ks = self._kernel_struct(plan.ks)
        # (as _reset_device_state: whatever was derived from the previous factor is stale)
        self._alpha = self._packed = self._packed_solve = self._work = self._xs = self._xs_key = None
        self._computed = False
        args = plan.args
        args[4] = float(self.mean.value)
        rc = plan.fn(*args)
        if rc != 0:
            self._reset_device_state()
            _lib.check(rc, "apgp_nll_eval")
        o = plan.o
        if o[4] != 0.0 or not np.isfinite(o[0]):
            self._reset_device_state()
            if o[4] != 0.0:
                raise LinAlgError("%d-th leading minor of the array is not positive definite" % int(o[4]))
            raise LinAlgError("non-finite log-determinant")
        logdet = float(o[0])
        self.log_determinant = logdet
        self.cond_estimate = float((o[2] / o[1]) ** 2)
        self._const = -0.5 * (plan.nlog2pi + logdet)
        self._ztz_host = float(o[3])
        self._alpha_mean = self.mean.value
        self._computed = True
        self.kernel.dirty = False
        self._factored_key = bytes(ks)
        return True

    def recompute(self, quiet=False, **kwargs):
        if self.kernel.dirty or not self._computed:
            if self._x is None:
                raise RuntimeError("You need to compute the model first")
            try:
                self._factor(None)
            except (ValueError, LinAlgError):
                if quiet:
                    return False
                raise
        return True

    # -- K3: z = L^-1 (y - mean), alpha = L^-T z -----------------------------------
    def _solve(self, y, need_alpha):
        """Ensures z (and alpha) for this y; returns z.z as a float.  With the dense
        W = L^-1 resident (every sweep set-up) and a trusted condition estimate, both are
        HBM-rate matrix-vector products (``apgp_winv_apply``); otherwise the triangular
        solves (``apgp_trsv``), which also serve ill-conditioned factors."""
        torch, dev, lib = self._rt()
        y = self._check_dimensions(y)
        n = len(y)
        same = (self._alpha_y is not None and self._alpha_mean == self.mean.value
                and np.array_equal(self._alpha_y, y))
        trust_w = self._trust_inverse()
        need_solve = (not same or self._z is None) or (need_alpha and self._alpha is None)
        if trust_w and need_solve and need_alpha and self._work is None and n >= W_FIRST_MIN_N:
            # alpha wanted (a sweep, a gradient or the sampler follows) and no inverse yet: L^-1
            # (0.8 ms at N = 4096) + two matrix-vector products beat the two triangular solves
            # (0.7 ms each), and the sweep / gradient needs W anyway.  A bare log_likelihood on a new
            # y (need_alpha False) keeps the ONE forward solve: it is cheaper than the O(N^3) inverse.
            self._ensure_linv()
        via_w = self._work is not None and trust_w
        np64 = (n + 63) // 64 * 64
        with torch.cuda.device(dev):
            st = self._stream(torch)
            if not same or self._z is None:
                self._y_d = torch.from_numpy(y).to(dev)
                self._z = torch.empty(n, dtype=torch.float64, device=dev)
                ztz = torch.empty(1, dtype=torch.float64, device=dev)
                if via_w:
                    _lib.check(lib.apgp_winv_apply(self._work.data_ptr(), np64, n, self._y_d.data_ptr(),
                                                   float(self.mean.value), 0, self._z.data_ptr(),
                                                   ztz.data_ptr(), None, st), "apgp_winv_apply(forward)")
                else:
                    _lib.check(lib.apgp_trsv(self._L.data_ptr(), n, self._ld, self._y_d.data_ptr(),
                                             float(self.mean.value), 0, self._z.data_ptr(),
                                             ztz.data_ptr(), st), "apgp_trsv(forward)")
                self._ztz_host = float(ztz.item())
                if not via_w and self._ztz_host != self._ztz_host:
                    # NaN: the persistent solve could not get its workgroups resident (apgp.h: it writes NaN instead of
                    # hanging) -- or the factor is not finite, in which case the re-run says so too.  Once more, a launch per
                    # 256 rows -- chosen for THIS call only (apgp_trsv_ex): other threads' and streams' solves keep their path.
                    _lib.check(lib.apgp_trsv_ex(self._L.data_ptr(), n, self._ld, self._y_d.data_ptr(),
                                                float(self.mean.value), 0, self._z.data_ptr(),
                                                ztz.data_ptr(), 1, st), "apgp_trsv(forward, multi-launch)")
                    self._ztz_host = float(ztz.item())
                self._alpha = None
                self._xs = None
                self._alpha_y = np.array(y, copy=True)
                self._alpha_mean = self.mean.value
            if need_alpha and self._alpha is None:
                self._alpha = torch.empty(n, dtype=torch.float64, device=dev)
                if via_w:
                    wk = torch.empty(int(lib.apgp_winv_apply_work_len(n)), dtype=torch.float64, device=dev)
                    _lib.check(lib.apgp_winv_apply(self._work.data_ptr(), np64, n, self._z.data_ptr(), 0.0, 1,
                                                   self._alpha.data_ptr(), None, wk.data_ptr(), st),
                               "apgp_winv_apply(backward)")
                else:
                    asq = torch.empty(1, dtype=torch.float64, device=dev)
                    _lib.check(lib.apgp_trsv(self._L.data_ptr(), n, self._ld, self._z.data_ptr(), 0.0, 1,
                                             self._alpha.data_ptr(), asq.data_ptr(), st), "apgp_trsv(backward)")
                    if float(asq.item()) != float(asq.item()):      # (NaN: see the forward solve)
                        _lib.check(lib.apgp_trsv_ex(self._L.data_ptr(), n, self._ld, self._z.data_ptr(), 0.0, 1,
                                                    self._alpha.data_ptr(), None, 1, st), "apgp_trsv(backward, multi-launch)")
                self._xs = None
        return self._ztz_host

    def log_likelihood(self, y, quiet=False):
        """george GP.log_likelihood (gpUtils.py:78,247): never raises when quiet.
        When the model is dirty (the gpUtils._nll pattern: set_parameter_vector then
        log_likelihood) the refactorisation carries y along, so the whole evaluation
        is gram + potrf + one reduction + one 40-byte copy."""
        try:
            if self.kernel.dirty or not self._computed:
                if self._x is None:
                    raise RuntimeError("You need to compute the model first")
                self._factor(y)
                ztz = self._ztz_host          # (z = L^-1 (y - mean) rode along with the factorisation)
            else:
                ztz = self._solve(y, need_alpha=False)
            ll = self._const - 0.5 * ztz
        except (ValueError, LinAlgError):
            if quiet:
                return -np.inf
            raise
        return ll if np.isfinite(ll) else -np.inf

    # Powell look-ahead (gpUtils._powellAhead): how many points beyond the one asked for are worth evaluating in the same
    # device call -- where a small batch costs little more than one evaluation (apgp_nll_eval_batch: one launch with a
    # workgroup per matrix up to n = 128; the persistent factorisations side by side in one launch above that, each on
    # 1 / batch of the CUs -- tools/nll_side_batch.py, profiles/r06c_*: 6 matrices 1.2 / 1.2 / 1.3 x one evaluation at
    # n = 512 / 832 / 1152, 4 matrices 1.4 x at 1664, 2 matrices 1.13 x at 2048); widths measured inside SciPy's Powell
    # (tools/nll_powell_rate.py --width 3,4,5; profiles/r06p_*): 5 is best up to n = 1152, 3 at 1600; 0 = off.
    lookahead = None              # None: by size; 0: off; k: that many points

    def lookahead_width(self):
        if self.lookahead is not None:
            return int(self.lookahead)
        n = 0 if self._x is None else len(self._x)
        if n <= 0:
            return 0
        return 5 if n <= 1216 else (3 if n <= 1728 else (1 if n <= 2112 else 0))

    def _fast_structs(self, P, out):
        """The C ABI's kernel structs + means of the hyper-vectors ``P`` (B x len(self)) written straight into ``out``
        (B x 36 doubles = B ``apgp_kernel_t``), for the two kernel shapes ``defaultGP`` builds without a linear term
        (gpUtils.py:160-165) -- every field by the expression ``_kernel_struct`` uses after ``set_parameter_vector``, so
        the same bits, without touching the object's state.  None for any other kernel."""
        k = self.kernel
        tk = type(k)
        if tk is ExpSquaredKernel:
            amp_dim, d = 0, k.ndim
        elif tk is Product and type(k.k1) is ConstantKernel and type(k.k2) is ExpSquaredKernel:
            amp_dim, d = k.k1.ndim, k.k2.ndim
        else:
            return None
        at = 0
        if self.fit_mean:
            means = P[:, 0].copy(); at += 1
        else:
            means = np.full(len(P), float(self.mean.value))
        ints = out.view(np.int32)
        out[:] = 0.0
        ints[:, 0] = d
        c = at + int(self.fit_white_noise)
        # (np.exp is an element-wise ufunc: the value of an element does not depend on the array it sits in, so these are
        # the bits of _kernel_struct's scalar / length-d calls -- asserted by the look-ahead tests, which compare an
        # optimiser's whole trajectory through this path with the one through single evaluations)
        if self.fit_white_noise:
            out[:, 2] = float(self._yerr2) + np.exp(np.ascontiguousarray(P[:, at]))      # (contiguous: the ufunc's SIMD loop)
        else:
            out[:, 2] = float(self._yerr2) + float(np.exp(self.white_noise.value))
        if amp_dim:
            out[:, 1] = amp_dim * np.exp(np.ascontiguousarray(P[:, c]))
            c += 1
        else:
            out[:, 1] = 1.0
        out[:, 3:3 + d] = np.exp(-P[:, c:c + d])
        return means

    def nll_batch(self, P, y):
        """Negative marginal log-likelihood at each hyper-parameter vector of ``P`` (B x P),
        evaluated by ONE batched Gram + Cholesky + solve call (``apgp_nll_eval_batch``;
        SURVEY.md section 8(f) rank 3).  Entry b is what ``gpUtils._nll(P[b], gp, y, None)``
        returns -- bit-identical, every matrix takes the code path of the single call --
        with ``+inf`` for a non-positive-definite Gram matrix.  The GP's own parameter
        vector is restored; its factorisation is marked stale."""
        torch, dev, lib = self._rt()
        if self._x is None:
            raise RuntimeError("You need to compute the model first")
        P = np.atleast_2d(np.asarray(P, dtype=np.float64))
        B, n = len(P), len(self._x)
        if P.shape[1] != len(self):
            raise ValueError("dimension mismatch")
        cache = self._batch_cache
        fast = None
        if (B <= 8 and cache is not None and cache["x"] is self._x and type(y) is np.ndarray and y.dtype == np.float64
                and y.size == n and y.flags.c_contiguous and cache["ybytes"] == y.tobytes()
                and torch.cuda.current_device() == dev.index and cache["stream"] == (self._stream(torch).value or 0)):
            # a small batch on the training set, y and stream of the previous one (the look-ahead of a Powell line search,
            # the rounds of a lock-step fit): buffers, argument list and struct array are ready
            fast = cache["plans"].get(B)
        if fast is not None:
            karr, o, args, fn = fast
            means = self._fast_structs(P, karr)
            if means is not None:
                marr = fast_means = cache["means"][:B]
                marr[:] = means
                self.kernel.dirty = True          # (the buffers the object's own factor may share are not touched; as below)
                self._computed = False
                _lib.check(fn(*args), "apgp_nll_eval_batch")
                with np.errstate(all="ignore"):
                    ll = (-0.5 * (n * np.log(2.0 * np.pi) + o[:, 0])) - 0.5 * o[:, 3]
                bad = (o[:, 4] != 0.0) | ~np.isfinite(o[:, 0]) | ~np.isfinite(ll)
                return np.where(bad, np.inf, -ll)
        yv = self._check_dimensions(y)
        out = np.full(B, np.inf)
        saved = self.get_parameter_vector()
        structs, means, live = [], [], []
        try:
            for b in range(B):
                try:
                    self.set_parameter_vector(P[b])
                    ks = self._kernel_struct()
                except (LinAlgError, ValueError, OverflowError):
                    continue
                structs.append(ks)
                means.append(float(self.mean.value))
                live.append(b)
        finally:
            self.set_parameter_vector(saved)
        if not live:
            return out
        # chunks bounded by 2 GiB of Gram-matrix work space
        per = max(1, int((2 << 30) // (8 * n * n)))
        with self._on(torch, dev):
            st = self._stream(torch)
            if getattr(self, "_x_d", None) is None:
                self._x_d = torch.from_numpy(self._x).to(dev)
            # the rounds of a lock-step fit (gpUtils._minimizeLockStep) come back with the same y, batch size and stream a few
            # hundred times: the device copy of y and the work buffers are kept (the call is synchronous: nothing of the
            # previous round is in flight)
            ybytes = yv.tobytes()
            if (cache is None or cache["x_d"] is not self._x_d or cache["stream"] != (st.value or 0)
                    or cache["ybytes"] != ybytes or cache["x"] is not self._x):
                cache = self._batch_cache = {"x_d": self._x_d, "x": self._x, "stream": st.value or 0, "ybytes": ybytes,
                                             "y_d": torch.from_numpy(yv).to(dev), "bufs": {}, "plans": {},
                                             "means": np.empty(8, dtype=np.float64)}
            y_d = cache["y_d"]
            for c0 in range(0, len(live), per):
                idx = live[c0:c0 + per]
                nb = len(idx)
                karr = (_lib.KernelStruct * nb)(*structs[c0:c0 + nb])
                marr = np.array(means[c0:c0 + nb], dtype=np.float64)
                bufs = cache["bufs"].get(nb)
                if bufs is None:
                    bufs = (torch.empty((nb, n, n), dtype=torch.float64, device=dev),
                            torch.empty((nb, n), dtype=torch.float64, device=dev),
                            torch.empty(nb, dtype=torch.int32, device=dev),
                            torch.empty((nb, 5), dtype=torch.float64, device=dev))
                    if 8 * nb * n * n <= (256 << 20):     # (larger work spaces are not hoarded; 6 matrices of N = 2300 fit)
                        cache["bufs"][nb] = bufs
                K, z, info, o_d = bufs
                o = np.empty((nb, 5), dtype=np.float64)
                _lib.check(lib.apgp_nll_eval_batch(self._x_d.data_ptr(), n, nb, ctypes.addressof(karr),
                                                   y_d.data_ptr(), marr.ctypes.data, K.data_ptr(), z.data_ptr(),
                                                   info.data_ptr(), o_d.data_ptr(), o.ctypes.data, st),
                           "apgp_nll_eval_batch")
                if nb == B and B <= 8 and nb in cache["bufs"] and ctypes.sizeof(_lib.KernelStruct) == 36 * 8:
                    # the next batch of this size on this training set, y and stream: everything but the hyper-vectors ready
                    ka = np.zeros((B, 36), dtype=np.float64)
                    oo = np.empty((B, 5), dtype=np.float64)
                    cache["plans"][B] = (ka, oo, [self._x_d.data_ptr(), n, B, ka.ctypes.data, y_d.data_ptr(),
                                                  cache["means"].ctypes.data, K.data_ptr(), z.data_ptr(), info.data_ptr(),
                                                  o_d.data_ptr(), oo.ctypes.data, ctypes.c_void_p(st.value)],
                                         lib.apgp_nll_eval_batch)
                for j, b in enumerate(idx):
                    if int(o[j, 4]) != 0 or not np.isfinite(o[j, 0]):
                        continue
                    const = -0.5 * (n * np.log(2.0 * np.pi) + float(o[j, 0]))
                    ll = const - 0.5 * float(o[j, 3])
                    if np.isfinite(ll):
                        out[b] = -ll
        return out

    # -- packed factor / training stream for the sweep -------------------------------
    def _ensure_linv(self):
        torch, dev, lib = self._rt()
        if self._packed is not None:
            return
        n = len(self._x)
        with torch.cuda.device(dev):
            st = self._stream(torch)
            self._work = torch.empty(lib.apgp_trtri_work_len(n), dtype=torch.float64, device=dev)
            self._packed = torch.empty(lib.apgp_packed_linv_len(n), dtype=torch.float64, device=dev)
            _lib.check(lib.apgp_trtri_pack(self._L.data_ptr(), n, self._ld, self._work.data_ptr(),
                                           self._packed.data_ptr(), None, st), "apgp_trtri_pack")

    def _ensure_lsolve(self):
        """Tiles of the substitution form of the sweep: one O(N^2) pass over the factor."""
        torch, dev, lib = self._rt()
        if self._packed_solve is not None:
            return
        n = len(self._x)
        with torch.cuda.device(dev):
            st = self._stream(torch)
            self._packed_solve = torch.empty(lib.apgp_packed_lsolve_len(n), dtype=torch.float64, device=dev)
            _lib.check(lib.apgp_pack_lsolve(self._L.data_ptr(), n, self._ld, self._packed_solve.data_ptr(), st),
                       "apgp_pack_lsolve")

    def _ensure_xs(self, y):
        torch, dev, lib = self._rt()
        self._solve(y, need_alpha=True)
        if self._xs is not None:
            return
        n = len(self._x)
        ks = self._kernel_struct()
        with torch.cuda.device(dev):
            st = self._stream(torch)
            self._xs = torch.empty(lib.apgp_packed_train_len(n, ks.ndim), dtype=torch.float64,
                                   device=dev)
            _lib.check(lib.apgp_pack_train(self._x_d.data_ptr(), self._alpha.data_ptr(), n,
                                           ctypes.byref(ks), self._xs.data_ptr(), st),
                       "apgp_pack_train")

    # -- predict (george GP.predict) ----------------------------
    def predict(self, y, t, return_cov=True, return_var=False, cache=True, **kwargs):
        if not return_var and not return_cov and self._mean_plan is not None:
            mu = self._predict_mean_again(y, t)
            if mu is not None:
                return mu
        if return_var and self._one_plan is not None:
            res = self._predict_one_again(y, t)
            if res is not None:
                return res
        self.recompute()
        xs = self.parse_samples(t)
        if return_cov and not return_var:
            return self._predict_cov(y, xs)
        if not return_var:
            mu, = self._sweep(y, xs, kind=None, want=("mu",))
            return mu
        mu, var = self._sweep(y, xs, kind=None, want=("mu", "var"))
        return mu, var

    def _predict_one_again(self, y, t):
        """Mean and variance at ONE more point for the model, y and stream of the previous such call (the reference's scalar
        utilities under Nelder-Mead, utility.py:131,178,224) -- the previous ``apgp_predict1_host`` call's arguments with a new
        point.  None: something differs, the caller takes the generic path."""
        plan = self._one_plan
        if (not self._computed or self.kernel.dirty or plan.xs is not self._xs or plan.p1 is not self._p1_work
                or plan.factor is not (self._L if plan.solve else self._work) or plan.mean != self.mean.value
                or plan.solve == self._trust_inverse()          # (variance_mode / the conditioning gate picked the other form)
                or type(y) is not np.ndarray or y.dtype != np.float64 or type(t) is not np.ndarray or t.dtype != np.float64
                or t.shape != (1, plan.ndim) or not t.flags.c_contiguous
                or y.size != plan.n or not y.flags.c_contiguous or y.tobytes() != plan.ybytes):
            return None
        cur = plan.current_device()
        if cur != plan.dev_index or plan.raw_stream is None or plan.raw_stream(cur) != plan.stream:
            return None
        o2 = np.empty(2, dtype=np.float64)
        if plan.solve:
            rc = plan.fn(t.ctypes.data, plan.xs_ptr, plan.n, plan.ks_ref, plan.mean, None, 0, plan.factor_ptr, plan.ld,
                         plan.p1_ptr, o2.ctypes.data, plan.stream_arg)
        else:
            rc = plan.fn(t.ctypes.data, plan.xs_ptr, plan.n, plan.ks_ref, plan.mean, plan.factor_ptr, plan.ld, None, 0,
                         plan.p1_ptr, o2.ctypes.data, plan.stream_arg)
        _lib.check(rc, "apgp_predict1_host")
        return np.array([o2[0]]), np.array([o2[1]])

    def _predict_mean_again(self, y, t):
        """The mean at a few more points for the model, y and stream of the previous such call (the walker ensembles of
        ``ApproxPosterior._gpllBatch``, approx.py:148-189 batched: 4e4 calls per chain in the README example) -- the previous
        call's arguments with new points, none of the generic path's checks that cannot have changed.  None: something
        differs, the caller takes the generic path."""
        plan = self._mean_plan
        if (not self._computed or self.kernel.dirty or plan.xs is not self._xs or plan.work is not self._mean_work
                or plan.mean != self.mean.value or type(y) is not np.ndarray or y.dtype != np.float64
                or type(t) is not np.ndarray or t.dtype != np.float64 or t.ndim != 2 or t.shape[1] != plan.ndim
                or not t.flags.c_contiguous):
            return None
        m = t.shape[0]
        if not 0 < m <= plan.max_m or y.size != plan.n or not y.flags.c_contiguous or y.tobytes() != plan.ybytes:
            return None
        cur = plan.current_device()
        if cur != plan.dev_index or plan.raw_stream is None or plan.raw_stream(cur) != plan.stream:
            return None
        mu_h = np.empty(m, dtype=np.float64)
        _lib.check(plan.fn(t.ctypes.data, m, plan.xs_ptr, plan.n, plan.ks_ref, plan.mean, mu_h.ctypes.data, plan.work_ptr,
                           plan.stream_arg), "apgp_predict_mean_host")
        return mu_h

    def acquire(self, y, t, kind, bounds=None, mask=None, zeta=0.01, return_all=False,
                idx_offset=0, device_record=False):
        """Fused predict + utility + arg-min over the candidate matrix ``t`` (M,D).

        The batched counterpart of utility.minimizeObjective (utility.py:253-372)
        for ``kind`` in {"agp","bape","jones"}.  ``bounds`` (sequence of (lo,hi))
        is the box prior fused into the kernel (candidates outside get +inf as
        utility.py:126-127 does); ``mask`` (M,) uint8/bool marks admissible
        candidates for arbitrary priors evaluated on the host.

        Returns (best_index, best_u) or, with return_all, additionally the
        arrays (u, mu, var).  best_index is -1 when no candidate is admissible.
        ``device_record``: return the sweep's 16-byte ``apgp_best_t`` record where the arg-min
        kernel left it -- a device int64[2] tensor (bit pattern of best_u, best_index), stream-ordered,
        no copy to the host: what ``dist.sharded_acquire`` hands to the RCCL all-gather.
        """
        if not self.computed:
            raise RuntimeError("ERROR: Need to compute GP before using it!")
        kind_id = UTILITY_KINDS[str(kind).lower()]
        want = ("best", "u", "mu", "var") if return_all else ("best",)
        if device_record:
            if return_all:
                raise ValueError("device_record returns the arg-min record only")
            want = ("best_device",)
        if hasattr(t, "data_ptr"):      # candidates already resident in HBM (torch tensor)
            if t.dim() != 2 or t.shape[1] != self.kernel.ndim or not t.is_contiguous() \
                    or str(t.dtype) != "torch.float64" or not t.is_cuda:
                raise ValueError("device candidates must be a contiguous (M, D) float64 CUDA tensor")
            res = self._sweep(y, None, kind=kind_id, want=want, bounds=bounds, mask=mask,
                              zeta=zeta, idx_offset=idx_offset, cand_device=t)
        else:
            xs = self.parse_samples(t)
            res = self._sweep(y, xs, kind=kind_id, want=want, bounds=bounds, mask=mask,
                              zeta=zeta, idx_offset=idx_offset)
        return res[0] if device_record else res

    def _sweep(self, y, cand, kind, want, bounds=None, mask=None, zeta=0.01, idx_offset=0,
               cand_device=None):
        torch, dev, lib = self._rt()
        y = self._check_dimensions(y)
        n = len(self._x)
        need_var = kind is not None or "var" in want
        ks = self._kernel_struct()
        use_solve = need_var and not self._trust_inverse()
        with self._on(torch, dev):
            st = self._stream(torch)
            one = need_var and kind is None and cand_device is None and cand is not None and len(cand) == 1
            if need_var and not use_solve:
                self._ensure_linv()     # first: with W resident alpha is two matrix-vector products
            elif need_var and not one:
                self._ensure_lsolve()   # (a single candidate solves against L itself)
            self._ensure_xs(y)
            if not need_var and cand_device is None and 0 < len(cand) <= 4096:
                # latency-bound mean-only call (the sampler's _gpll batches): host buffers
                # in and out through ONE library call and one synchronisation
                m = len(cand)
                need = m * (ks.ndim + 1)
                if self._mean_work is None or self._mean_work.numel() < need:
                    self._mean_work = torch.empty(max(need, 1024), dtype=torch.float64, device=dev)
                mu_h = np.empty(m, dtype=np.float64)
                _lib.check(lib.apgp_predict_mean_host(cand.ctypes.data, m, self._xs.data_ptr(), n,
                                                      ctypes.byref(ks), float(self.mean.value),
                                                      mu_h.ctypes.data, self._mean_work.data_ptr(), st),
                           "apgp_predict_mean_host")
                # the sampler asks again, for another few points of the same model and y, 4e4 times per chain
                self._mean_plan = _MeanPlan(self, torch, dev, st.value or 0, ks, y, n)
                return (mu_h,)
            if need_var and kind is None and cand_device is None and len(cand) == 1:
                # ONE candidate with variance: the reference's scalar utilities (utility.py:131,178,224), once per
                # Nelder-Mead step of minimizeObjective -- three small launches, the result through the mailbox
                if self._p1_work is None or self._p1_work.numel() < int(lib.apgp_predict1_work_len(n)):
                    self._p1_work = torch.empty(int(lib.apgp_predict1_work_len(n)), dtype=torch.float64, device=dev)
                o2 = np.empty(2, dtype=np.float64)
                if use_solve:
                    _lib.check(lib.apgp_predict1_host(cand.ctypes.data, self._xs.data_ptr(), n, ctypes.byref(ks),
                                                      float(self.mean.value), None, 0, self._L.data_ptr(), self._ld,
                                                      self._p1_work.data_ptr(), o2.ctypes.data, st), "apgp_predict1_host")
                else:
                    _lib.check(lib.apgp_predict1_host(cand.ctypes.data, self._xs.data_ptr(), n, ctypes.byref(ks),
                                                      float(self.mean.value), self._work.data_ptr(), (n + 63) // 64 * 64,
                                                      None, 0, self._p1_work.data_ptr(), o2.ctypes.data, st),
                               "apgp_predict1_host")
                if want == ("mu", "var"):
                    # the reference's scalar utilities ask again at the next simplex point, ~460 times per search
                    self._one_plan = _OnePlan(self, torch, dev, st.value or 0, ks, y, n, use_solve)
                res = {"mu": np.array([o2[0]]), "var": np.array([o2[1]])}
                return tuple(res[w_] for w_ in want)
            T = cand_device if cand_device is not None else torch.from_numpy(cand).to(dev)
            m = T.shape[0]
            if m == 0:
                # empty candidate set: nothing admissible (index -1, +inf), empty arrays
                empty = {"best": (-1, float("inf")), "mu": (np.empty(0),), "var": (np.empty(0),),
                         "u": (np.empty(0),)}
                if "best_device" in want:
                    empty["best_device"] = (torch.tensor([int(np.float64(np.inf).view(np.int64)), -1],
                                                         dtype=torch.int64, device=dev),)
                return tuple(v for w_ in want for v in empty[w_])
            if not need_var:
                mu = torch.empty(m, dtype=torch.float64, device=dev)
                _lib.check(lib.apgp_predict_mean(T.data_ptr(), m, self._xs.data_ptr(), n,
                                                 ctypes.byref(ks), float(self.mean.value),
                                                 mu.data_ptr(), st), "apgp_predict_mean")
                return (mu.cpu().numpy(),)
            mu = torch.empty(m, dtype=torch.float64, device=dev) if "mu" in want else None
            var = torch.empty(m, dtype=torch.float64, device=dev) if "var" in want else None
            u = torch.empty(m, dtype=torch.float64, device=dev) if "u" in want else None
            nwork = int(lib.apgp_acquire_work_len(m, n))
            part = torch.empty(max(nwork, 2), dtype=torch.float64, device=dev)
            best = torch.empty(2, dtype=torch.float64, device=dev)
            lo = hi = None
            if bounds is not None:
                b = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
                if len(b) != ks.ndim:
                    raise ValueError("bounds must have one (lo, hi) pair per dimension")
                lo = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 0])
                hi = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 1])
            mask_d = None
            if mask is not None:
                mk = np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
                if mk.shape != (m,):
                    raise ValueError("mask must have one entry per candidate")
                mask_d = torch.from_numpy(mk).to(dev)
            kid = _lib.UTIL_NONE if kind is None else kind
            ybest = float(np.max(y))
            ev = getattr(self, "kernel_events", None)   # bench.py: HIP events around the launch
            if ev is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            common = (ctypes.byref(ks), float(self.mean.value), kid, lo, hi,
                      mask_d.data_ptr() if mask_d is not None else None, float(zeta), ybest,
                      mu.data_ptr() if mu is not None else None,
                      var.data_ptr() if var is not None else None,
                      u.data_ptr() if u is not None else None,
                      part.data_ptr(), best.data_ptr(), st)
            if use_solve:
                _lib.check(lib.apgp_acquire_solve(T.data_ptr(), m, int(idx_offset), self._packed_solve.data_ptr(),
                                                  self._xs.data_ptr(), n, *common), "apgp_acquire_solve")
            else:
                _lib.check(lib.apgp_acquire(T.data_ptr(), m, int(idx_offset), self._packed.data_ptr(),
                                            self._xs.data_ptr(), n, *common), "apgp_acquire")
            if ev is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                ev.append((e0, e1))
            out = []
            for w in want:
                if w == "best_device":
                    import torch as _t
                    out.append(best.view(_t.int64))
                elif w == "best":
                    bb = best.cpu().numpy()
                    out.append(int(bb[1:2].view(np.int64)[0]))
                    out.append(float(bb[0]))
                elif w == "mu":
                    out.append(mu.cpu().numpy())
                elif w == "var":
                    out.append(var.cpu().numpy())
                elif w == "u":
                    out.append(u.cpu().numpy())
        return tuple(out)

    # -- candidate matrix of the sweep drawn on the device -----------------------------
    def box_candidates(self, m, bounds, seed, idx_offset=0):
        """Rows ``idx_offset .. idx_offset + m - 1`` of the global candidate matrix ``U[bounds]`` keyed by ``seed``
        (counter-based Philox: a row depends on (seed, row number) only), as a device tensor (m, D) ready for
        :meth:`acquire` -- the batched counterpart of the ``sampleFn`` draws utility.minimizeObjective starts from
        (utility.py:334-338) without the host draw and the H2D copy (26 ms per 1e6 x 8 against an 18 ms sweep)."""
        torch, dev, lib = self._rt()
        D = self.kernel.ndim
        b = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
        if len(b) != D:
            raise ValueError("bounds must have one (lo, hi) pair per dimension")
        lo = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 0])
        hi = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 1])
        with self._on(torch, dev):
            T = torch.empty((int(m), D), dtype=torch.float64, device=dev)
            if int(m) == 0:
                return T
            _lib.check(lib.apgp_box_candidates(T.data_ptr(), int(m), D, lo, hi, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                               int(idx_offset), self._stream(torch)), "apgp_box_candidates")
        return T

    # -- on-device ensemble MCMC over the GP mean ------------------------------------
    def sample_ensemble(self, y, initial_state, iterations, bounds, a=2.0, seed=0, store=True):
        """Run the stretch-move ensemble sampler entirely on the device with
        log-probability = GP mean (what ApproxPosterior._gpll returns) and the box
        prior ``bounds``.  ``initial_state`` is (W, D) for one ensemble or (E, W, D)
        for E independent ensembles (one workgroup each).  Returns a dict with
        ``chain`` (iterations, E*W, D), ``log_prob`` (iterations, E*W), ``coords``,
        ``final_log_prob`` and ``naccept``."""
        self.recompute()
        torch, dev, lib = self._rt()
        y = self._check_dimensions(y)
        p0 = np.ascontiguousarray(np.asarray(initial_state, dtype=np.float64))
        D = self.kernel.ndim
        if p0.ndim == 1:
            p0 = p0.reshape(-1, D)
        if p0.ndim == 2:
            p0 = p0[None]
        if p0.ndim != 3 or p0.shape[2] != D:
            raise ValueError("initial_state must be (W, D) or (E, W, D)")
        if not np.all(np.isfinite(p0)):
            raise ValueError("At least one parameter value was NaN or infinite")
        E, W, _ = p0.shape
        b = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
        if len(b) != D:
            raise ValueError("bounds must have one (lo, hi) pair per dimension")
        lo = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 0])
        hi = (ctypes.c_double * _lib.MAX_DIM)(*b[:, 1])
        n = len(self._x)
        ks = self._kernel_struct()
        iterations = int(iterations)
        with torch.cuda.device(dev):
            st = self._stream(torch)
            self._ensure_xs(y)
            coords = torch.from_numpy(p0).to(dev)
            logp = torch.empty((E, W), dtype=torch.float64, device=dev)
            nacc = torch.empty((E, W), dtype=torch.int64, device=dev)
            chain = torch.empty((iterations, E, W, D), dtype=torch.float64, device=dev) if store else None
            lchain = torch.empty((iterations, E, W), dtype=torch.float64, device=dev) if store else None
            def launch(mode):
                coords.copy_(torch.from_numpy(p0))
                _lib.check(lib.apgp_ensemble_sample_ex(
                    self._xs.data_ptr(), n, ctypes.byref(ks), float(self.mean.value), lo, hi, W, E,
                    iterations, float(a), int(seed) & 0xFFFFFFFFFFFFFFFF, coords.data_ptr(), logp.data_ptr(),
                    chain.data_ptr() if store else None, lchain.data_ptr() if store else None,
                    nacc.data_ptr(), mode, st), "apgp_ensemble_sample")
                return logp.cpu().numpy().reshape(E * W)
            final = launch(-1)
            if np.any(np.isnan(final)):
                # the multi-workgroup launch could not get its workgroups resident (another stream or process holds
                # compute units; the launch's sticky word turned every log-probability into NaN): once more on the
                # single-workgroup kernel -- slower, same posterior -- chosen for THIS call only (no process-wide switch
                # is flipped under other threads' calls)
                self.ensemble_fallbacks = getattr(self, "ensemble_fallbacks", 0) + 1
                final = launch(1)
                if np.any(np.isnan(final)):
                    raise FloatingPointError("the ensemble sampler returned NaN log-probabilities on both of its kernels: "
                                             "the GP mean is not finite at the walkers' positions")
            out = {"coords": coords.cpu().numpy().reshape(E * W, D),
                   "final_log_prob": final,
                   "naccept": nacc.cpu().numpy().reshape(E * W),
                   "chain": chain.cpu().numpy().reshape(iterations, E * W, D) if store else None,
                   "log_prob": lchain.cpu().numpy().reshape(iterations, E * W) if store else None}
        return out

    # -- K4: gradient of the log-likelihood ------------------------------------------
    def grad_log_likelihood(self, y, quiet=False):
        """george GP.grad_log_likelihood (gpUtils.py:110): zeros on failure when quiet."""
        try:
            if not self.recompute(quiet=quiet):
                return np.zeros(len(self), dtype=np.float64)
            torch, dev, lib = self._rt()
            y = self._check_dimensions(y)
            n = len(y)
            ks = self._kernel_struct()
            with torch.cuda.device(dev):
                st = self._stream(torch)
                self._solve(y, need_alpha=True)
                work = torch.empty(lib.apgp_grad_work_len(n), dtype=torch.float64, device=dev)
                out = torch.empty(4 + _lib.MAX_DIM, dtype=torch.float64, device=dev)
                np64 = (n + 63) // 64 * 64
                if self._trust_inverse():
                    # K^-1 = W^T W with the resident dense W = L^-1 (one MFMA-f64 product)
                    self._ensure_linv()
                    winv = self._work.data_ptr()
                else:
                    # above the conditioning gate: K^-1 by two triangular solves against the identity, as george's
                    # cho_solve(L, I) (gpUtils.py:110 -> GP.grad_log_likelihood) -- never a product of inverses
                    xw = torch.empty(int(lib.apgp_kinv_solve_work_len(n)), dtype=torch.float64, device=dev)
                    _lib.check(lib.apgp_kinv_solve(self._L.data_ptr(), n, self._ld, xw.data_ptr(), work.data_ptr(), st),
                               "apgp_kinv_solve")
                    winv = None
                _lib.check(lib.apgp_grad_loglik(self._x_d.data_ptr(), self._alpha.data_ptr(),
                                                winv, np64, n, ctypes.byref(ks),
                                                work.data_ptr(), out.data_ptr(), st),
                           "apgp_grad_loglik")
                o = out.cpu().numpy()
        except (ValueError, LinAlgError):
            if quiet:
                return np.zeros(len(self), dtype=np.float64)
            raise
        return self._assemble_gradient(o, ks.ndim)

    def _assemble_gradient(self, o, ndim):
        """Order device results as george orders its parameter vector."""
        grad = []
        if self.fit_mean:
            grad.append(o[0])
        if self.fit_white_noise:
            # d/d white_noise = 0.5 * exp(wn) * trace(alpha alpha^T - K^-1)  
            grad.append(float(np.exp(self.white_noise.value)) * o[2 + _lib.MAX_DIM])

        o_lin = o[3 + _lib.MAX_DIM]       # 0.5 sum A K_lin

        def walk(k, in_linear_term):
            if isinstance(k, Sum):
                for term in (k.k1, k.k2):
                    walk(term, _flatten_term(term)[2] is not None)
            elif isinstance(k, Product):
                walk(k.k1, in_linear_term); walk(k.k2, in_linear_term)
            elif isinstance(k, ConstantKernel):
                grad.append(o_lin if in_linear_term else o[1])     # d/d log_constant = that term
            elif isinstance(k, LinearKernel):
                grad.append(-o_lin)                                # d/d log_gamma2 = -K_lin
            else:
                grad.extend(o[2:2 + ndim])
        walk(self.kernel, False)
        return np.array(grad, dtype=np.float64)
