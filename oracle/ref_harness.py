# -*- coding: utf-8 -*-
"""
oracle/ref_harness.py -- TEST INFRASTRUCTURE (runs ONLY in the build container).

Imports the *reference* approxposterior package from /root/reference (read-only,
never copied, never shipped) on top of stub ``george`` / ``emcee`` modules so
that the reference's own control flow and acquisition formulas
(``gpUtils.py``, ``utility.py``, ``approx.py``, ``likelihood.py``) can be
driven in this container, where george/emcee are not installable.

The stub ``george`` is bound to ``oracle/george_oracle.py`` (the NumPy
restatement of george's GP algebra).  Everything else that runs is the
reference's code.  Used by ``oracle/make_golden.py`` to write the fixtures in
``tests/golden/``; nothing here is importable on the GPU box (no
/root/reference there) and nothing in ``tests -m gpu`` / ``bench.py`` /
``smoke()`` uses it.

Harness-side shims (recorded in the fixtures' metadata):
  Q1/Q2: ``utility.py:336,351`` hands a (1,D) x0 to
  ``scipy.optimize.minimize`` and the utilities return shape-(1,) arrays;
  SciPy >= 1.11 rejects both.  ``utility.minimize`` is rebound to a wrapper that
  ravels x0 and casts the objective to float.  No reference file is modified.
"""

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def load_reference():
    """Return the reference ``approxposterior`` package (stub george/emcee)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present (expected only in the "
                           "build container)")
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import george_oracle

    george = types.ModuleType("george")
    george.GP = george_oracle.GP
    george.kernels = george_oracle.kernels
    george.__version__ = "0.3.1-oracle-restatement"
    sys.modules["george"] = george
    sys.modules["george.kernels"] = george_oracle.kernels

    emcee = types.ModuleType("emcee")
    emcee.__version__ = "3.0.0-stub"
    sys.modules.setdefault("emcee", emcee)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import approxposterior  # noqa: E402  (the reference package)

    # Q1/Q2 shim, harness side only.
    import numpy as np
    from scipy.optimize import minimize as _sp_minimize
    from approxposterior import utility as ref_utility

    def _minimize(fn, x0, args=(), **kw):
        def f(x, *a):
            return float(np.asarray(fn(x, *a)).ravel()[0])
        return _sp_minimize(f, np.asarray(x0, dtype=float).ravel(), args=args, **kw)

    ref_utility.minimize = _minimize
    return approxposterior
