# -*- coding: utf-8 -*-
"""
approxposterior_amd -- MI355X-native GP-surrogate inner loop for
approxposterior-style inference.

Same module and function names as the reference package for the hot path
(``approx.ApproxPosterior``, ``gpUtils.defaultGP/optimizeGP``, ``utility.*``,
``likelihood.*``); ``gp`` stands in for the ``george`` names the reference
imports.  The HIP extension (csrc/libapgp.so) is loaded lazily on first use and
there is no CPU fallback.
"""

__version__ = "0.1.0"

from . import gp, gpUtils, utility, likelihood, mcmc, mcmcUtils, approx, dist  # noqa: F401
from .approx import ApproxPosterior  # noqa: F401
from .gpUtils import defaultHyperPrior, defaultGP, optimizeGP  # noqa: F401
from .mcmcUtils import validateMCMCKwargs, batchMeansMCSE, estimateBurnin  # noqa: F401
from .utility import (logsubexp, AGPUtility, BAPEUtility, JonesUtility,  # noqa: F401
                      minimizeObjective, sweepObjective)
