"""The reference README's Rosenbrock example, unchanged in structure, on the MI355X path:
only the import line differs (``approxposterior`` -> ``approxposterior_amd``).

    python examples/rosenbrock_bape.py            # host-loop sampler, any lnprior
    python examples/rosenbrock_bape.py --device   # final MCMC inside one persistent kernel (box prior)
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
from approxposterior_amd import approx, gpUtils, likelihood as lh

m0, m, nmax = 50, 20, 2                       # initial design, points per iteration, iterations
bounds = [(-5, 5), (-5, 5)]
np.random.seed(57)

theta = lh.rosenbrockSample(m0)
y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
gp = gpUtils.defaultGP(theta, y, white_noise=-12)

ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lh.rosenbrockLnlike,
                            priorSample=lh.rosenbrockSample, bounds=bounds, algorithm="bape")
ap.run(m=m, nmax=nmax, estBurnin=True, nGPRestarts=3, mcmcKwargs={"iterations": 20000},
       samplerKwargs={"nwalkers": 20}, cache=False, verbose=True, thinChains=False, onlyLastMCMC=True)

if "--device" in sys.argv:
    ap.runMCMC(samplerKwargs={"nwalkers": 20}, mcmcKwargs={"iterations": 20000}, cache=False, onDevice=True)
samples = ap.sampler.get_chain(discard=ap.iburns[-1], flat=True, thin=ap.ithins[-1])
print("posterior mean", samples.mean(axis=0), "std", samples.std(axis=0), "from", len(samples), "samples;",
      "training set", len(ap.y))
