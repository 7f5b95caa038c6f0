"""cProfile of the README example (examples/rosenbrock_bape.py: BASELINE config 1, N = 50 -> 90, 20 walkers x 2e4 host-loop
MCMC) on the GPU box: where its seconds go.  Usage: python tools/c1_profile.py"""
import cProfile, io, os, pstats, runpy, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from approxposterior_amd import gp as agp
g = agp.GP(kernel=agp.ExpSquaredKernel(np.ones(2), ndim=2)); g.compute(np.random.RandomState(0).uniform(size=(8, 2)))   # (library warm-up)
sys.argv = ["rosenbrock_bape.py"]
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
runpy.run_path(os.path.join(ROOT, "examples", "rosenbrock_bape.py"), run_name="__main__")
pr.disable()
print("example: %.2f s" % (time.perf_counter() - t0))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32)
print(s.getvalue()[:6500])
