"""cProfile of the Python path of gpUtils._nll at C5's mid size (N = 832, D = 8): where the host microseconds between two
device evaluations go (GPU box).  Usage: python tools/nll_pyprofile.py [n]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from approxposterior_amd import gp as agp, gpUtils
from bench import synthetic_c3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 832
X, y = synthetic_c3(n, 8)
g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
g.compute(X)
p = g.get_parameter_vector()
ps = [p + 1e-4 * np.random.RandomState(i).randn(len(p)) for i in range(3000)]
for q in ps[:50]:
    gpUtils._nll(q, g, y, gpUtils.defaultHyperPrior)
t0 = time.perf_counter()
for q in ps[50:1050]:
    gpUtils._nll(q, g, y, gpUtils.defaultHyperPrior)
dt = (time.perf_counter() - t0) / 1000
print("gpUtils._nll at N = %d: %.1f us per call" % (n, dt * 1e6))
# the same evaluations straight through the C ABI (what the device path alone costs)
import ctypes, torch
from approxposterior_amd import _lib
lib = _lib.load()
ks = g._kernel_struct()
X_d, y_d = torch.from_numpy(X).cuda(), torch.from_numpy(np.ascontiguousarray(y)).cuda()
K = torch.zeros((n, n), dtype=torch.float64, device="cuda"); z = torch.empty(n, dtype=torch.float64, device="cuda")
info = torch.empty(1, dtype=torch.int32, device="cuda"); o5 = torch.empty(5, dtype=torch.float64, device="cuda"); o = np.empty(5)
args = (X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), float(np.median(y)), K.data_ptr(), z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
for _ in range(50): lib.apgp_nll_eval(*args)
t0 = time.perf_counter()
for _ in range(1000): lib.apgp_nll_eval(*args)
dd = (time.perf_counter() - t0) / 1000
print("apgp_nll_eval alone: %.1f us per call -> Python path %.1f us" % (dd * 1e6, (dt - dd) * 1e6))
pr = cProfile.Profile(); pr.enable()
for q in ps[1050:3000]:
    gpUtils._nll(q, g, y, gpUtils.defaultHyperPrior)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:4000])
