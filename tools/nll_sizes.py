"""apgp_nll_eval (Gram + Cholesky + summary, default path) timed back to back over a range of training-set sizes; with a
library path as the first argument an experimental build is timed instead of the shipped one (A/B of kernel variants).
NLL_MODE=<m> sets apgp_potrf_mode(m) first.
Usage (GPU box): [NLL_MODE=m] python tools/nll_sizes.py [path/to/libapgp.so]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from approxposterior_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from approxposterior_amd import gp as agp
lib = _lib.load(); dev = torch.device("cuda:0")
if os.environ.get("NLL_MODE"): lib.apgp_potrf_mode(int(os.environ["NLL_MODE"]))     # e.g. 1 = launch per step, 32 = no deferred tiles
out = []
for n in (512, 800, 1152, 1600, 2048, 2560, 3072, 3712, 4096):
    D = 8; rs = np.random.RandomState(n)
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    k = agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
    g = agp.GP(kernel=k, fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False); g._x = X; g._yerr2 = 0.0
    ks = g._kernel_struct()
    X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
    K = torch.zeros((n, n), dtype=torch.float64, device=dev); z = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev); o5 = torch.empty(5, dtype=torch.float64, device=dev); o = np.empty(5)
    args = (X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), 0.0, K.data_ptr(), z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
    for _ in range(5): lib.apgp_nll_eval(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): lib.apgp_nll_eval(*args)
    torch.cuda.synchronize(); out.append("%d: %.3f" % (n, (time.perf_counter() - t0) / 40 * 1e3))
print(sys.argv[1:] or "shipped", "mode", os.environ.get("NLL_MODE", "0"), " | ".join(out), "fallbacks", lib.apgp_potrf_fallbacks())
