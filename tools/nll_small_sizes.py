"""apgp_nll_eval latency at README sizes (N = 32 .. 384, D = 2): where the single-workgroup fused kernel (n <= 64) hands over to
gram + persistent launch + finish.  Usage (GPU box): python tools/nll_small_sizes.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from approxposterior_amd import _lib, gp as agp
lib = _lib.load(); out = []
for n in (32, 50, 64, 65, 90, 128, 129, 192, 256, 320, 384):
    D = 2; rs = np.random.RandomState(n)
    X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False); g._x = X; g._yerr2 = 0.0
    ks = g._kernel_struct()
    X_d = torch.from_numpy(X).cuda(); y_d = torch.from_numpy(y).cuda()
    K = torch.zeros((n, n), dtype=torch.float64, device="cuda"); z = torch.empty(n, dtype=torch.float64, device="cuda")
    info = torch.empty(1, dtype=torch.int32, device="cuda"); o5 = torch.empty(5, dtype=torch.float64, device="cuda"); o = np.empty(5)
    args = (X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), 0.0, K.data_ptr(), z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
    for _ in range(20): lib.apgp_nll_eval(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): lib.apgp_nll_eval(*args)
    torch.cuda.synchronize(); out.append("%d: %.1f" % (n, (time.perf_counter() - t0) / 300 * 1e6))
print("apgp_nll_eval us:", " | ".join(out))
