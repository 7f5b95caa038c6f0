import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp, _lib
from bench import synthetic_c3
n, d, m = int(sys.argv[1]), int(sys.argv[2]), 1000000
X, y = synthetic_c3(n, d)
T = torch.from_numpy(np.random.RandomState(1).uniform(-5, 5, size=(m, d))).cuda()
g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
g.variance_mode = "inverse"
g.compute(X)
for _ in range(2):
    b = g.acquire(y, T, "agp", bounds=[(-5, 5)] * d)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(ROOT, "approxposterior_amd/csrc/libapgp.so"))
buf = (ctypes.c_ulonglong * 8192)()
print("rc", lib.apgp_debug_stamps(buf))
s = np.array(buf, dtype=np.uint64).astype(np.int64).reshape(1024, 8)
t0 = s[0, 0]
print("tile  m_arrive  m_wait | f_start  f_prod0  f_prod1  f_arrive  f_leave | period")
for i in range(0, 170):
    r = s[i]
    if r[0] == 0: break
    nxt = s[i + 1, 0] - r[0] if s[i + 1, 0] else 0
    f = lambda v: (v - t0) if v else -1
    print("%4d %9d %7d | %8d %8d %8d %8d %8d | %6d" % (i, f(r[0]), r[1] - r[0], f(r[2]), f(r[3]), f(r[4]), f(r[5]), f(r[6]), nxt))
