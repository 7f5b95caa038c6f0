"""Developer timing of the fit path pieces (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp, gpUtils
from scipy.optimize import rosen

def sync(): torch.cuda.synchronize()

for N, D in ((50, 2), (512, 8), (1152, 8), (4096, 8)):
    rs = np.random.RandomState(0)
    X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
    k = agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
    gp = agp.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(X); gp.log_likelihood(y); sync()
    p = gp.get_parameter_vector()
    t0 = time.time(); R = 20
    for i in range(R):
        gp.set_parameter_vector(p + 1e-3 * (i % 3)); ll = gp.log_likelihood(y, quiet=True)
    sync(); t_nll = (time.time() - t0) / R
    # pieces
    import ctypes
    from approxposterior_amd import _lib
    lib = _lib.load(); ks = gp._kernel_struct(); st = gp._stream(torch)
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    def timeit(f, R=20):
        f(); sync(); t0 = time.time()
        for _ in range(R): f()
        sync(); return (time.time() - t0) / R
    t_gram = timeit(lambda: lib.apgp_gram(gp._x_d.data_ptr(), N, ctypes.byref(ks), K.data_ptr(), N, st))
    Ks = torch.tril(K) + torch.tril(K, -1).T          # apgp_gram writes the lower triangle only
    t_chol = timeit(lambda: torch.linalg.cholesky_ex(Ks, upper=True))
    info = torch.empty(1, dtype=torch.int32, device='cuda'); K2 = K.clone()
    def own():
        K2.copy_(K); lib.apgp_potrf(K2.data_ptr(), N, N, None, 0.0, None, info.data_ptr(), st)
    t_copy = timeit(lambda: K2.copy_(K))
    t_own = timeit(own) - t_copy
    yd = torch.from_numpy(y).cuda(); z = torch.empty_like(yd); s = torch.empty(1, dtype=torch.float64, device="cuda")
    t_trsv = timeit(lambda: lib.apgp_trsv(gp._L.data_ptr(), N, N, yd.data_ptr(), 0.0, 0, z.data_ptr(), s.data_ptr(), st))
    t_trsvb = timeit(lambda: lib.apgp_trsv(gp._L.data_ptr(), N, N, yd.data_ptr(), 0.0, 1, z.data_ptr(), None, st))
    work = torch.empty(lib.apgp_trtri_work_len(N), dtype=torch.float64, device="cuda")
    packed = torch.empty(lib.apgp_packed_linv_len(N), dtype=torch.float64, device="cuda")
    t_tri = timeit(lambda: lib.apgp_trtri_pack(gp._L.data_ptr(), N, N, work.data_ptr(), packed.data_ptr(), None, st), R=5)
    t_grad = timeit(lambda: gp.grad_log_likelihood(y), R=3)
    wk = torch.empty(int(lib.apgp_winv_apply_work_len(N)), dtype=torch.float64, device="cuda")
    NP = (N + 63) // 64 * 64
    t_wf = timeit(lambda: lib.apgp_winv_apply(work.data_ptr(), NP, N, yd.data_ptr(), 0.0, 0, z.data_ptr(), s.data_ptr(), None, st))
    t_wb = timeit(lambda: lib.apgp_winv_apply(work.data_ptr(), NP, N, yd.data_ptr(), 0.0, 1, z.data_ptr(), None, wk.data_ptr(), st))
    def setup():
        g2 = agp.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
        g2.compute(X); g2._ensure_linv(); g2._ensure_xs(y)
    t_setup = timeit(setup, R=5)
    print("N=%5d  winv_apply fwd %.3f ms, bwd %.3f ms | sweep set-up (compute + L^-1 + pack + alpha + xs) %.3f ms" % (N, t_wf*1e3, t_wb*1e3, t_setup*1e3))
    T = torch.rand((64, D), dtype=torch.float64, device="cuda") * 10 - 5
    gp.recompute(); gp._ensure_xs(y)
    mu = torch.empty(64, dtype=torch.float64, device="cuda")
    t_mean = timeit(lambda: lib.apgp_predict_mean(T.data_ptr(), 64, gp._xs.data_ptr(), N, ctypes.byref(ks), 0.0, mu.data_ptr(), st))
    t_pred64 = timeit(lambda: gp.predict(y, T.cpu().numpy(), return_cov=False, return_var=False))
    print("N=%5d D=%d  _nll eval %.3f ms | gram %.3f chol(rocsolver) %.3f chol(own) %.3f trsv_fwd %.3f trsv_bwd %.3f trtri+pack %.3f grad %.3f | mean(64 pts) kernel %.3f ms, python predict %.3f ms"
          % (N, D, t_nll*1e3, t_gram*1e3, t_chol*1e3, t_own*1e3, t_trsv*1e3, t_trsvb*1e3, t_tri*1e3, t_grad*1e3, t_mean*1e3, t_pred64*1e3))
