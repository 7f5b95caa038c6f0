"""Developer check (GPU box): apgp_potrf (+ riding forward solve), apgp_trsv both ways and the dense L^-1 of
apgp_trtri_pack through the C ABI at ~50 random sizes in [1, 1700] (incl. the 768 threshold of the blocked
solve) against LAPACK; prints the worst errors in units of cond * eps."""
import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scipy.linalg import solve_triangular
from approxposterior_amd import _lib
lib = _lib.load()
rs = np.random.RandomState(7)
st = torch.cuda.current_stream().cuda_stream
worst = {"L": 0, "z": 0, "W": 0, "tf": 0, "tb": 0}
sizes = sorted(set([int(x) for x in rs.randint(1, 1700, size=45)] + [767, 768, 769, 1023, 1025, 1535, 1600]))
for n in sizes:
    X = rs.uniform(-3, 3, size=(n, 3))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    K = np.exp(-0.5 * d2) + 1e-5 * np.eye(n)
    y = rs.randn(n)
    Lr = np.linalg.cholesky(K)
    cond = np.linalg.cond(Lr)
    A = torch.from_numpy(np.tril(K)).cuda(); yd = torch.from_numpy(y).cuda()
    z = torch.empty(n, dtype=torch.float64, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.apgp_potrf(A.data_ptr(), n, n, yd.data_ptr(), 0.1, z.data_ptr(), info.data_ptr(), st) == 0
    torch.cuda.synchronize(); assert int(info.item()) == 0
    L = np.tril(A.cpu().numpy())
    eL = np.abs(L - Lr).max() / np.abs(Lr).max() / (cond ** 2 * 2.2e-16)
    zr = solve_triangular(Lr, y - 0.1, lower=True)
    ez = np.abs(z.cpu().numpy() - zr).max() / max(1, np.abs(zr).max()) / (cond ** 2 * 2.2e-16)
    # trsv both ways on the exact factor
    Ld = torch.from_numpy(Lr).cuda(); x = torch.empty(n, dtype=torch.float64, device="cuda"); ss = torch.empty(1, dtype=torch.float64, device="cuda")
    assert lib.apgp_trsv(Ld.data_ptr(), n, n, yd.data_ptr(), 0.1, 0, x.data_ptr(), ss.data_ptr(), st) == 0
    xf = x.cpu().numpy(); etf = np.abs(xf - zr).max() / max(1, np.abs(zr).max()) / (cond * 2.2e-16)
    assert abs(ss.item() - xf @ xf) <= 1e-12 * (xf @ xf)
    ar = solve_triangular(Lr, y, lower=True, trans="T")
    assert lib.apgp_trsv(Ld.data_ptr(), n, n, yd.data_ptr(), 0.0, 1, x.data_ptr(), None, st) == 0
    etb = np.abs(x.cpu().numpy() - ar).max() / max(1, np.abs(ar).max()) / (cond * 2.2e-16)
    # inverse
    work = torch.empty(int(lib.apgp_trtri_work_len(n)), dtype=torch.float64, device="cuda")
    packed = torch.empty(int(lib.apgp_packed_linv_len(n)), dtype=torch.float64, device="cuda")
    Wd = torch.empty((n, n), dtype=torch.float64, device="cuda")
    assert lib.apgp_trtri_pack(Ld.data_ptr(), n, n, work.data_ptr(), packed.data_ptr(), Wd.data_ptr(), st) == 0
    Wr = solve_triangular(Lr, np.eye(n), lower=True)
    eW = np.abs(Wd.cpu().numpy() - Wr).max() / np.abs(Wr).max() / (cond * 2.2e-16)
    for k, v in (("L", eL), ("z", ez), ("W", eW), ("tf", etf), ("tb", etb)):
        worst[k] = max(worst[k], v)
    assert max(eL, ez) < 20 and max(eW, etf, etb) < 50, (n, eL, ez, eW, etf, etb)
print("sizes", len(sizes), "worst errors in units of cond*eps (cond^2 for the factor):", {k: round(v, 3) for k, v in worst.items()})
