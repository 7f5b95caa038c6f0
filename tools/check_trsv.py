"""Persistent triangular solve (linalg.hip trsv_persist_kernel) against the launch-per-256-rows path on the GPU box:
bit-identity of x and x.x, forward and transposed, at ragged sizes, aliasing b = x, repeated calls; timings of both."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import _lib
if "--lib" in sys.argv:          # an experimental build: python tools/check_trsv.py --lib path/to/libapgp.so
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
lib = _lib.load()
dev = torch.device("cuda:0")
bad = 0


def run(L, b, n, ld, trans, mode, alias=False):
    lib.apgp_trsv_mode(mode)
    x = b.clone() if alias else torch.empty(n, dtype=torch.float64, device=dev)
    ss = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
    src = x if alias else b
    rc = lib.apgp_trsv(L.data_ptr(), n, ld, src.data_ptr(), 0.125, trans, x.data_ptr(), ss.data_ptr(), None)
    assert rc == 0, lib.apgp_last_error()
    torch.cuda.synchronize()
    lib.apgp_trsv_mode(0)
    return x, float(ss.item())


for n in (256, 300, 512, 1100, 1152, 1153, 2048, 4095, 4096, 8000):
    rs = np.random.RandomState(n)
    ld = n + (3 if n % 2 else 0)
    A = rs.normal(size=(n, n)) * 0.05
    Lh = np.tril(A) + np.diag(1.0 + rs.uniform(size=n))
    Lbuf = np.zeros((n, ld)); Lbuf[:, :n] = Lh
    L = torch.from_numpy(Lbuf).to(dev)
    b = torch.from_numpy(rs.normal(size=n)).to(dev)
    for trans in (0, 1):
        x1, s1 = run(L, b, n, ld, trans, 1)
        x0, s0 = run(L, b, n, ld, trans, 0)
        xa, sa = run(L, b, n, ld, trans, 0, alias=True)
        ref = np.linalg.solve(Lh.T if trans else Lh, b.cpu().numpy() - 0.125)
        same = torch.equal(x0, x1) and s0 == s1 and torch.equal(xa, x1) and sa == s1
        err = np.abs(x0.cpu().numpy() - ref).max() / np.abs(ref).max()
        print("n=%5d trans=%d identical=%s (x diffs %d, x.x %r vs %r) rel err vs LAPACK %.1e" % (
            n, trans, same, int((x0 != x1).sum().item()), s0, s1, err), flush=True)
        bad += (not same) or err > 1e-10
# repeated calls, two sizes alternating on one stream
cases = []
for n in (1152, 700):
    rs = np.random.RandomState(n + 1)
    Lh = np.tril(rs.normal(size=(n, n)) * 0.05) + np.diag(1.0 + rs.uniform(size=n))
    cases.append((n, torch.from_numpy(Lh).to(dev), torch.from_numpy(rs.normal(size=n)).to(dev)))
first = [[run(L, b, n, n, tr, 0) for tr in (0, 1)] for n, L, b in cases]
dev_ = 0
for it in range(100):
    for (n, L, b), f in zip(cases, first):
        for tr in (0, 1):
            x, s = run(L, b, n, n, tr, 0)
            dev_ += not (torch.equal(x, f[tr][0]) and s == f[tr][1])
print("stress: %d deviating calls" % dev_)
bad += dev_
for n in (512, 1152, 4096):
    rs = np.random.RandomState(n)
    Lh = np.tril(rs.normal(size=(n, n)) * 0.05) + np.diag(1.0 + rs.uniform(size=n))
    L = torch.from_numpy(Lh).to(dev); b = torch.from_numpy(rs.normal(size=n)).to(dev)
    x = torch.empty(n, dtype=torch.float64, device=dev); ss = torch.empty(1, dtype=torch.float64, device=dev)
    for trans in (0, 1):
        res = []
        for mode in (1, 0):
            lib.apgp_trsv_mode(mode)
            for _ in range(3):
                lib.apgp_trsv(L.data_ptr(), n, n, b.data_ptr(), 0.0, trans, x.data_ptr(), ss.data_ptr(), None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                lib.apgp_trsv(L.data_ptr(), n, n, b.data_ptr(), 0.0, trans, x.data_ptr(), ss.data_ptr(), None)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 50)
        lib.apgp_trsv_mode(0)
        print("apgp_trsv n=%4d trans=%d: multi-launch %.3f ms, persistent %.3f ms" % (n, trans, res[0] * 1e3, res[1] * 1e3), flush=True)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
