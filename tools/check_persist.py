"""Persistent Cholesky (csrc/potrf_persist.h) against the multi-launch path on the GPU box: bit-identity of the
factor, z and the 5-value record over ragged sizes, non-positive-definite inputs (LAPACK info), the give-up /
fallback path, a repeated-call stress, and timings of both paths.
Usage: python tools/check_persist.py [quick] [--lib path/to/an/experimental/libapgp.so]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import _lib
from approxposterior_amd import gp as agp

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
lib = _lib.load()
dev = torch.device("cuda:0")
st = None
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def kern(D, amp=1.0, metric=8.0, wn=-12.0):
    k = agp.ExpSquaredKernel(np.full(D, metric), ndim=D)
    g = agp.GP(kernel=k, fit_mean=True, mean=0.0, white_noise=wn, fit_white_noise=False)
    g._x = np.zeros((1, D))
    g._yerr2 = 0.0
    return g._kernel_struct()


def nll(X_d, y_d, n, ks, mean, mode):
    lib.apgp_potrf_mode(mode)
    K = torch.zeros((n, n), dtype=torch.float64, device=dev)
    z = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    o5 = torch.empty(5, dtype=torch.float64, device=dev)
    o = np.empty(5)
    rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), mean, K.data_ptr(), z.data_ptr(),
                           info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
    assert rc == 0, lib.apgp_last_error()
    torch.cuda.synchronize()
    return torch.tril(K).clone(), z.clone(), o.copy(), int(info.item())


def timeit(X_d, y_d, n, ks, mean, mode, reps):
    lib.apgp_potrf_mode(mode)
    K = torch.zeros((n, n), dtype=torch.float64, device=dev)
    z = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.empty(1, dtype=torch.int32, device=dev)
    o5 = torch.empty(5, dtype=torch.float64, device=dev)
    o = np.empty(5)
    args = (X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), mean, K.data_ptr(), z.data_ptr(), info.data_ptr(),
            o5.data_ptr(), o.ctypes.data, None)
    for _ in range(3):
        lib.apgp_nll_eval(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.apgp_nll_eval(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


bad = 0
sizes = [65, 100, 128, 129, 192, 300, 512, 700, 1152, 1153] + ([] if quick else [2048, 3000, 4095, 4096])
for n in sizes:
    for D in ((8,) if n > 1200 else (2, 8)):
        rs = np.random.RandomState(n + D)
        X = rs.uniform(-5, 5, size=(n, D))
        y = rs.normal(size=n)
        X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
        ks = kern(D)
        L1, z1, o1, i1 = nll(X_d, y_d, n, ks, 0.25, 1)
        fb0 = lib.apgp_potrf_fallbacks()
        L0, z0, o0, i0 = nll(X_d, y_d, n, ks, 0.25, 3)
        fb1 = lib.apgp_potrf_fallbacks()
        same = torch.equal(L0, L1) and torch.equal(z0, z1) and np.array_equal(o0, o1) and i0 == i1
        nd = int((L0 != L1).sum().item()); ndz = int((z0 != z1).sum().item())
        if nd:
            idx = (L0 != L1).nonzero()[0].tolist()
        else:
            idx = None
        print("n=%5d D=%d  identical=%s  (L diffs %d, first at %s; z diffs %d; record %s vs %s; info %d/%d; fallbacks +%d)" %
              (n, D, same, nd, idx, ndz, o0.tolist() if not same else "=", o1.tolist() if not same else "=", i0, i1, fb1 - fb0), flush=True)
        if nd and "--diffs" in sys.argv:
            dd = (L0 - L1).abs()
            rows = (L0 != L1).any(dim=1).nonzero().flatten().tolist()
            r0 = rows[0]
            cols = (L0[r0] != L1[r0]).nonzero().flatten().tolist()
            print("      max |dL| %.3e (max |L| %.3e); first differing row %d: columns %s, |dL| there %s" %
                  (dd.max().item(), L1.abs().max().item(), r0, cols[:8], ["%.2e" % dd[r0, c].item() for c in cols[:8]]), flush=True)
        bad += (not same) or (fb1 != fb0)

# non-positive-definite: duplicated points, no white noise to speak of -> a pivot fails somewhere
for n, dup_at in ((200, 70), (700, 650), (1152, 64), (1152, 1100)):
    rs = np.random.RandomState(7 * n)
    X = rs.uniform(-5, 5, size=(n, 3)); X[dup_at] = X[dup_at - 1]; X[dup_at + 1] = X[dup_at - 1]
    y = rs.normal(size=n)
    X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
    ks = kern(3, wn=-60.0)
    L1, z1, o1, i1 = nll(X_d, y_d, n, ks, 0.0, 1)
    L0, z0, o0, i0 = nll(X_d, y_d, n, ks, 0.0, 0)
    ok = i0 == i1 and o0[4] == o1[4] and i0 > 0
    print("non-PD n=%d dup at %d: info persistent %d, multi-launch %d -> %s" % (n, dup_at, i0, i1, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok

# hybrid (default mode above PP_AUTO_NB = 58 block columns): the first block columns a launch per step, the last 48 as one
# persistent launch on the trailing matrix -- against the launch-per-step path
if not quick:
    for n, D, dup in ((3300, 8, None), (3800, 8, None), (4095, 5, None), (4096, 8, None), (5000, 8, None), (4096, 3, 3000), (4096, 3, 500)):
        rs = np.random.RandomState(n + D)
        X = rs.uniform(-5, 5, size=(n, D))
        if dup is not None:
            X[dup] = X[dup - 1]; X[dup + 1] = X[dup - 1]
        y = rs.normal(size=n)
        X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
        ks = kern(D, wn=-60.0) if dup is not None else kern(D)
        L1, z1, o1, i1 = nll(X_d, y_d, n, ks, 0.25, 1)
        fb0 = lib.apgp_potrf_fallbacks()
        L0, z0, o0, i0 = nll(X_d, y_d, n, ks, 0.25, 0)
        fb1 = lib.apgp_potrf_fallbacks()
        if dup is None:
            same = torch.equal(L0, L1) and torch.equal(z0, z1) and np.array_equal(o0, o1) and i0 == i1 and fb1 == fb0
        else:
            same = i0 == i1 and o0[4] == o1[4] and i0 > 0 and fb1 == fb0
        print("hybrid n=%5d D=%d%s  %s  (L diffs %d, z diffs %d, info %d/%d, fallbacks +%d)" %
              (n, D, "" if dup is None else " non-PD (dup at %d)" % dup, "identical" if same else "MISMATCH",
               int((L0 != L1).sum().item()), int((z0 != z1).sum().item()), i0, i1, fb1 - fb0), flush=True)
        bad += not same

# give-up path: mode 2 makes workgroup 0 abort at once; the call must come back with the multi-launch result
n, D = 1152, 8
rs = np.random.RandomState(3)
X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
ks = kern(D)
L1, z1, o1, i1 = nll(X_d, y_d, n, ks, 0.0, 1)
fb0 = lib.apgp_potrf_fallbacks()
t0 = time.perf_counter()
L2, z2, o2, i2 = nll(X_d, y_d, n, ks, 0.0, 2)
dt = time.perf_counter() - t0
fb1 = lib.apgp_potrf_fallbacks()
ok = torch.equal(L2, L1) and torch.equal(z2, z1) and np.array_equal(o2, o1) and fb1 == fb0 + 1
print("give-up + fallback: identical=%s fallbacks +%d  (%.1f ms)" % (ok, fb1 - fb0, dt * 1e3), flush=True)
bad += not ok

# stress: the same evaluation over and over (call-unique tags; no state may leak from call to call)
L0, z0, o0, i0 = nll(X_d, y_d, n, ks, 0.0, 0)
nbad = 0
for it in range(50 if quick else 300):
    La, za, oa, ia = nll(X_d, y_d, n, ks, 0.0, 0)
    nbad += not (torch.equal(La, L0) and torch.equal(za, z0) and np.array_equal(oa, o0))
print("stress n=%d: %d deviating calls, fallbacks so far %d" % (n, nbad, lib.apgp_potrf_fallbacks()), flush=True)
bad += nbad

for n in (512, 1152, 2048, 2560, 3072, 3712, 4096, 6144):
    if quick and n > 1152:
        break
    rs = np.random.RandomState(n)
    X = rs.uniform(-5, 5, size=(n, 8)); y = rs.normal(size=n)
    X_d = torch.from_numpy(X).to(dev); y_d = torch.from_numpy(y).to(dev)
    ks = kern(8)
    t1 = timeit(X_d, y_d, n, ks, 0.0, 1, 20)
    t0 = timeit(X_d, y_d, n, ks, 0.0, 3, 20)
    td = timeit(X_d, y_d, n, ks, 0.0, 0, 20)
    print("apgp_nll_eval n=%4d: multi-launch %.3f ms, persistent %.3f ms, default %.3f ms" % (n, t1 * 1e3, t0 * 1e3, td * 1e3), flush=True)
lib.apgp_potrf_mode(0)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
