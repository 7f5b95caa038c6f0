"""Randomised stress of the persistent triangular solve against the launch-per-256-rows path: random sizes, leading dimensions,
orientations, shifts, aliasing; x and x.x compared bit for bit.   Usage (GPU box): python tools/stress_trsv.py [cases] [max_n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rs = np.random.RandomState(7)
bad = 0
t0 = time.time()
gen = torch.Generator(device="cuda"); gen.manual_seed(11)
for c in range(cases):
    n = int(rs.randint(256, max_n))
    ld = n + int(rs.randint(0, 3)) * int(rs.randint(1, 9))
    trans = int(rs.randint(0, 2)); alias = bool(rs.randint(0, 2)); shift = float(rs.normal())
    L = torch.zeros((n, ld), dtype=torch.float64, device=dev)
    L[:, :n] = torch.tril(torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen) * 0.03)
    L[:, :n] += torch.diag(1.0 + torch.rand(n, dtype=torch.float64, device=dev, generator=gen))
    b = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    out = []
    for mode in (1, 0, 0):
        lib.apgp_trsv_mode(mode)
        x = b.clone() if alias else torch.empty(n, dtype=torch.float64, device=dev)
        ss = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
        rc = lib.apgp_trsv(L.data_ptr(), n, ld, (x if alias else b).data_ptr(), shift, trans, x.data_ptr(), ss.data_ptr(), None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        out.append((x, float(ss.item())))
    lib.apgp_trsv_mode(0)
    for o in out[1:]:
        if not (torch.equal(o[0], out[0][0]) and o[1] == out[0][1]):
            bad += 1
            print("MISMATCH n=%d ld=%d trans=%d alias=%s" % (n, ld, trans, alias), flush=True)
    if (c + 1) % 100 == 0:
        print("%d cases, %d mismatches, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
