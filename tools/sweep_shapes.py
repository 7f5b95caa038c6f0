#!/usr/bin/env python
"""Sweep throughput away from C3 (developer script, GPU box): HIP-event time of the fused sweep
for the BASELINE.json shapes C2 (N=1024, D=2, 1e5, BAPE), C5 (N=512 / 1152, D=8, 1e6, AGP) and C3,
inverse and substitution form, with the roofline fraction on F_var = N^2 + N(3D+4) flops."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from approxposterior_amd import gp as agp
from bench import synthetic_c3, f_var, PEAK_F64_TFLOPS

shapes = [(1024, 2, 100000, "bape"), (512, 8, 1000000, "agp"), (1152, 8, 1000000, "agp"),
          (2048, 8, 1000000, "agp"), (4096, 8, 1000000, "agp")]
if "--quick" in sys.argv:
    shapes = [(1024, 2, 100000, "bape"), (1152, 8, 1000000, "agp"), (4096, 8, 1000000, "agp")]
if "--extra" in sys.argv:
    shapes = [(1024, 2, 100000, "bape"), (1024, 8, 100000, "bape"), (1024, 2, 1000000, "bape"), (1152, 2, 1000000, "agp"),
              (1152, 8, 1000000, "agp"), (4096, 8, 1000000, "agp")]
if "--dsweep" in sys.argv:
    shapes = [(n, d, 1000000, "agp") for d in (2, 4, 8, 16) for n in (1024, 1152, 2048)]
if "--ring" in sys.argv:
    shapes = [(1024, 2, 100000, "bape"), (1024, 2, 1000000, "agp"), (1152, 8, 1000000, "agp"), (1152, 2, 1000000, "agp"),
              (1100, 8, 1000000, "agp"), (1280, 8, 1000000, "agp"), (2048, 8, 1000000, "agp"), (2304, 16, 1000000, "agp"),
              (4096, 8, 1000000, "agp")]
if "--elim" in sys.argv:
    shapes = [(1152, 8, 1000000, "agp"), (1152, 2, 1000000, "agp"), (1024, 2, 1000000, "agp"), (1280, 8, 1000000, "agp"),
              (4096, 8, 1000000, "agp")]
if "--partial" in sys.argv:
    shapes = [(1100, 8, 1000000, "agp"), (1152, 8, 1000000, "agp"), (1200, 8, 1000000, "agp"), (1216, 2, 1000000, "agp"),
              (1250, 8, 1000000, "agp"), (600, 8, 1000000, "agp"), (2100, 4, 1000000, "agp"), (4200, 8, 500000, "agp"),
              (4096, 8, 1000000, "agp")]
if "--big" in sys.argv:
    shapes = [(8192, 8, 262144, "agp"), (6144, 8, 524288, "agp"), (4096, 8, 1000000, "agp")]
reps = 5
out = []
for n, d, m, kind in shapes:
    X, y = synthetic_c3(n, d)
    T = torch.from_numpy(np.random.RandomState(1).uniform(-5, 5, size=(m, d))).cuda()
    for mode in (("inverse",) if "--inverse-only" in sys.argv else ("inverse", "solve")):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
                   white_noise=-12, fit_white_noise=False)
        g.variance_mode = mode
        g.compute(X)
        b0 = g.acquire(y, T, kind, bounds=[(-5, 5)] * d)
        g.kernel_events = ev = []
        for _ in range(reps):
            b = g.acquire(y, T, kind, bounds=[(-5, 5)] * d)
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b_) for a, b_ in ev]))
        tf = f_var(n, d) * m / (ms * 1e-3) / 1e12
        rec = dict(n=n, d=d, m=m, kind=kind, mode=mode, ms=ms, cand_per_s=m / (ms * 1e-3), tflops=tf,
                   frac=tf / PEAK_F64_TFLOPS, best=[int(b[0]), float(b[1])])
        out.append(rec)
        print("N=%5d D=%d M=%7d %-4s %-7s  %9.3f ms  %.3e cand/s  %6.2f TF  frac %.3f  best %s" % (
            n, d, m, kind, mode, ms, rec["cand_per_s"], tf, rec["frac"], rec["best"]), flush=True)
print(json.dumps(out))
