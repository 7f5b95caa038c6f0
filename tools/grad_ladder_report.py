"""Gradient of the log-likelihood on the conditioning ladder: HIP (explicit inverse / solve route / automatic gate)
and the oracle against the 60-digit mpmath truth (tests/golden/*cond1e*.npz, *_opt_illcond.npz: grad_truth)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from approxposterior_amd import gp as agp
from test_gpu_parity import build
G = os.path.join(ROOT, "tests", "golden")
for name in ("rosen2d_n50_amp_cond1e8", "rosen2d_n50_amp_cond1e11", "rosen2d_n50_amp_cond1e13", "rosen2d_n50_amp_opt_illcond"):
    g = np.load(os.path.join(G, name + ".npz"))
    t = g["grad_truth"]
    print("%s: true cond %.2e; truth %s" % (name, float(g["cond"]), t))
    print("   oracle           abs err %s" % np.abs(g["grad"] - t))
    for mode in ("inverse", "solve", None):
        gp = build(agp, g)
        gp.variance_mode = mode
        gr = gp.grad_log_likelihood(g["y"])
        print("   HIP %-12s abs err %s   (estimate %.2e, trusted %s)" % (mode, np.abs(gr - t), gp.cond_estimate, gp._trust_inverse()))
