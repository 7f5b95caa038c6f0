import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from approxposterior_amd import gp as agp
from test_gpu_parity import build
for name in ("rosen2d_n50_amp_cond1e8", "rosen2d_n50_amp_cond1e11", "rosen2d_n50_amp_cond1e13", "rosen2d_n50_amp_opt_illcond"):
  g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % name))
  print("==", name, "true cond %.3e" % g["cond"])
  for mode in ("solve", "inverse", ""):
      gp = build(agp, g)
      gp.variance_mode = mode or None
      mu, var = gp.predict(g["y"], g["cands"], return_var=True)
      vt, mt = g["var_truth"], g["mu_truth"]
      print("mode=%-8s cond_est %.3g | var rel err vs truth: mine median %.3g max %.3g | oracle median %.3g max %.3g | mu rel err mine max %.3g oracle max %.3g" % (
          mode or "auto", gp.cond_estimate,
          np.median(np.abs(var - vt) / np.abs(vt)), np.max(np.abs(var - vt) / np.abs(vt)),
          np.median(np.abs(g["var"] - vt) / np.abs(vt)), np.max(np.abs(g["var"] - vt) / np.abs(vt)),
          np.max(np.abs(mu - mt) / np.abs(mt)), np.max(np.abs(g["mu"] - mt) / np.abs(mt))))
      print("   per-cand |mine-truth|/|oracle-truth| :", np.array2string(np.abs(var - vt) / np.maximum(np.abs(g["var"] - vt), 1e-300), precision=2, max_line_width=200))
