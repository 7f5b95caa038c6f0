"""Developer probe: ONE fit + L^-1 + gradient at N (default 4096), for rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fit_trace -o fit -- python3 tools/fit_trace.py
then  python3 tools/fit_trace.py --summarise gpurun_out/fit_trace  prints the launches in order."""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = [r for r in rows if any(s in r["Kernel_Name"] for s in ("trtri", "syrk", "grad_", "potrf", "pack_linv", "gram"))]
    # the last fit: everything after the last gram kernel
    last = max(i for i, r in enumerate(keep) if "gram" in r["Kernel_Name"])
    t0 = int(keep[last]["Start_Timestamp"])
    agg = {}
    for r in keep[last:]:
        name = r["Kernel_Name"].split("(")[0]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "potrf" in name:
            a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += dur
        else:
            print("%9.1f us  +%8.1f  %-28s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur, name[:28], r.get("Grid_Size", "?")))
    for k, v in agg.items():
        print("%-28s %4d launches, %9.1f us total" % (k, v[0], v[1]))
    sys.exit(0)

import numpy as np, torch
from approxposterior_amd import gp as agp
from scipy.optimize import rosen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = 8
rs = np.random.RandomState(0)
X = rs.uniform(-5, 5, size=(N, D)); y = np.array([-rosen(x) / 100 for x in X])
k = agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
for rep in range(2):
    gp = agp.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(X); gp.log_likelihood(y); gp.grad_log_likelihood(y)
    torch.cuda.synchronize()
