"""Kernel-by-kernel timeline of ONE apgp_nll_eval at a hybrid size (default N = 4096): run under
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hyb -o hyb -- python3 tools/hybrid_trace.py 4096
and then `python3 tools/hybrid_trace.py --report gpurun_out/hyb` prints the last evaluation's launches (start offset,
duration, gap to the previous launch's end)."""
import csv, ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last evaluation: from the last gram kernel on
    idx = max(i for i, r in enumerate(rows) if "gram_kernel" in r["Kernel_Name"])
    ev = rows[idx:]
    t0 = int(ev[0]["Start_Timestamp"])
    prev_end = t0
    print("launch                       grid      start(us)  dur(us)  gap(us)")
    for r in ev:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]
        print("%-28s %-9s %9.1f %8.1f %8.1f" % (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")), (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3))
        prev_end = en
    print("total %.1f us" % ((prev_end - t0) / 1e3))
    sys.exit(0)
import numpy as np, torch
from approxposterior_amd import _lib, gp as agp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load(); D = 8; rs = np.random.RandomState(n)
X = rs.uniform(-5, 5, size=(n, D)); y = rs.normal(size=n)
g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=0.0, white_noise=-12, fit_white_noise=False)
g._x = X; g._yerr2 = 0.0
ks = g._kernel_struct()
X_d = torch.from_numpy(X).cuda(); y_d = torch.from_numpy(y).cuda()
K = torch.zeros((n, n), dtype=torch.float64, device="cuda"); z = torch.empty(n, dtype=torch.float64, device="cuda")
info = torch.empty(1, dtype=torch.int32, device="cuda"); o5 = torch.empty(5, dtype=torch.float64, device="cuda"); o = np.empty(5)
for _ in range(6):
    lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), 0.0, K.data_ptr(), z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
torch.cuda.synchronize()
print("ok", o)
