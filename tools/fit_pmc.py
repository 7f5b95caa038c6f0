"""Summarise a rocprofv3 --pmc pass of tools/fit_trace.py: counters per kernel (last fit only for
the multi-launch kernels), with MFMA-busy / wait fractions of the SIMD cycles."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
per = {}
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not any(s in name for s in ("trtri_merge", "syrk", "potrf_update", "potrf_panel")):
        continue
    d = per.setdefault((name, int(r["Dispatch_Id"])), {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
best = {}
for (name, disp), d in per.items():
    # the largest dispatch of each kernel (by wave cycles)
    if name not in best or d.get("SQ_WAVE_CYCLES", 0) > best[name][1].get("SQ_WAVE_CYCLES", 0):
        best[name] = (disp, d)
for name, (disp, d) in best.items():
    busy = d.get("SQ_BUSY_CYCLES", 0)
    line = "%-22s dispatch %5d" % (name[:22], disp)
    for k in sorted(d):
        line += "  %s=%.3g" % (k.replace("SQ_", ""), d[k])
    print(line)
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("    mfma busy fraction of SIMD cycles: %.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)))
