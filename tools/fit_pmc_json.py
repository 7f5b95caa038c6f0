"""Condense tools/profile_fit.sh's rocprofv3 output: <dir>/kernel_stats.csv (the --stats summary) and <dir>/fit_pmc.json --
counters per (kernel, grid size) averaged over its dispatches, with the MFMA-busy fraction of the whole chip
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)) and the LDS bank-conflict share of LDS-active cycles."""
import csv, glob, json, os, shutil, sys
root = sys.argv[1]
stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, "kernel_stats.csv"))
acc = {}
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not any(s in name for s in ("potrf", "gram_kernel", "gram_batch", "nll_small")):
            continue
        key = "%s grid %s" % (name, r.get("Grid_Size", "?"))
        d = per.setdefault((key, r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (key, _), d in per.items():
        a = acc.setdefault(key, {})
        for c, v in d.items():
            s = a.setdefault(c, [0.0, 0])
            s[0] += v; s[1] += 1
out = {}
for key, a in sorted(acc.items()):
    d = {c: s[0] / s[1] for c, s in a.items()}
    d["dispatches"] = max(s[1] for s in a.values())
    if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_fraction_of_chip"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    out[key] = d
json.dump(out, open(os.path.join(root, "fit_pmc.json"), "w"), indent=1)
for key, d in out.items():
    if "persist" in key:
        print(key, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if "fraction" in k or k in ("dispatches", "GRBM_GUI_ACTIVE")})
