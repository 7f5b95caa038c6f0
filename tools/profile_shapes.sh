#!/bin/bash
# rocprofv3 kernel trace + SQ / fetch counters of the fused sweep at C2's and C5's sizes (tools/sweep_shapes.py --quick
# --inverse-only: N=1024 D=2 1e5 BAPE, N=1152 D=8 1e6 AGP, C3) -- separate passes, as in tools/profile_round.sh.
#   bash tools/profile_shapes.sh r03ao      (GPU box, from the repo root)
tag=${1:-r03x}
B="python3 tools/sweep_shapes.py --quick --inverse-only"
out=gpurun_out/prof_shapes_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o shapes -- $B > $out/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o shapes -- $B > $out/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mem -o shapes -- $B > $out/pmc_mem.log 2>&1; echo "pmc_mem rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT TCC_MISS --output-format csv -d $out/pmc_write -o shapes -- $B > $out/pmc_write.log 2>&1; echo "pmc_write rc=$?"
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
root = sys.argv[1]
res = collections.OrderedDict()
# group sweep2 dispatches by grid size signature -> shape (persistent launches: the three shapes run in order)
def rows(pat):
    f = glob.glob(os.path.join(root, pat, "**", "*counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
for pat in ("pmc_sq", "pmc_mem", "pmc_write"):
    per = collections.OrderedDict()
    for r in rows(pat):
        if "sweep2_kernel<" not in r["Kernel_Name"]:
            continue
        key = r["Kernel_Name"].split("(")[0].strip()
        per.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, cs in per.items():
        res.setdefault(k, {}).update({n: sum(v) / max(1, len(v)) * (1 if True else 1) for n, v in cs.items()})
        res[k]["dispatches_" + pat] = max(len(v) for v in cs.values())
tr = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    dur = collections.OrderedDict()
    for r in csv.DictReader(open(tr[0])):
        if "sweep2_kernel<" in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0].strip()
            dur.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for k, v in dur.items():
        res.setdefault(k, {})["trace_ms_per_dispatch_mean"] = sum(v) / len(v)
        res[k]["trace_ms_max"] = max(v)
        res[k]["trace_dispatches"] = len(v)
for k, c in res.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        c["mfma_busy_fraction_mean_over_dispatches"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_HIT" in c and "TCC_MISS" in c and c["TCC_HIT"] + c["TCC_MISS"] > 0:
        c["l2_hit_rate"] = c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"])
    if "FETCH_SIZE" in c:
        c["hbm_bytes_per_dispatch_mean"] = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0
json.dump(res, open(os.path.join(root, "pmc_shapes.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
