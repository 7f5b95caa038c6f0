#!/usr/bin/env python
"""Condense the rocprofv3 output of tools/profile_round.sh into
gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_sweep.json} (copy into profiles/)."""
import csv, glob, hashlib, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.join("gpurun_out", "prof_" + tag)
KERNEL = "sweep2_kernel"      # the two-role sweep

stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, "kernel_stats.csv"))

# one acquire call = the persistent launch + (short last round) the row-block-split launch of
# the same kernel + sweep_finish_kernel + argmin_final_kernel: counters are summed over a
# call's sweep_kernel dispatches and averaged over the calls (= argmin_final_kernel dispatches)
counters = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    tot, calls = {}, set()
    for row in csv.DictReader(open(f)):
        if "argmin_final_kernel" in row["Kernel_Name"]:
            calls.add(row["Dispatch_Id"])
        if KERNEL + "<" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for name, v in tot.items():
        counters[name] = {"launches": len(calls), "mean": v / max(len(calls), 1)}

trace = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)
sweep_ms = None
if trace:
    dur, calls = 0.0, 0
    for row in csv.DictReader(open(trace[0])):
        if KERNEL + "<" in row["Kernel_Name"] or "sweep_finish_kernel" in row["Kernel_Name"]:
            dur += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        if "argmin_final_kernel" in row["Kernel_Name"]:
            calls += 1
    sweep_ms = dur / max(calls, 1)

def c(name):
    return counters[name]["mean"] if name in counters else None

derived = {}
if sweep_ms is not None:
    derived["sweep_kernels_ms_per_call"] = sweep_ms   # rocprofv3 kernel trace: all sweep launches of one call
if c("FETCH_SIZE") is not None and c("WRITE_SIZE") is not None:
    # FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 under-counts 16 B/lane read
    # streams by 2x (MI355X_MICROARCH.md, HBM section)
    derived["hbm_traffic_bytes_per_launch"] = (2.0 * c("FETCH_SIZE") + c("WRITE_SIZE")) * 1024.0
    derived["fetch_correction"] = ("FETCH_SIZE x2 (gfx950 16B/lane streams, MI355X_MICROARCH.md HBM section); "
                                   "WRITE_SIZE as reported")
if c("TCC_HIT") is not None and c("TCC_MISS") is not None:
    derived["l2_hit_rate"] = c("TCC_HIT") / (c("TCC_HIT") + c("TCC_MISS"))
if c("SQ_VALU_MFMA_BUSY_CYCLES") is not None and c("GRBM_GUI_ACTIVE") is not None:
    # busy cycles summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs
    derived["shader_cycles_per_launch"] = c("GRBM_GUI_ACTIVE") / 8.0
    derived["mfma_busy_fraction"] = c("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * c("GRBM_GUI_ACTIVE") / 8.0)
# identity of the profiled kernel: bench.py only reports `traffic` from a summary whose hash
# matches the sources it runs (same function as bench.sweep_source_hash)
sha = hashlib.sha256()
for name in ("sweep.hip", "apgp_common.h"):
    sha.update(open(os.path.join("approxposterior_amd", "csrc", name), "rb").read())
out = {"kernel": KERNEL, "sweep_source_sha": sha.hexdigest()[:16], "workload": "bench.py default: N_train=4096, D=8, 1e6 candidates, AGP",
       "how": "tools/profile_round.sh %s (separate rocprofv3 --pmc passes, 1 warm-up + 3 timed launches each)" % tag,
       "counters": counters, "derived": derived}
json.dump(out, open(os.path.join(root, "pmc_sweep.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
