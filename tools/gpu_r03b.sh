#!/bin/bash
# round 3, second GPU call: fused small-N _nll, per-kernel trace at the mid-N shapes, C3 profiles (both forms)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r03b.txt 2>&1
tail -4 gpurun_out/pytest_r03b.txt
timeout 600 python tests/gpu_fit_timing.py > gpurun_out/fit_timing_r03b.txt 2>&1
grep "_nll eval" gpurun_out/fit_timing_r03b.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b_shapes -o shapes -- python3 tools/sweep_shapes.py --quick > gpurun_out/prof_r03b_shapes.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_r03b_shapes/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print the last acquire call of each (shape, mode): sequences ending in argmin_final_kernel
seq = []
out = open("gpurun_out/shapes_trace_r03b.txt", "w")
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("sweep2_kernel", "sweep_finish", "argmin_final")):
        seq.append(r)
        if "argmin_final" in n:
            t0 = int(seq[0]["Start_Timestamp"])
            line = " | ".join("%s grid %s: +%.1f..%.1f us" % (s["Kernel_Name"][:40], s["Grid_Size"] if "Grid_Size" in s else s.get("Workgroup_Size", "?"),
                              (int(s["Start_Timestamp"]) - t0) / 1e3, (int(s["End_Timestamp"]) - t0) / 1e3) for s in seq)
            out.write(line + "\n")
            seq = []
out.close()
PY
tail -30 gpurun_out/shapes_trace_r03b.txt
bash tools/profile_round.sh r03a > gpurun_out/profile_round_r03a.log 2>&1
tail -12 gpurun_out/profile_round_r03a.log
timeout 600 python bench.py --steps 5 --warmup 1 --variance solve --no-fit-leg > gpurun_out/bench_r03b_solve.json 2> gpurun_out/bench_r03b_solve.err
cat gpurun_out/bench_r03b_solve.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03b_solve -o sweep -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --variance solve > gpurun_out/prof_r03b_solve.log 2>&1
cp $(find gpurun_out/prof_r03b_solve -name "*kernel_stats.csv" | head -1) gpurun_out/r03b_solve_kernel_stats.csv
head -8 gpurun_out/r03b_solve_kernel_stats.csv
