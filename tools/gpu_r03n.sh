#!/bin/bash
# round 3: closing measurements, part 1 (long): C5 exactly as written, C3 profiles of both forms, fit kernel stats
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh r03n > gpurun_out/profile_round_r03n.log 2>&1
tail -9 gpurun_out/profile_round_r03n.log
rm -rf gpurun_out/prof_r03n_solve
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03n_solve -o sweep -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --variance solve > gpurun_out/prof_r03n_solve.log 2>&1
cp $(find gpurun_out/prof_r03n_solve -name "*kernel_stats.csv" | head -1) gpurun_out/r03n_solve_kernel_stats.csv
head -3 gpurun_out/r03n_solve_kernel_stats.csv
rm -rf gpurun_out/fit_stats_r03n
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_stats_r03n -o fit -- python3 tools/fit_trace.py > gpurun_out/fit_stats_r03n.log 2>&1
cp $(find gpurun_out/fit_stats_r03n -name "*kernel_stats.csv" | head -1) gpurun_out/r03n_fit_kernel_stats.csv
python3 tools/fit_trace.py --summarise gpurun_out/fit_stats_r03n | tee gpurun_out/fit_trace_summary_r03n.txt | tail -12
timeout 300 python tests/gpu_fit_timing.py > gpurun_out/fit_timing_r03n.txt 2>&1
grep "_nll eval" gpurun_out/fit_timing_r03n.txt
timeout 600 python tools/run_configs.py > gpurun_out/run_configs_r03n.txt 2>&1
cat gpurun_out/run_configs_r03n.txt
timeout 2400 python tools/run_configs.py --c5-as-written > gpurun_out/c5_as_written_r03n.txt 2>&1
cat gpurun_out/c5_as_written_r03n.txt
