#!/bin/bash
# round 3: closing profiles at the final sweep sources (hash-guarded traffic figure), bench lines, config timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile_round.sh r03u > gpurun_out/profile_round_r03u.log 2>&1
tail -9 gpurun_out/profile_round_r03u.log
timeout 600 python bench.py > gpurun_out/bench_r03u.json 2> gpurun_out/bench_r03u.err
cut -c1-2600 gpurun_out/bench_r03u.json
timeout 900 python tools/run_configs.py > gpurun_out/run_configs_r03u.txt 2>&1
grep -E "^C1|same MCMC|^C5" gpurun_out/run_configs_r03u.txt
timeout 300 python tools/predict_latency.py 2>&1 | grep "N=" > gpurun_out/predict_latency_r03u.txt
cat gpurun_out/predict_latency_r03u.txt
