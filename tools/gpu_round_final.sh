#!/bin/bash
# round-2 closing measurements: fit-path kernel statistics (one fit + L^-1 + gradient at N = 4096,
# twice), fit timings, bench line
export TMPDIR=/tmp
rm -rf gpurun_out/fit_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_stats -o fit -- python3 tools/fit_trace.py > gpurun_out/fit_stats.log 2>&1
python3 tools/fit_trace.py --summarise gpurun_out/fit_stats | tee gpurun_out/fit_trace_summary_r02e.txt
timeout 300 python tests/gpu_fit_timing.py 2>&1 | tee gpurun_out/fit_timing_r02e.txt | grep "N= 4096\|N= 1152"
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r02e.json | cut -c1-1500
