#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fit_trace -o fit -- python3 tools/fit_trace.py > gpurun_out/fit_trace.log 2>&1
python3 tools/fit_trace.py --summarise gpurun_out/fit_trace | tee gpurun_out/fit_trace_summary.txt
