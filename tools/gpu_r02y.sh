#!/bin/bash
./tools/tmp/potrf_check
for i in 1 2 3; do ( timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_endtoend.py -m gpu -x -q -k "c2_as_written or optimizeGP or findNext or batch" 2>&1 | tail -1 ); done
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 ) | tee gpurun_out/pytest_r02y.txt
./tools/tmp/panel_phases 2>&1 | tail -2 | cut -c1-700
timeout 300 python tests/gpu_fit_timing.py 2>&1 | grep "N=   50\|N=  512\|N= 4096\|N= 1152" | tee gpurun_out/fit_timing_r02y.txt
bash tools/gpu_r02u.sh | grep potrf
