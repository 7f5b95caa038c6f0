#!/bin/bash
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 ) | tee gpurun_out/pytest_r02y.txt
timeout 300 python tests/gpu_fit_timing.py 2>&1 | grep "N=   50\|N=  512\|N= 4096\|N= 1152" | tee gpurun_out/fit_timing_r02y.txt
bash tools/gpu_r02u.sh
