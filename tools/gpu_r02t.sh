#!/bin/bash
./tools/tmp/panel_phases 2>&1 | tail -2 | tee gpurun_out/panel_phases_r02t.txt
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) | tee gpurun_out/pytest_r02t.txt
timeout 300 python tests/gpu_fit_timing.py 2>&1 | grep "N= 4096\|N= 1152\|N=  256" | tee gpurun_out/fit_timing_r02t.txt
