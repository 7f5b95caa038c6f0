#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_endtoend.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tests/gpu_fit_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/fit_timing_r03ai.txt
cat gpurun_out/fit_timing_r03ai.txt | cut -c1-260
timeout 300 python tools/rand_linalg_check.py 2>&1 | tail -4
