#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_endtoend.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tests/gpu_fit_timing.py 2>&1 | grep "_nll eval" | sed 's/.*| mean/mean/'
timeout 900 python tools/run_configs.py 2>&1 | grep -E "^C1|same MCMC|C5 MCMC|C5 point search"
