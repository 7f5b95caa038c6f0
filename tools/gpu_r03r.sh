#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_endtoend.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/nll_latency.py 2>&1 | grep "N=" | head -5
timeout 300 python tests/gpu_fit_timing.py 2>&1 | grep "_nll eval"
