#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r03m.txt 2>&1
tail -3 gpurun_out/pytest_r03m.txt
timeout 600 python tools/nll_latency.py > gpurun_out/nll_latency_r03m.txt 2>&1
grep "N=" gpurun_out/nll_latency_r03m.txt
head -40 gpurun_out/nll_latency_r03m.txt | tail -32
