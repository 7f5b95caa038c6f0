#!/bin/bash
# round 3: stress -- repeat-launch determinism of both sweep forms and of the mailbox _nll path, the suite three times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python tools/stress_sweep.py > gpurun_out/stress_r03p.txt 2>&1
tail -22 gpurun_out/stress_r03p.txt
for i in 1 2; do ( timeout 900 python -m pytest tests -m gpu -x -q -k "not c5_run_loop" 2>&1 | tail -1 ); done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
