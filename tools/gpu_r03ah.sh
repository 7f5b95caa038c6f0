#!/bin/bash
# round 3: profile passes + full validation at the final sweep source (compile-time partial-row-block bodies)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_round.sh r03ah > gpurun_out/profile_r03ah.log 2>&1
tail -5 gpurun_out/profile_r03ah.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r03ah.txt 2>&1
tail -2 gpurun_out/pytest_r03ah.txt
