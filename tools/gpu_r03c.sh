#!/bin/bash
# round 3, third GPU call: burst-generation sweep: parity, same-box A/B against the r03b kernel, _nll latency split
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_r03c.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 gpurun_out/smoke_r03c.txt; exit 1; }
tail -1 gpurun_out/smoke_r03c.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deterministic or seeded or substitution" > gpurun_out/pytest_r03c_first.txt 2>&1 || { echo "FIRST TESTS FAILED"; tail -30 gpurun_out/pytest_r03c_first.txt; exit 1; }
tail -2 gpurun_out/pytest_r03c_first.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r03c.txt 2>&1
tail -6 gpurun_out/pytest_r03c.txt
cp $C/libapgp.so /tmp/ship.so
for rep in 1 2; do
  for v in ship base; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/libbase.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03c.txt
    timeout 600 python tools/sweep_shapes.py 2>&1 | grep "N=" >> gpurun_out/ab_r03c.txt
  done
done
cp /tmp/ship.so $C/libapgp.so
cat gpurun_out/ab_r03c.txt
timeout 600 python tools/nll_latency.py > gpurun_out/nll_latency_r03c.txt 2>&1
head -60 gpurun_out/nll_latency_r03c.txt
