#!/bin/bash
# round 3: structured feeder (ported r02 experiment) vs the shipped feeder at the mid-N shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
cp tools/tmp/libsf.so $C/libapgp.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for v in ship sf; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03f.txt
    timeout 600 python tools/sweep_shapes.py --extra 2>&1 | grep -E "N=" >> gpurun_out/ab_r03f.txt
done
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-110 gpurun_out/ab_r03f.txt
