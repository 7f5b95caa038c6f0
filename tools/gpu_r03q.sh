#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
cp tools/tmp/libmix8.so $C/libapgp.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "substitution or ladder or illcond" 2>&1 | tail -2
for rep in 1 2; do
for v in ship mix8; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03q.txt
    timeout 600 python tools/sweep_shapes.py 2>&1 | grep -E "N=.*solve|N= 4096" >> gpurun_out/ab_r03q.txt
done
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-105 gpurun_out/ab_r03q.txt
