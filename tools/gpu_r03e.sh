#!/bin/bash
# round 3: what slows the substitution form's straight tiles (sc1 parked loads? the matrix role's vmcnt wait?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
for rep in 1 2; do
for v in ship sa sb sc; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03e.txt
    timeout 600 python tools/sweep_shapes.py --quick 2>&1 | grep -E "N=" >> gpurun_out/ab_r03e.txt
done
done
cp /tmp/ship.so $C/libapgp.so
cat gpurun_out/ab_r03e.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
