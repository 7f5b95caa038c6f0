#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
for v in ship sf; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> gpurun_out/ab_r03g.txt
    timeout 900 python tools/sweep_shapes.py --dsweep 2>&1 | grep -E "N=" >> gpurun_out/ab_r03g.txt
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-110 gpurun_out/ab_r03g.txt
