#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_r03k.txt 2>&1 || { echo "SMOKE FAILED"; tail -20 gpurun_out/smoke_r03k.txt; exit 1; }
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
cp $C/libapgp.so /tmp/ship.so
for rep in 1 2; do
for v in ship base; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03k.txt
    timeout 600 python tools/sweep_shapes.py 2>&1 | grep -E "N=" >> gpurun_out/ab_r03k.txt
done
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-105 gpurun_out/ab_r03k.txt
