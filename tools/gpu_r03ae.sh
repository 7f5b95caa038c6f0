#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
O=gpurun_out/ab_part6_solve.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship part6 ship part6; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --partial 2>&1 | grep -E "solve" >> $O
    timeout 600 python tools/sweep_shapes.py --dsweep 2>&1 | grep -E "solve" >> $O
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-100 $O
