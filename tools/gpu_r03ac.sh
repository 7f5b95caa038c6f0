#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
O=gpurun_out/ab_part6.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship part6; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== bits $v" >> $O
    timeout 300 python tools/ab_bits.py 2>&1 | grep -E "N=|Error|error" >> $O
done
for v in ship part part6 ship part part6; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --partial --inverse-only 2>&1 | grep -E "N=" >> $O
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-105 $O
