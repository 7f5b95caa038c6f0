#!/bin/bash
# same-box A/B: shipped sweep vs predicated body without inactive-pair reads (+ partial image requests, structured feeder)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
V=${VARIANT:-pred}
O=gpurun_out/ab_${V}.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship $V; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== bits $v" >> $O
    timeout 300 python tools/ab_bits.py 2>&1 | grep -E "N=|Error|error" >> $O
done
for v in ship $V ship $V; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --ring 2>&1 | grep -E "N=" >> $O
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-125 $O
