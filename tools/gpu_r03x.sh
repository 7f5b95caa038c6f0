#!/bin/bash
# same-box A/B: shipped sweep vs the four-slot image ring with the image stream two tiles ahead
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
O=gpurun_out/ab_r03x.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship ring4 ship ring4; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --ring 2>&1 | grep -E "N=" >> $O
done
cp tools/tmp/libring4.so $C/libapgp.so
echo "== ring4 parity" >> $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 >> $O
cp /tmp/ship.so $C/libapgp.so
cut -c1-118 $O
