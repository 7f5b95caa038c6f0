#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
O=gpurun_out/ab_part6s.txt
: > $O
cp $C/libapgp.so /tmp/ship.so
for v in ship part6s ship part6s; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v" >> $O
    timeout 600 python tools/sweep_shapes.py --partial 2>&1 | grep -E "N=" >> $O
    timeout 600 python tools/sweep_shapes.py --quick 2>&1 | grep -E "N= 1024" >> $O
done
cp tools/tmp/libpart6s.so $C/libapgp.so
echo "== part6s parity + fuzz" >> $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 >> $O
timeout 900 python tools/fuzz_sweep.py 2>&1 | tail -3 >> $O
cp /tmp/ship.so $C/libapgp.so
cut -c1-118 $O
