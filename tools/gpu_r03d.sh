#!/bin/bash
# round 3, fourth GPU call: cycle stamps of the burst and the base sweep kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
for v in burstprof baseprof; do
    cp tools/tmp/lib$v.so $C/libapgp.so
    echo "== $v" >> gpurun_out/prof_r03d.txt
    timeout 600 python tools/sweep_shapes.py --quick 2>&1 | grep -E "N=|prof" | awk '!seen[$0]++' >> gpurun_out/prof_r03d.txt
done
cp /tmp/ship.so $C/libapgp.so
cut -c1-400 gpurun_out/prof_r03d.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or fixture_fit" 2>&1 | tail -3
