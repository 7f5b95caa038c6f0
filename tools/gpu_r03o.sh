#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ship.so
for rep in 1 2; do
for v in ship half; do
    if [ $v = ship ]; then cp /tmp/ship.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "== $v rep $rep" >> gpurun_out/ab_r03o.txt
    timeout 600 python tools/sweep_shapes.py --quick 2>&1 | grep -E "N=.*inverse" >> gpurun_out/ab_r03o.txt
done
done
cp tools/tmp/libhalf.so $C/libapgp.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeded or deterministic or multi_row" 2>&1 | tail -2
cp /tmp/ship.so $C/libapgp.so
cut -c1-105 gpurun_out/ab_r03o.txt
timeout 1100 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5_run_loop" --durations=3 2>&1 | tail -8
