#!/bin/bash
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ab_keep.so
cp tools/tmp/libwin4.so $C/libapgp.so
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/pytest_r02k.txt
for rep in 1 2 3; do
  for v in win4 win8 win1 ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee gpurun_out/ab_r02k.txt
for v in prof prof_win4; do
cp tools/tmp/lib$v.so $C/libapgp.so
echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o "sweep2 profile.*\|kernel_ms[^,]*" | tail -2
done | tee gpurun_out/prof_r02k.txt
cp /tmp/ab_keep.so $C/libapgp.so
