#!/bin/bash
C=approxposterior_amd/csrc
O=gpurun_out
mkdir -p $O
cp $C/libapgp.so /tmp/ab_keep.so
for rep in 1 2 3; do
  for v in r3c1 r4c1 nobar2 ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee $O/ab_r02d.txt
for v in prof prof_r3c1 prof_r4c1 prof prof_r4c1; do
  cp tools/tmp/lib$v.so $C/libapgp.so
  echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o "sweep2 profile.*\|kernel_ms[^,]*" | tail -2
done | tee $O/prof_r02d.txt
cp tools/tmp/libr4c1.so $C/libapgp.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3 ) | tee $O/pytest_r02d_r4c1.txt
cp /tmp/ab_keep.so $C/libapgp.so
