#!/bin/bash
C=approxposterior_amd/csrc
O=gpurun_out
mkdir -p $O
cp $C/libapgp.so /tmp/ab_keep.so
cp tools/tmp/libring4.so $C/libapgp.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5 ) | tee $O/pytest_r02c_ring4.txt
for rep in 1 2 3; do
  for v in ring4 ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee $O/ab_r02c.txt
for v in prof prof_ring4; do
  cp tools/tmp/lib$v.so $C/libapgp.so
  echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep "sweep2 profile" | tail -1
done | tee $O/prof_r02c.txt
cp /tmp/ab_keep.so $C/libapgp.so
