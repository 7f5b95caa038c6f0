#!/bin/bash
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ab_keep.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3 ) | tee gpurun_out/pytest_r02r.txt
for rep in 1 2 3; do
  for v in old ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee gpurun_out/ab_r02r.txt
cp /tmp/ab_keep.so $C/libapgp.so
