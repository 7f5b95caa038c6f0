#!/bin/bash
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ab_keep.so
for v in prof_r3c0 prof_r3c1 prof_r3c0 prof_r3c1; do
  cp tools/tmp/lib$v.so $C/libapgp.so
  echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o "sweep2 profile.*\|kernel_ms[^,]*" | tail -4
done | tee gpurun_out/prof_r02e.txt
cp /tmp/ab_keep.so $C/libapgp.so
