#!/bin/bash
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ab_keep.so
for rep in 1 2 3; do
  for v in prio1 prio2 prio3 ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee gpurun_out/ab_r02q.txt
for v in prof prof_prio1 prof_prio2; do
cp tools/tmp/lib$v.so $C/libapgp.so
echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o "sweep2 profile.*\|kernel_ms[^,]*" | tail -2
done | tee gpurun_out/prof_r02q.txt
cp /tmp/ab_keep.so $C/libapgp.so
