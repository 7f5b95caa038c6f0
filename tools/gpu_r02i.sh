#!/bin/bash
C=approxposterior_amd/csrc
cp $C/libapgp.so /tmp/ab_keep.so
cp tools/tmp/libfprof.so $C/libapgp.so
for r in 1 2; do timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o "sweep2 profile.*\|feeder profile.*\|kernel_ms[^,]*" | tail -5; done | tee gpurun_out/prof_r02i.txt
cp /tmp/ab_keep.so $C/libapgp.so
