#!/bin/bash
# round-2 GPU call B: new fit-side paths (GEMV alpha, lower-only Gram, sync-free append) through the
# GPU suite; elimination builds of the spread-read sweep; row-block cycle profile; fit timing.
C=approxposterior_amd/csrc
O=gpurun_out
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_r02b.txt
tail -3 $O/pytest_r02b.txt
cp $C/libapgp.so /tmp/ab_keep.so
for rep in 1 2; do
  for v in nospread noa nob nogen noexp ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee $O/ab_r02b.txt
for v in prof prof_nogen; do
  cp tools/tmp/lib$v.so $C/libapgp.so
  echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep "sweep2 profile" | tail -1
done | tee $O/prof_r02b.txt
cp /tmp/ab_keep.so $C/libapgp.so
timeout 300 python tests/gpu_fit_timing.py 2>&1 | tail -8 | tee $O/fit_timing_r02b.txt
