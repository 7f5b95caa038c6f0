#!/bin/bash
# round-2 GPU call A: full GPU test-suite on the cleaned library, default bench line,
# same-box A/B of the S2_SPREAD variants, per-tile cycle counts of the timing builds.
C=approxposterior_amd/csrc
O=gpurun_out
mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -40 ) > $O/pytest_r02a.txt
tail -5 $O/pytest_r02a.txt
timeout 600 python bench.py > $O/bench_r02a.json 2> $O/bench_r02a.err; tail -c 1500 $O/bench_r02a.json
cp $C/libapgp.so /tmp/ab_keep.so
for rep in 1 2 3; do
  for v in spread1 spread2 ship; do
    if [ $v = ship ]; then cp /tmp/ab_keep.so $C/libapgp.so; else cp tools/tmp/lib$v.so $C/libapgp.so; fi
    echo "$v $(timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep -o 'kernel_ms[^,]*')"
  done
done | tee $O/ab_r02a.txt
for v in timing timing_sp2; do
  cp tools/tmp/lib$v.so $C/libapgp.so
  echo "== $v"; timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check 2>&1 | grep "sweep2 timing" | tail -2
done | tee $O/timing_r02a.txt
cp /tmp/ab_keep.so $C/libapgp.so
