#!/bin/bash
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest_r02j.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_round.sh r02a 2>&1 | tail -30
timeout 600 python bench.py > $O/bench_r02j.json 2> $O/bench_r02j.err; tail -c 600 $O/bench_r02j.json
