#!/bin/bash
# round 3, first GPU call: new parity tests, sweep shapes (both variance forms), full suite, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "substitution or ladder or illcond" > gpurun_out/pytest_r03a_new.txt 2>&1
tail -5 gpurun_out/pytest_r03a_new.txt
timeout 600 python tests/gpu_illcond_report.py > gpurun_out/illcond_r03a.txt 2>&1
timeout 900 python tools/sweep_shapes.py > gpurun_out/shapes_r03a.txt 2>&1
tail -12 gpurun_out/shapes_r03a.txt
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/pytest_r03a.txt 2>&1
tail -25 gpurun_out/pytest_r03a.txt
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err
cat gpurun_out/bench_r03a.json
