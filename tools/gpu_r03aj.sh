#!/bin/bash
# round 3: closing validation at the final commit: smoke, full GPU suite, bench line (both forms), one torch.distributed.run rank
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r03aj.txt 2>&1
tail -2 gpurun_out/pytest_r03aj.txt
timeout 600 python bench.py > gpurun_out/bench_r03aj.json 2> gpurun_out/bench_r03aj.err
cat gpurun_out/bench_r03aj.json
timeout 600 python bench.py --variance solve --no-fit-leg --no-cpu-baseline > gpurun_out/bench_r03aj_solve.json 2>> gpurun_out/bench_r03aj.err
cut -c1-700 gpurun_out/bench_r03aj_solve.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r03aj_dist1.json 2>> gpurun_out/bench_r03aj.err
cut -c1-400 gpurun_out/bench_r03aj_dist1.json
tail -3 gpurun_out/bench_r03aj.err
