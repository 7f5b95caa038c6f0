#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_round.sh r02b 2>&1 | tail -12
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fit_r02b -o fit -- python3 tools/fit_prof.py > $O/prof_fit_r02b.log 2>&1; echo "fit trace rc=$?"
timeout 600 python bench.py > $O/bench_r02b.json 2> $O/bench_r02b.err; tail -c 900 $O/bench_r02b.json
echo; echo "== launched, 1 rank, C4 share"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --total-candidates 1250000 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700
