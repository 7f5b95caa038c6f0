"""MI355X: ``bench.py --gpus N`` starts its own ranks (missing 1 / weak 4).

``python bench.py --gpus 2 --backend gloo --share-device ...`` -- the shape of the one command the driver runs,
without a launcher -- must start two fresh rank processes (both on ``cuda:0``, host collectives), run the whole
multi-rank bench (shard bounds, the sweep's device record, the all-gather, ``ranks_seen``, ``best_check.vs_ranks``) and
report ``n_gpus == 2``; its winner is the winner of the single-process sweep over the same global candidate matrix.
A launcher whose world size differs from ``--gpus`` is an error, never a silent one-rank line.
Reference: /root/reference/approxposterior/approx.py:396-424 (run), :664-672 (point search)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--n-train", "700", "--candidates", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]

pytestmark = [pytest.mark.timeout(900)]


def _run(argv, env=None):
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    proc = subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          env=e, timeout=800)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    return proc, lines


@pytest.mark.gpu
def test_bench_gpus2_starts_two_ranks_on_one_gpu():
    proc, lines = _run(["--gpus", "2", "--backend", "gloo", "--share-device"] + SMALL)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert len(lines) == 1, proc.stdout            # ONE JSON line, from rank 0
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert two["config"]["candidates_total"] == 40000 and two["config"]["candidates_per_gpu"] == 20000
    assert two["config"]["backend"] == "gloo" and "NOT a scaling" in two["config"]["rehearsal"]
    assert two["best_checked"] and two["best_check"]["vs_ranks"] is True
    assert two["scaling"] == "weak" and two["steps"] == 2

    # the same global matrix (RandomState(1), 40000 x 8) swept by ONE process
    proc1, lines1 = _run(["--gpus", "1", "--total-candidates", "40000"] + SMALL)
    assert proc1.returncode == 0, proc1.stderr[-3000:]
    one = json.loads(lines1[0])
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1 and one["config"]["backend"] is None
    assert one["best"] == two["best"]              # index and utility, every bit


def test_bench_refuses_a_world_that_is_not_gpus():
    """Fewer devices than ranks without --share-device -> exit 3; a launcher of another size -> exit 4.
    (Both refusals come before any GPU call: this test also runs in the CPU suite.)"""
    proc, lines = _run(["--gpus", "64"] + SMALL)
    assert proc.returncode == 3 and not lines and "refusing" in proc.stderr
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=env, timeout=300)
    assert proc.returncode == 4 and '{"metric"' not in proc.stdout and "refusing" in proc.stderr
