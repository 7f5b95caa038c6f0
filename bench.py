#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""
bench.py -- GP-predict + acquisition candidates/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic candidates:
the fused sweep (k* generation -> mu -> L^-1 k* contraction on the f64 matrix
cores -> variance -> AGP utility -> arg-min), the final arg-min reduction, and
(N > 1) the all-gather of the per-rank winners -- from "candidates resident in
HBM" to "(best_idx, best_u) on the host".  The fit (Gram + Cholesky + L^-1) is
outside the timed region and reported separately.

Workload at N=1: BASELINE.json configs[2] ("C3"): synthetic D=8 log-likelihood,
N_train=4096, 1e6 candidates, AGP utility, box prior [-5,5]^8, fp64.  For N>1
each rank sweeps its own 1e6-candidate shard (weak scaling, configs[3] shape).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 bench.py --gpus 8 ...

Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_TFLOPS = 78.6   # AMD public MI355X FP64 matrix = FP64 vector peak


def synthetic_c3(n_train, ndim):
    """X seed 0 ~ U[-5,5]^D, y = -rosen(X)/100."""
    from scipy.optimize import rosen
    rs = np.random.RandomState(0)
    X = rs.uniform(-5.0, 5.0, size=(n_train, ndim))
    y = np.array([-rosen(x) / 100.0 for x in X])
    return X, y


def profiled_traffic(n, d, m):
    """HBM-side bytes per sweep launch from the committed rocprofv3 PMC passes
    (profiles/r01f_pmc_sweep.json: FETCH_SIZE x2 for the gfx950 wide-stream
    correction + WRITE_SIZE, per MI355X_MICROARCH.md), only for the exact workload
    that was profiled; None otherwise (PMC counters are not collected inline)."""
    path = os.path.join(ROOT, "profiles", "r01f_pmc_sweep.json")
    if (n, d, m) != (4096, 8, 1000000) or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["derived"]["hbm_traffic_bytes_per_launch"])
    except Exception:
        return None


def f_var(n, d):
    """Algorithmic flops per candidate, SURVEY.md section 8(d)."""
    return float(n) * n + float(n) * (3 * d + 4)


def cpu_baseline(X, y, metric, ndim, seconds=12.0):
    """Reference-library batched path on the oracle:
    predict(y, T_chunk, return_var=True) on 4096-candidate chunks + vectorised
    AGP utility + arg-min, all host cores for BLAS.  george itself is not
    installable here, so this runs the NumPy/SciPy restatement (same LAPACK
    calls george's BasicSolver makes): kind = "port"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import george_oracle as go
    k = go.ExpSquaredKernel(np.full(ndim, metric), ndim=ndim)
    gp = go.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    t0 = time.time()
    gp.compute(X)
    gp._compute_alpha(y, True)
    fit_s = time.time() - t0
    rs = np.random.RandomState(1)
    done = 0
    best = (np.inf, -1)
    t0 = time.time()
    while time.time() - t0 < seconds:
        T = rs.uniform(-5.0, 5.0, size=(4096, ndim))
        mu, var = gp.predict(y, T, return_var=True)
        with np.errstate(all="ignore"):
            u = -(mu + 0.5 * np.log(2.0 * np.pi * np.e * var))
        i = int(np.nanargmin(u))
        if u[i] < best[0]:
            best = (float(u[i]), done + i)
        done += len(T)
    dt = time.time() - t0
    # (i) reference-faithful scalar path: one candidate per predict call, as
    # utility.minimizeObjective drives it (utility.py:131) -- small subsample
    t1 = time.time()
    ns = 0
    while time.time() - t1 < max(2.0, seconds / 4.0):
        t = rs.uniform(-5.0, 5.0, size=(1, ndim))
        mu, var = gp.predict(y, t, return_var=True)
        with np.errstate(all="ignore"):
            _ = -(mu + 0.5 * np.log(2.0 * np.pi * np.e * var))
        ns += 1
    ds = time.time() - t1
    return {"value": done / dt, "unit": "candidates/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d candidates in %.1f s (4096-candidate chunks, N_train=%d, D=%d, "
                      "oracle predict+utility+argmin; fit %.2f s excluded)" % (done, dt, len(y), ndim, fit_s),
            "scalar_path_value": ns / ds,
            "scalar_path_sample": "%d single-candidate predict+utility calls in %.1f s "
                                  "(what the reference's Nelder-Mead search evaluates)" % (ns, ds)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-train", type=int, default=4096)
    ap.add_argument("--ndim", type=int, default=8)
    ap.add_argument("--candidates", type=int, default=1000000, help="candidates per GPU per step")
    ap.add_argument("--metric", type=float, default=8.0)
    ap.add_argument("--utility", default="agp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("WARNING: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # torch.distributed.run
    if launched:
        dist.init_process_group(backend="nccl", device_id=dev)

    from approxposterior_amd import gp as agp
    from approxposterior_amd import dist as adist

    N, D, M = args.n_train, args.ndim, args.candidates
    X, y = synthetic_c3(N, D)
    kernel = agp.ExpSquaredKernel(np.full(D, args.metric), ndim=D)
    gp = agp.GP(kernel=kernel, fit_mean=True, mean=np.median(y), white_noise=-12,
                fit_white_noise=False, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    gp.compute(X)
    gp._ensure_xs(y)
    gp._ensure_linv()
    torch.cuda.synchronize()
    fit_ms = (time.time() - t0) * 1e3

    # candidates: resident in HBM before the timed region; rank r owns global rows
    # [r*M, (r+1)*M) (seed 1 + rank so shards differ)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    T = (torch.rand((M, D), dtype=torch.float64, device=dev, generator=gen) * 10.0 - 5.0).contiguous()
    bounds = [(-5.0, 5.0)] * D
    offset = rank * M

    def step():
        return adist.sharded_acquire(
            lambda off: gp.acquire(y, T, args.utility, bounds=bounds, idx_offset=off), offset)

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events around the sweep launch, recorded by GP._sweep on the stream the
    # kernel is launched on (torch's current stream)
    gp.kernel_events = kernel_ms = []
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        best = step()
    barrier()
    elapsed = time.time() - t0
    if launched:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kms = [a.elapsed_time(b) for a, b in kernel_ms]

    if rank == 0:
        total = float(M) * world * args.steps
        value = total / elapsed
        k_avg_ms = float(np.mean(kms))
        achieved = f_var(N, D) * M / (k_avg_ms * 1e-3) / 1e12
        out = {
            "metric": "GP-predict+acquisition candidates/sec (N_train, D fixed)",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: synthetic D=%d log-likelihood (-rosen/100), N_train=%d, "
                                   "%d candidates per GPU, %s utility, box prior [-5,5]^D, "
                                   "ExpSquaredKernel metric %.1f, white_noise -12"
                                   % (D, N, M, args.utility.upper(), args.metric),
                       "n_train": N, "ndim": D, "candidates_per_gpu": M,
                       "sharding": "candidates split by rank, one 16 B/rank all-gather",
                       "fit_ms_excluded": fit_ms},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F64_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F64_TFLOPS,
                         "traffic": profiled_traffic(N, D, M),
                         "kernel": "sweep2_kernel<%d, false>" % (2 if D <= 2 else 4 if D <= 4 else 8 if D <= 8 else 16),
                         "kernel_ms": k_avg_ms,
                         "algorithmic_flops_per_candidate": f_var(N, D)},
            "best": {"index": int(best[0]), "u": float(best[1])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, y, args.metric, D, args.cpu_seconds)
        print(json.dumps(out))
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
