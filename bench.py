#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""
bench.py -- GP-predict + acquisition candidates/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic candidates:
the fused sweep (k* generation -> mu -> L^-1 k* contraction on the f64 matrix
cores -> variance -> AGP utility -> arg-min), the final arg-min reduction, and
(N > 1) the all-gather of the per-rank winners -- from "candidates resident in
HBM" to "(best_idx, best_u) on the host".  The fit (Gram + Cholesky + L^-1) and
the H2D copy of the candidates are outside the timed region and reported
separately (``config.fit_ms_warm``, ``config.h2d_candidates_ms``).

Workload at N=1: BASELINE.json configs[2] ("C3"): synthetic D=8 log-likelihood,
N_train=4096, 1e6 candidates, AGP utility, box prior [-5,5]^8, fp64.  The candidate
matrix is ONE NumPy ``RandomState(1)`` draw of (total candidates) x D (SURVEY.md
section 8d); rank r owns its rows [r M/world, (r+1) M/world), so the printed ``best`` can be
re-derived by anybody.  For N>1:
  default                      weak scaling: 1e6 candidates per rank (total = N x 1e6);
  --total-candidates 10000000  configs[3] ("C4") as written: 1e7 candidates split over the
                               ranks (1.25e6 per rank at N=8), reported as "scaling": "strong".

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 bench.py --gpus 8 ...
    python bench.py --gpus 8 ...        # no launcher: starts the eight ranks itself (below)

Rank 0 prints ONE JSON line.

``--gpus N`` (N > 1) without a launcher environment: this process -- before it has made a single GPU
call, and it never makes one -- starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...``
as a CHILD process (never ``exec``), passes rank 0's JSON line through and exits with the child's code.
Fewer than N visible devices is an error (exit 3), never a silent one-rank run; a launcher whose
``WORLD_SIZE`` differs from ``--gpus`` likewise (exit 4).  ``--backend gloo --share-device`` puts all N
ranks on ``cuda:0`` with host collectives: the whole multi-rank bench (shard bounds, the sweep's device
record, the all-gather, ``ranks_seen``, ``best_check.vs_ranks``) on a single GPU -- a rehearsal of the
code path, not a scaling measurement (``config.rehearsal`` says so in the line).
"""

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_TFLOPS = 78.6   # AMD public MI355X FP64 matrix = FP64 vector peak
CSRC = os.path.join(ROOT, "approxposterior_amd", "csrc")


def synthetic_c3(n_train, ndim):
    """X seed 0 ~ U[-5,5]^D, y = -rosen(X)/100."""
    from scipy.optimize import rosen
    rs = np.random.RandomState(0)
    X = rs.uniform(-5.0, 5.0, size=(n_train, ndim))
    y = np.array([-rosen(x) / 100.0 for x in X])
    return X, y


def sweep_source_hash():
    """Identity of the kernel a profile was taken of: sha256 over the sweep's sources."""
    h = hashlib.sha256()
    for name in ("sweep.hip", "apgp_common.h"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profiled_traffic(n, d, m):
    """HBM-side bytes per sweep call from the newest committed rocprofv3 PMC summary
    (profiles/r*_pmc_sweep.json: FETCH_SIZE x2 for the gfx950 wide-stream correction +
    WRITE_SIZE, per MI355X_MICROARCH.md).  PMC counters cannot be collected inside this
    process, so the figure is only reported when it was measured on THIS kernel (the
    summary's ``sweep_source_sha`` equals the hash of the sources that are compiled now) and
    on this workload; otherwise None -- a stale number is never printed."""
    if (n, d, m) != (4096, 8, 1000000):
        return None, None
    import glob
    here = sweep_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sweep.json")), reverse=True):
        try:
            rec = json.load(open(path))
            if rec.get("sweep_source_sha") == here:
                return float(rec["derived"]["hbm_traffic_bytes_per_launch"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


def candidate_rows(m_total, ndim, lo_row, hi_row, keep_all=False):
    """Rows [lo_row, hi_row) of THE candidate matrix: ``RandomState(1).uniform(-5, 5, (m_total, ndim))``
    (SURVEY.md section 8d).  Every rank draws from the same stream and keeps its own rows, so the
    shards of any world size tile one global matrix and the printed winner can be re-derived.
    ``keep_all`` additionally returns the whole matrix (rank 0: oracle check of the winner)."""
    rs = np.random.RandomState(1)
    if keep_all:
        cands_all = rs.uniform(-5.0, 5.0, size=(m_total, ndim))
        return np.ascontiguousarray(cands_all[lo_row:hi_row]), cands_all
    left = lo_row                      # skip the lower ranks' rows without holding them
    while left > 0:
        n = min(left, 1 << 20)
        rs.uniform(-5.0, 5.0, size=(n, ndim))
        left -= n
    return np.ascontiguousarray(rs.uniform(-5.0, 5.0, size=(hi_row - lo_row, ndim))), None


def f_var(n, d):
    """Algorithmic flops per candidate, SURVEY.md section 8(d)."""
    return float(n) * n + float(n) * (3 * d + 4)


def oracle_gp(X, y, metric, ndim):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import george_oracle as go
    k = go.ExpSquaredKernel(np.full(ndim, metric), ndim=ndim)
    gp = go.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    t0 = time.time()
    gp.compute(X)
    gp._compute_alpha(y, True)
    return gp, time.time() - t0


def agp_utility(mu, var, inside):
    with np.errstate(all="ignore"):
        u = -(mu + 0.5 * np.log(2.0 * np.pi * np.e * var))
    return np.where(inside, u, np.inf)


def check_best(gpo, y, cands, best, records=None, sample_chunks=6, lo=-5.0, hi=5.0):
    """The winner against the oracle (the checker, never the thing measured).  Returns a dict:
      ``in_chunk``     the oracle's utility over the 4,096-candidate chunk that contains the
                       winning row has its minimum at that row, with the same value (1e-9 rel.);
      ``vs_sample``    no candidate of ``sample_chunks`` further 4,096-row chunks, drawn at
                       random over the WHOLE global matrix (all shards), beats it on the oracle;
      ``vs_ranks``     it is <= every rank's gathered winner, and each rank's winner row lies in
                       that rank's shard and carries the oracle's utility (multi-rank runs).
    A lost shard or a bad cross-rank combine fails ``vs_sample`` / ``vs_ranks`` even when the
    reported winner is the minimum of its own chunk."""
    bi, bu = int(best[0]), float(best[1])
    res = {"in_chunk": False, "vs_sample": False, "vs_ranks": None}
    if bi < 0:
        return res
    tol = 1e-9 * max(1.0, abs(bu))

    def oracle_u(rows):
        chunk = cands[rows]
        mu, var = gpo.predict(y, chunk, return_var=True)
        return agp_utility(mu, var, np.all((chunk >= lo) & (chunk <= hi), axis=1))
    c0 = (bi // 4096) * 4096
    u = oracle_u(slice(c0, c0 + 4096))
    j = bi - c0
    res["in_chunk"] = bool(abs(u[j] - bu) <= tol and u[j] <= np.nanmin(u) + tol)
    nchunk = (len(cands) + 4095) // 4096
    rs = np.random.RandomState(12345)
    ok = True
    for c in rs.choice(nchunk, size=min(sample_chunks, nchunk), replace=False):
        uu = oracle_u(slice(int(c) * 4096, int(c) * 4096 + 4096))
        ok = ok and bool(np.nanmin(uu) >= bu - tol)
    res["vs_sample"] = ok
    if records is not None and len(records) > 1:
        world = len(records)
        okr = True
        for r, (ru, ri) in enumerate(records):
            base, rem = divmod(len(cands), world)
            lo_r = r * base + min(r, rem)
            hi_r = lo_r + base + (1 if r < rem else 0)
            okr = okr and lo_r <= ri < hi_r and ru >= bu - tol
            ur = oracle_u(slice(ri, ri + 1))
            okr = okr and bool(abs(ur[0] - ru) <= 1e-9 * max(1.0, abs(ru)))
        res["vs_ranks"] = bool(okr)
    return res


def fit_leg(agp, dev, seconds_cap=20.0):
    """The fit side (the reference's stated bottleneck: ``gpUtils._nll`` inside ``optimizeGP``,
    gpUtils.py:46-80,223-247): one ``_nll`` evaluation = set_parameter_vector + log_likelihood
    (Gram + Cholesky + forward solve + log-determinant) at C1's, C5's final and C3's training-set
    sizes -- median of the GPU path and, beside it, the oracle's on the host cores (median of 5;
    BASELINE.md section 3(iii)).  Outside the headline's timed region; extra keys only."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import george_oracle as go
    from threadpoolctl import threadpool_info
    blas_threads = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    out = []
    for n, d in ((50, 2), (1152, 8), (4096, 8)):
        X, y = synthetic_c3(n, d)
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
                   white_noise=-12, fit_white_noise=False, device=dev)
        g.compute(X)
        p = g.get_parameter_vector()
        reps = 30 if n <= 1152 else 12
        ts = []
        for i in range(reps + 3):
            torch.cuda.synchronize()
            t0 = time.time()
            g.set_parameter_vector(p + 1e-3 * (i % 3))
            ll = g.log_likelihood(y, quiet=True)          # one 40-byte D2H copy = the synchronisation
            ts.append(time.time() - t0)
        gpu_ms = float(np.median(ts[3:])) * 1e3
        g.set_parameter_vector(p)
        ll = g.log_likelihood(y, quiet=True)
        o = go.GP(kernel=go.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
                  white_noise=-12, fit_white_noise=False)
        o.compute(X)
        tc = []
        for i in range(5):
            t0 = time.time()
            o.set_parameter_vector(p + 1e-3 * (i % 3))
            llo = o.log_likelihood(y, quiet=True)
            tc.append(time.time() - t0)
        o.set_parameter_vector(p)
        llo = o.log_likelihood(y, quiet=True)
        flops = n ** 3 / 3.0
        # what an optimiser pays per evaluation (gpUtils.py:238: SciPy's Powell on _nll), without / with the look-ahead of
        # gpUtils._powellAhead -- same points, same values, fewer device rounds
        powell = {}
        if n > 64:
            from scipy.optimize import minimize
            from approxposterior_amd import gpUtils
            for tag, ahead in (("off", 0), ("on", None)):
                g.lookahead = ahead
                g._nllMemo = None
                g.set_parameter_vector(p)
                cnt = [0]

                def fobj(q):
                    cnt[0] += 1
                    return gpUtils._nll(q, g, y)
                torch.cuda.synchronize()
                t0 = time.time()
                with np.errstate(all="ignore"):
                    res = minimize(fobj, p, method="powell", options={"maxfev": 500 if n <= 1152 else 200})
                powell[tag] = {"ms_per_evaluation": (time.time() - t0) / max(cnt[0], 1) * 1e3, "evaluations": cnt[0],
                               "fun": float(res["fun"])}
            g.lookahead = None
            powell["same_optimum"] = powell["off"]["fun"] == powell["on"]["fun"]
            powell["lookahead_width"] = g.lookahead_width()
        out.append({"n_train": n, "ndim": d, "nll_ms": gpu_ms, "tflops": flops / (gpu_ms * 1e-3) / 1e12, "powell": powell,
                    "frac_of_f64_peak": flops / (gpu_ms * 1e-3) / 1e12 / PEAK_F64_TFLOPS,
                    "cpu_nll_ms": float(np.median(tc)) * 1e3, "cpu_cores": os.cpu_count(), "cpu_blas_threads": blas_threads,
                    "cpu_kind": "port",
                    "ll_rel_diff": abs(ll - llo) / max(1.0, abs(llo))})
    return out


def mid_n_leg(agp, dev):
    """The same fused sweep at BASELINE.json's other sweep sizes -- C2 (N=1024, D=2, 1e5 candidates, BAPE) and
    C5's final size (N=1152, D=8, 1e6, AGP) -- HIP events around the launches of one call, median of 5 calls,
    fraction of the FP64 roof on F_var.  Outside the headline's timed region; extra keys only
    (tools/sweep_shapes.py is the developer version of this leg)."""
    import torch
    out = []
    for n, d, m, kind in ((1024, 2, 100000, "bape"), (1152, 8, 1000000, "agp")):
        X, y = synthetic_c3(n, d)
        T = torch.from_numpy(np.random.RandomState(1).uniform(-5.0, 5.0, size=(m, d))).to(dev)
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
                   white_noise=-12, fit_white_noise=False, device=dev)
        g.compute(X)
        g.acquire(y, T, kind, bounds=[(-5.0, 5.0)] * d)
        g.kernel_events = ev = []
        for _ in range(5):
            best = g.acquire(y, T, kind, bounds=[(-5.0, 5.0)] * d)
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        tf = f_var(n, d) * m / (ms * 1e-3) / 1e12
        out.append({"n_train": n, "ndim": d, "candidates": m, "utility": kind, "kernel_ms": ms,
                    "candidates_per_s": m / (ms * 1e-3), "tflops": tf, "frac_of_f64_peak": tf / PEAK_F64_TFLOPS,
                    "best": [int(best[0]), float(best[1])]})
        del T, g
    return out


def cpu_baseline(n_train, ndim, metric, seconds=12.0, scalar_calls=2000):
    """Reference-library batched path on the oracle, timed in a CHILD
    process that never touches the GPU (``bench.py --cpu-baseline-worker``, below): the host cores are
    what is measured, so the chunks run over a process pool -- OpenBLAS alone stops scaling long before
    256 cores and the oracle's kernel-row generation is single-threaded NumPy.  Returns the worker's
    record (``kind`` "port": george is not installable, this is its NumPy/SciPy restatement making the
    LAPACK calls george's BasicSolver makes)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--n-train", str(n_train),
           "--ndim", str(ndim), "--metric", repr(float(metric)), "--cpu-seconds", repr(float(seconds)),
           "--cpu-scalar-calls", str(int(scalar_calls))]
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # a launcher's "1 thread per rank"
        env.pop(k, None)
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, timeout=420.0 + 4.0 * seconds)
    except subprocess.TimeoutExpired:
        # the bench line is not held hostage by the host-side baseline: say so instead of a number
        return {"value": None, "unit": "candidates/s", "cores": None, "kind": "port",
                "sample": "cpu-baseline worker exceeded its time limit and was stopped"}
    if proc.returncode != 0:
        raise RuntimeError("cpu-baseline worker failed with exit code %d" % proc.returncode)
    return json.loads(proc.stdout.strip().splitlines()[-1])


_POOL = {}


def _pool_init(threads, n_train, ndim, metric):
    """A pool worker: its own oracle GP on the same training set (an identical factor in every worker;
    nothing is shared, so the workers scale like independent processes do) with ``threads`` BLAS threads."""
    from threadpoolctl import threadpool_limits
    import scipy.linalg  # noqa: F401 -- SciPy carries its OWN OpenBLAS: loaded before the limit below, or it is not limited
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import george_oracle  # noqa: F401
    _POOL["limits"] = threadpool_limits(limits=int(threads))
    X, y = synthetic_c3(n_train, ndim)
    gpo, _ = oracle_gp(X, y, metric, ndim)
    _POOL["gp"], _POOL["y"], _POOL["ndim"] = gpo, y, ndim


def _pool_chunk(job):
    """One chunk of candidates through the oracle's ``predict(return_var=True)`` -- its four statements
    (oracle/george_oracle.py GP.predict) timed one by one -- + the vectorised AGP utility + arg-min."""
    c, m = job
    gpo, y, ndim = _POOL["gp"], _POOL["y"], _POOL["ndim"]
    T = np.random.RandomState(1000 + int(c)).uniform(-5.0, 5.0, size=(int(m), ndim))
    t0 = time.time()
    Kxs = gpo.kernel.get_value(T, gpo._x)                      # kernel rows k*(T, X): NumPy, one thread
    t1 = time.time()
    mu = np.dot(Kxs, gpo._compute_alpha(y, True)) + gpo.mean.get_value(T)
    t2 = time.time()
    KinvKxs = gpo.apply_inverse(Kxs.T)                         # scipy cho_solve, one right-hand side per candidate
    t3 = time.time()
    var = gpo.kernel.get_value(T, diag=True) - np.sum(Kxs.T * KinvKxs, axis=0)
    u = agp_utility(mu, var, True)
    i = int(np.nanargmin(u))
    t4 = time.time()
    return len(T), float(u[i]), int(c) * int(m) + i, t1 - t0, t2 - t1, t3 - t2, t4 - t3


def cpu_baseline_worker(args):
    """Body of ``bench.py --cpu-baseline-worker`` (imports no torch, makes no GPU call).  Prints one JSON
    object: the ``cpu_baseline`` record of the bench line."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_info
    n, d = args.n_train, args.ndim
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:      # pragma: no cover
        cores = os.cpu_count() or 1
    blas = [{"api": i.get("internal_api"), "threads": i.get("num_threads"), "version": i.get("version")}
            for i in threadpool_info() if i.get("user_api") == "blas"]
    # one single-threaded worker per core while memory allows (kernel-row generation is single-threaded NumPy and a
    # one-thread triangular solve is the most efficient one); fewer, wider workers when it does not
    workers = cores
    try:
        import psutil
        avail = psutil.virtual_memory().available
        per_worker = 8.0 * n * n + 4 * 8.0 * 4096 * n + 3e8    # factor + Kxs, its transpose, the solve, slack
        workers = max(1, min(workers, int(0.5 * avail / per_worker)))
    except Exception:       # pragma: no cover
        pass
    threads = max(1, cores // workers)              # BLAS threads per pool worker
    X, y = synthetic_c3(n, d)
    gpo, fit_s = oracle_gp(X, y, args.metric, d)                # all BLAS threads: the fit time reported
    # (i) reference-faithful scalar path: one candidate per predict call, as utility.minimizeObjective drives it
    # (utility.py:131) -- SURVEY.md 8(d)(i): >= 2,000 calls; all BLAS threads, one process (the search is sequential)
    rs = np.random.RandomState(1)
    t1 = time.time()
    for _ in range(int(args.cpu_scalar_calls)):
        t = rs.uniform(-5.0, 5.0, size=(1, d))
        mu, var = gpo.predict(y, t, return_var=True)
        _ = agp_utility(mu, var, True)
    ds = time.time() - t1
    # (ii) batched path over the pool
    ctx = mp.get_context("spawn")       # fresh interpreters: no BLAS thread pool inherited across a fork
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = str(threads)    # read by the workers' BLAS libraries when THEY load them (this process's are loaded)
    pool = ctx.Pool(workers, initializer=_pool_init, initargs=(threads, n, d, args.metric))
    # chunks of 4,096 candidates while a wave of them over all workers stays inside the time budget; on a
    # many-core host, where the workers share the memory bandwidth (one chunk took 92 s with 256 of them at N = 4096), 1,024
    chunk = 4096 if workers <= 32 else 1024
    try:
        pool.map(_pool_chunk, [(c, 64) for c in range(workers)], chunksize=1)     # untimed: every worker fitted, pages touched
        done, best, parts, at = 0, (np.inf, -1), np.zeros(4), workers
        t0 = time.time()
        while True:
            for m, bu, bi, a, b, c, e in pool.map(_pool_chunk, [(c, chunk) for c in range(at, at + workers)], chunksize=1):
                done += m
                parts += (a, b, c, e)
                if bu < best[0]:
                    best = (bu, bi)
            at += workers
            if time.time() - t0 >= args.cpu_seconds:
                break
        dt = time.time() - t0
    finally:
        pool.terminate()
        pool.join()
    tot = float(parts.sum())
    print(json.dumps({
        "value": done / dt, "unit": "candidates/s", "cores": workers * threads, "kind": "port",
        "host_cores_visible": cores, "pool_workers": workers, "blas_threads_per_worker": threads,
        "blas_threadpoolctl": blas,
        "split": {"kernel_rows": parts[0] / tot, "mean": parts[1] / tot, "cho_solve": parts[2] / tot,
                  "variance_utility_argmin": parts[3] / tot,
                  "note": "share of the workers' summed time per statement of the oracle's predict"},
        "sample": "%d candidates in %.1f s (%d-candidate chunks over %d processes x %d BLAS threads, N_train=%d, "
                  "D=%d, oracle predict+utility+argmin; fit %.2f s excluded)" % (done, dt, chunk, workers, threads, n, d, fit_s),
        "scalar_path_value": args.cpu_scalar_calls / ds,
        "scalar_path_sample": "%d single-candidate predict+utility calls in %.1f s, one process, %s BLAS threads "
                              "(what the reference's Nelder-Mead search evaluates)"
                              % (args.cpu_scalar_calls, ds, blas[0]["threads"] if blas else "?")}))
    return 0


def self_launch(args, argv):
    """``--gpus N`` (N > 1) without a launcher: start the N ranks as a child ``torch.distributed.run`` and pass
    rank 0's line through.  Nothing here touches the GPU (``torch.cuda.device_count()`` only counts), and the
    program is never replaced (no ``os.exec*``): a child process, its exit code returned."""
    import torch
    ndev = torch.cuda.device_count()
    need = 1 if args.share_device else args.gpus
    if ndev < need:
        sys.stderr.write("bench.py: --gpus %d needs %d visible GPU(s), %d found -- refusing to run fewer ranks than "
                         "asked for (use --backend gloo --share-device to rehearse on one GPU)\n" % (args.gpus, need, ndev))
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-train", type=int, default=4096)
    ap.add_argument("--ndim", type=int, default=8)
    ap.add_argument("--candidates", type=int, default=1000000, help="candidates per GPU per step (weak scaling)")
    ap.add_argument("--total-candidates", type=int, default=0,
                    help="total candidates per step, split over the ranks (strong scaling; 10000000 = C4)")
    ap.add_argument("--metric", type=float, default=8.0)
    ap.add_argument("--utility", default="agp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of the winner")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-scalar-calls", type=int, default=2000)
    ap.add_argument("--variance", default="auto", choices=["auto", "inverse", "solve"],
                    help="predictive-variance formulation: the explicit L^-1 contraction (what 'auto' picks for this "
                         "well-conditioned workload; the headline) or the blocked substitution against L "
                         "(secondary measurement; what ill-conditioned factors get)")
    ap.add_argument("--no-fit-leg", action="store_true", help="skip the _nll timings (GPU and oracle)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collectives of a multi-rank run: RCCL over xGMI (nccl) or host collectives (gloo)")
    ap.add_argument("--share-device", action="store_true",
                    help="all ranks on cuda:0 (needs --backend gloo): rehearses the multi-rank path on ONE GPU")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.share_device and args.backend != "gloo":
        ap.error("--share-device needs --backend gloo (RCCL refuses two ranks on one device)")

    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ   # torch.distributed.run
    if args.gpus > 1 and not launched:
        return self_launch(args, sys.argv[1:])

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s) -- refusing to report a line "
                             "whose n_gpus is not what was asked for\n" % (args.gpus, world))
        return 4
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.share_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d has no device %d (%d visible)\n" % (rank, local_rank, torch.cuda.device_count()))
        return 3
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if launched:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from approxposterior_amd import gp as agp
    from approxposterior_amd import dist as adist

    N, D = args.n_train, args.ndim
    strong = args.total_candidates > 0
    M_total = args.total_candidates if strong else args.candidates * world
    lo_row, hi_row = adist.shard_bounds(M_total, world, rank)
    M = hi_row - lo_row
    X, y = synthetic_c3(N, D)

    def fit(phases=None):
        kernel = agp.ExpSquaredKernel(np.full(D, args.metric), ndim=D)
        g = agp.GP(kernel=kernel, fit_mean=True, mean=np.median(y), white_noise=-12,
                   fit_white_noise=False, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        g.compute(X)            # Gram + Cholesky (hybrid persistent plan at N = 4096) + summary
        if phases is not None:
            torch.cuda.synchronize(); phases["compute"] = (time.time() - t0) * 1e3; t1 = time.time()
        g._ensure_linv()        # W = L^-1 first: alpha is then two matrix-vector products (as GP._sweep does)
        if phases is not None:
            torch.cuda.synchronize(); phases["linv_pack"] = (time.time() - t1) * 1e3; t1 = time.time()
        g._ensure_xs(y)
        torch.cuda.synchronize()
        if phases is not None:
            phases["alpha_pack_train"] = (time.time() - t1) * 1e3
        return g, (time.time() - t0) * 1e3
    gp, fit_ms_cold = fit()     # includes module load, first allocations (~0.9 GB from the driver), attribute set-up
    # what a refit costs (approx.py:712-717: a new GP per appended point replaces the old one, whose buffers go back
    # to the caching allocator first): Gram + Cholesky + L^-1 + packing + solves -- median of three, then one more
    # pass with a synchronisation after each phase for the breakdown
    fit_runs = []
    for _ in range(3):
        del gp
        gp, ms = fit()
        fit_runs.append(ms)
    fit_ms = float(np.median(fit_runs))
    fit_phases = {}
    del gp
    gp, _ = fit(fit_phases)
    if args.variance != "auto":
        gp.variance_mode = args.variance

    # candidates: ONE global NumPy seed-1 draw; this rank's rows; resident in HBM before the
    # timed region (the H2D copy is reported separately)
    mine, cands_all = candidate_rows(M_total, D, lo_row, hi_row, keep_all=(rank == 0 and not args.no_check))
    torch.cuda.synchronize()
    t0 = time.time()
    T = torch.from_numpy(mine).to(dev)
    torch.cuda.synchronize()
    h2d_ms = (time.time() - t0) * 1e3
    bounds = [(-5.0, 5.0)] * D

    records = []                # the per-rank (u, index) records the last all-gather returned

    def step():
        return adist.sharded_acquire(
            lambda off: gp.acquire(y, T, args.utility, bounds=bounds, idx_offset=off, device_record=world > 1), lo_row,
            records=records)

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
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kms = [a.elapsed_time(b) for a, b in kernel_ms]

    if rank == 0:
        total = float(M_total) * args.steps
        value = total / elapsed
        k_avg_ms = float(np.mean(kms))
        achieved = f_var(N, D) * M / (k_avg_ms * 1e-3) / 1e12
        traffic, traffic_src = profiled_traffic(N, D, M)
        name = "C3" if (N, D, M_total, world) == (4096, 8, 1000000, 1) else \
               "C4" if (N, D, M_total) == (4096, 8, 10000000) else "C3-shaped"
        out = {
            "metric": "GP-predict+acquisition candidates/sec (N_train, D fixed)",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: synthetic D=%d log-likelihood (-rosen/100), N_train=%d, "
                                   "%d candidates in total (%d on rank 0), %s utility, box prior [-5,5]^D, "
                                   "ExpSquaredKernel metric %.1f, white_noise -12"
                                   % (name, D, N, M_total, M, args.utility.upper(), args.metric),
                       "n_train": N, "ndim": D, "candidates_total": M_total, "candidates_per_gpu": M,
                       "candidate_draw": "numpy RandomState(1).uniform(-5, 5, (candidates_total, D)); "
                                         "rank r owns rows [r M/world, (r+1) M/world)",
                       "sharding": "candidates split by rank, one 16 B/rank all-gather",
                       "backend": (args.backend if launched else None),
                       "rehearsal": ("all %d ranks share cuda:0 (host collectives): exercises the multi-rank code "
                                     "path, NOT a scaling measurement" % world) if args.share_device else None,
                       "fit_ms_warm": fit_ms, "fit_ms_warm_runs": fit_runs, "fit_phases_ms": fit_phases,
                       "fit_ms_cold_first_call": fit_ms_cold,
                       "h2d_candidates_ms": h2d_ms,
                       "h2d_candidates_GBps": mine.nbytes / (h2d_ms * 1e-3) / 1e9},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F64_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F64_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "sweep2_kernel<%d, false, 0>" % (2 if D <= 2 else 4 if D <= 4 else 8 if D <= 8 else 16),
                         "kernel_ms": k_avg_ms,
                         "algorithmic_flops_per_candidate": f_var(N, D)},
            "best": {"index": int(best[0]), "u": float(best[1])},
            "ranks_seen": len(records),      # records the last step's all-gather returned (== n_gpus)
            "variance": "solve" if not gp._trust_inverse() else "inverse",
        }
        if out["variance"] == "solve":
            # substitution form: MODE 2 (statically unrolled diagonal tiles) up to N = 2048, MODE 1 above
            out["roofline"]["kernel"] = out["roofline"]["kernel"].replace(", 0>", ", 2>" if N <= 2048 else ", 1>")
            out["roofline"]["traffic"] = None
            out["roofline"]["traffic_source"] = None
        if not args.no_check:
            gpo, fit_s = oracle_gp(X, y, args.metric, D)
            chk = check_best(gpo, y, cands_all, best, records=records)
            out["best_checked"] = bool(chk["in_chunk"] and chk["vs_sample"] and chk["vs_ranks"] is not False)
            out["best_check"] = chk
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, D, args.metric, args.cpu_seconds, args.cpu_scalar_calls)
            # the same reference-faithful scalar path on the GPU (one candidate per predict call:
            # apgp_predict1_host), beside the oracle's scalar_path_value
            rs1 = np.random.RandomState(2)
            t1 = rs1.uniform(-5.0, 5.0, size=(300, D))
            for i in range(20):
                gp.predict(y, t1[i:i + 1], return_var=True)
            t0 = time.time()
            for i in range(20, 300):
                gp.predict(y, t1[i:i + 1], return_var=True)
            out["cpu_baseline"]["scalar_path_gpu_value"] = 280.0 / (time.time() - t0)
        if world == 1 and not args.no_fit_leg:
            out["fit"] = fit_leg(agp, dev)
            out["mid_n"] = mid_n_leg(agp, dev)
        print(json.dumps(out))
    if launched:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
