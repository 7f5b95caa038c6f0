"""BASELINE configuration C5 on the GPUs of one node: ``ApproxPosterior.run`` at D = 8, m0 = 512, m = 64, nmax = 10
(N grows 512 -> 1152), 1e6 sweep candidates per design point, 64 walkers x 2e4 iterations per rank -- one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/run_c5_dist.py [--m 64 --nmax 10 --candidates 1000000 --iterations 20000 --restarts 1]

(also runs as a plain ``python tools/run_c5_dist.py`` = one rank; ``--backend gloo`` keeps the collectives on the host, the
arithmetic still runs on the GPU: there is no CPU path; ``--backend gloo --share-device`` puts all ranks on cuda:0 -- the
whole multi-rank loop on the one GPU the driver has, ``tests/test_gpu_dist_nccl.py``).  The process group is created BEFORE anything touches the GPU and the script never
re-executes itself.  What the ranks share (approxposterior_amd/dist.py): the candidate sweep is sharded by rank with one
16-byte-per-rank all-gather, every rank samples its own replica ensemble and the chains are gathered once, optimiser
restarts are spread over the ranks, the forward model runs on rank 0 and its value is broadcast.  Rank 0 prints one JSON
line: wall time, time per phase, and a cross-rank digest check (every rank must hold the same training set).
Reference loop: /root/reference/approxposterior/approx.py:396-424 (run), :664-672 (point search), :839-856 (sampler)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--m", "--points", dest="m", type=int, default=64,
                     help="design points per outer iteration (--points: torch.distributed.run's own parser takes a bare --m as "
                          "an ambiguous prefix of its options)")
    ap_.add_argument("--nmax", type=int, default=10)
    ap_.add_argument("--m0", type=int, default=512)
    ap_.add_argument("--dim", type=int, default=8)
    ap_.add_argument("--candidates", type=int, default=1_000_000)
    ap_.add_argument("--iterations", type=int, default=20000)
    ap_.add_argument("--walkers", type=int, default=64)
    ap_.add_argument("--restarts", type=int, default=1)
    ap_.add_argument("--backend", default="nccl")
    ap_.add_argument("--share-device", action="store_true",
                     help="all ranks on cuda:0 (needs --backend gloo): the multi-rank loop rehearsed on ONE GPU")
    args = ap_.parse_args()
    if args.share_device and args.backend != "gloo":
        ap_.error("--share-device needs --backend gloo (RCCL refuses two ranks on one device)")

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0 if args.share_device else local)
    dist.init_process_group(args.backend, rank=rank, world_size=world)

    from scipy.optimize import rosen
    from approxposterior_amd import approx, gpUtils
    D = args.dim
    lo, hi = -5.0, 5.0
    bounds = [(lo, hi)] * D

    def lnprior(t):
        t = np.asarray(t)
        return 0.0 if np.all((t >= lo) & (t <= hi)) else -np.inf

    def sample(n=1):
        return np.random.uniform(lo, hi, size=(n, D))

    def lnlike(t, *a, **k):
        return -rosen(np.asarray(t).ravel()) / 100.0

    np.random.seed(11)                 # the same initial training set on every rank
    theta = sample(args.m0)
    y = np.array([lnlike(t) + lnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, y)
    driver = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lnprior, lnlike=lnlike, priorSample=sample,
                                    bounds=bounds, algorithm="agp")
    dist.barrier()
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        driver.run(m=args.m, nmax=args.nmax, nCandidates=args.candidates, nGPRestarts=args.restarts, cache=False,
                   verbose=False, onDevice=True, estBurnin=True, thinChains=True, timing=True,
                   mcmcKwargs={"iterations": args.iterations}, samplerKwargs={"nwalkers": args.walkers})
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    digest = hashlib.sha256(driver.theta.tobytes() + driver.y.tobytes()
                            + driver.gp.get_parameter_vector().tobytes()).digest()[:8]
    mine = torch.tensor(list(digest), dtype=torch.int64)
    if args.backend == "nccl":
        mine = mine.cuda()
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    same = all(bool((g == got[0]).all()) for g in got)
    if rank == 0:
        print(json.dumps({"config": "C5", "world": world, "backend": args.backend, "wall_s": round(wall, 2),
                          "training_s": [round(v, 2) for v in driver.trainingTime],
                          "mcmc_s": [round(v, 2) for v in driver.mcmcTime],
                          "n_train": int(len(driver.y)), "chain_walkers": int(driver.sampler.get_chain().shape[1]),
                          "ranks_agree": same, "digest": digest.hex()}))
    dist.destroy_process_group()
    if not same:
        sys.exit(3)


if __name__ == "__main__":
    main()
