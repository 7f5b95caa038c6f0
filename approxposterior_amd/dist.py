# -*- coding: utf-8 -*-
"""
:py:mod:`dist.py` - candidate-sharded acquisition over the GPUs of one node
---------------------------------------------------------------------------

The candidate sweep shards naturally: candidates are independent units
(SURVEY.md section 8e).  Every rank holds the whole (small) training set and its
own factor (replicated fit, zero data-path communication), evaluates its
contiguous slice of the candidate matrix with the fused HIP sweep, and the
per-rank winners (utility, global index) are exchanged with ONE all-gather of
16 bytes per rank (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in the
CPU tests).  Ties resolve to the lowest global index, so the result is
independent of the number of ranks.

The reference has no counterpart (it minimises a scalar utility with
Nelder-Mead, utility.py:253-372); this is the multi-GPU form of
``GP.acquire`` and obeys the same arg-min contract.
"""

import numpy as np

__all__ = ["shard_bounds", "combine_best", "sharded_acquire", "replicated_ensembles"]


def shard_bounds(m, world_size, rank):
    """Contiguous row range [lo, hi) of ``m`` candidates owned by ``rank``."""
    base, rem = divmod(int(m), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def combine_best(pairs):
    """Arg-min over (u, global_index) pairs: NaN and +inf never win, index -1 =
    no admissible candidate, ties -> lowest global index (same rule as the
    device reduction)."""
    best_u, best_i = np.inf, -1
    for u, i in pairs:
        i = int(i)
        if i < 0 or np.isnan(u) or u == np.inf:
            continue
        if u < best_u or (u == best_u and (best_i < 0 or i < best_i)):
            best_u, best_i = float(u), i
    return best_i, best_u


def sharded_acquire(local_acquire, idx_offset, group=None, device=None, records=None):
    """Run ``local_acquire(idx_offset) -> (best_global_index, best_u)`` on this
    rank's shard and all-gather the winners.

    ``local_acquire`` is typically
    ``lambda off: gp.acquire(y, T_local, kind, bounds=..., idx_offset=off, device_record=True)``
    (the 16-byte record stays in HBM until the gather) or the same without ``device_record``
    (a host ``(index, u)`` pair).
    Returns the same (index, u) on every rank.  ``records`` (a list) receives the gathered
    per-rank (u, index) pairs in rank order -- one per rank the collective actually saw.
    """
    import torch
    import torch.distributed as dist

    res = local_acquire(idx_offset)
    on_device = torch.is_tensor(res)        # the sweep's own 16-byte record, still in HBM (GP.acquire(device_record=True))
    if on_device:
        if res.dtype != torch.int64 or res.numel() != 2:
            raise ValueError("a device record is an int64[2] tensor: bit pattern of best_u, best_index")
        mine = res.reshape(2)
    else:
        bi, bu = res
    if not (dist.is_available() and dist.is_initialized()):
        if on_device:
            h = mine.cpu().numpy()
            bu, bi = float(h[0:1].view(np.float64)[0]), int(h[1])
        if records is not None:
            records[:] = [(float(bu), int(bi))]
        return combine_best([(bu, bi)])
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
    # one 16-byte record per rank as two int64 words: the utility's bit pattern and the index.
    # (Integer words survive any transport unchanged -- a float64 carrier for the index would be
    # at the mercy of NaN canonicalisation the day someone swaps the gather for a reduction.)
    # With a device record and the nccl backend nothing touches the host before the gather: the words go from
    # the arg-min kernel's output straight into the collective, and ONE copy brings all ranks' records back.
    if not on_device:
        mine = torch.tensor([int(np.float64(bu).view(np.int64)), int(bi)], dtype=torch.int64, device=device)
    elif mine.device != device:
        mine = mine.to(device)
    world = dist.get_world_size(group)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    h = torch.stack(gathered).cpu().numpy()                 # (world, 2) int64, one copy
    pairs = [(float(h[r, 0:1].view(np.float64)[0]), int(h[r, 1])) for r in range(world)]
    if records is not None:
        records[:] = pairs
    return combine_best(pairs)


def replicated_ensembles(local_sample, seed=0, group=None, device=None):
    """MCMC over the GP surrogate on several GPUs: *replicas only* (SURVEY.md
    section 8e, row "MCMC _gpll").  A stretch-move step couples all walkers of an
    ensemble, so an ensemble does not shard; instead every rank runs its own
    independent ensemble(s) with a rank-specific seed --
    ``local_sample(seed + rank) -> (chain (iterations, W, D), log_prob (iterations, W))``,
    typically ``lambda s: (lambda r: (r["chain"], r["log_prob"]))(gp.sample_ensemble(y, p0, iters,
    bounds, seed=s))`` -- and the chains are concatenated along the walker axis with
    ONE all-gather at the end (no collective inside the sampling loop).
    Returns (chain (iterations, world*W, D), log_prob (iterations, world*W)) on every rank.
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    chain, logp = local_sample(int(seed) + rank)
    chain = np.ascontiguousarray(chain, dtype=np.float64)
    logp = np.ascontiguousarray(logp, dtype=np.float64)
    if chain.ndim != 3 or logp.shape != chain.shape[:2]:
        raise ValueError("local_sample must return chain (iterations, W, D) and log_prob (iterations, W)")
    if not distributed:
        return chain, logp
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
    world = dist.get_world_size(group)
    # one record per rank: [chain | log_prob] flattened (same shape on every rank)
    mine = torch.from_numpy(np.concatenate([chain.ravel(), logp.ravel()])).to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    chains, logps = [], []
    for g in gathered:
        h = g.cpu().numpy()
        chains.append(h[:chain.size].reshape(chain.shape))
        logps.append(h[chain.size:].reshape(logp.shape))
    return np.concatenate(chains, axis=1), np.concatenate(logps, axis=1)
