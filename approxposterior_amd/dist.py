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

__all__ = ["shard_bounds", "combine_best", "sharded_acquire", "replicated_ensembles", "context",
           "broadcast_bytes", "sync_random_state", "all_gather_arrays", "spread_restarts", "raise_together"]


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


def sharded_acquire(local_acquire, idx_offset, group=None, device=None, records=None, enabled=None):
    """Run ``local_acquire(idx_offset) -> (best_global_index, best_u)`` on this
    rank's shard and all-gather the winners.

    ``local_acquire`` is typically
    ``lambda off: gp.acquire(y, T_local, kind, bounds=..., idx_offset=off, device_record=True)``
    (the 16-byte record stays in HBM until the gather) or the same without ``device_record``
    (a host ``(index, u)`` pair).
    Returns the same (index, u) on every rank.  ``records`` (a list) receives the gathered
    per-rank (u, index) pairs in rank order -- one per rank the collective actually saw.
    ``enabled`` as in :func:`context`: False = this rank's own record only, whatever process group the
    process happens to have open (``ApproxPosterior(distributed=False)``).
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
    if context(group, enabled) is None:
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


def replicated_ensembles(local_sample, seed=0, group=None, device=None, enabled=None):
    """MCMC over the GP surrogate on several GPUs: *replicas only* (SURVEY.md
    section 8e, row "MCMC _gpll").  A stretch-move step couples all walkers of an
    ensemble, so an ensemble does not shard; instead every rank runs its own
    independent ensemble(s) with a rank-specific seed --
    ``local_sample(seed + rank) -> (chain (iterations, W, D), log_prob (iterations, W))``,
    typically ``lambda s: (lambda r: (r["chain"], r["log_prob"]))(gp.sample_ensemble(y, p0, iters,
    bounds, seed=s))`` -- and the chains are concatenated along the walker axis with
    ONE all-gather at the end (no collective inside the sampling loop).
    Returns (chain (iterations, world*W, D), log_prob (iterations, world*W)) on every rank.
    ``local_sample`` may return further arrays: (iterations, W) per-step values (e.g. the lnprior blobs)
    or (W,) per-walker values (e.g. the accepted-move counts); each is concatenated along its walker
    axis and returned after the first two.  ``enabled`` as in :func:`context` (False: the local ensemble,
    seed unchanged, no collective).
    """
    ctx = context(group, enabled)
    rank = ctx[0] if ctx is not None else 0
    res = local_sample(int(seed) + rank)
    chain = np.ascontiguousarray(res[0], dtype=np.float64)
    logp = np.ascontiguousarray(res[1], dtype=np.float64)
    if chain.ndim != 3 or logp.shape != chain.shape[:2]:
        raise ValueError("local_sample must return chain (iterations, W, D) and log_prob (iterations, W)")
    extras = [np.ascontiguousarray(e, dtype=np.float64) for e in res[2:]]
    for e in extras:
        if e.shape != logp.shape and e.shape != logp.shape[1:]:
            raise ValueError("extra arrays must be (iterations, W) or (W,)")
    if ctx is None:
        return (chain, logp) + tuple(extras)
    per_rank = all_gather_arrays([chain, logp] + extras, group, device, enabled)
    out = []
    for k in range(2 + len(extras)):
        axis = 0 if per_rank[0][k].ndim == 1 else 1
        out.append(np.concatenate([r[k] for r in per_rank], axis=axis))
    return tuple(out)


# ------------------------------------------------------------------------------------------------
# The pieces ApproxPosterior / gpUtils.optimizeGP use when they run under torch.distributed.run
# (one process per GPU): every rank executes the SAME outer loop on an identical training set; the
# candidate sweep is sharded (above), MCMC ensembles and optimiser restarts are spread over the ranks,
# and the few host values that must agree everywhere (NumPy's global random state, a selected point,
# the forward-model value rank 0 computed) travel by broadcast.
# ------------------------------------------------------------------------------------------------

def context(group=None, enabled=None):
    """``(rank, world)`` when the multi-GPU paths apply, else ``None``.

    ``enabled`` None: whenever a torch.distributed process group is initialised (world size 1 included:
    the collectives then run on a single rank); False: never; True: required (``RuntimeError`` without a
    group)."""
    if enabled is False:
        return None
    try:
        import torch.distributed as dist
    except ImportError:      # pragma: no cover
        dist = None
    if dist is None or not (dist.is_available() and dist.is_initialized()):
        if enabled:
            raise RuntimeError("distributed=True needs an initialised torch.distributed process group")
        return None
    return dist.get_rank(group), dist.get_world_size(group)


def _carrier(group, device):
    """Device the collectives' tensors live on: the current GPU for RCCL (``nccl``), the host for gloo."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return device
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _global_rank(group, group_rank):
    import torch.distributed as dist
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def broadcast_bytes(buf, src=0, group=None, device=None, enabled=None):
    """Every rank returns rank ``src``'s ``buf`` (a C-contiguous NumPy array; same shape and dtype
    everywhere -- only the bytes travel)."""
    import torch
    import torch.distributed as dist
    arr = np.ascontiguousarray(buf)
    if context(group, enabled) is None:
        return arr
    raw = np.frombuffer(arr.tobytes(), dtype=np.uint8).copy()
    t = torch.from_numpy(raw).to(_carrier(group, device))
    dist.broadcast(t, src=_global_rank(group, src), group=group)
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=arr.dtype).reshape(arr.shape).copy()


def sync_random_state(src=0, group=None, device=None, enabled=None):
    """Give every rank rank ``src``'s global NumPy random state (MT19937 key, position, cached Gaussian):
    afterwards identical calls draw identical numbers on all ranks -- the candidate matrix, restart start
    points and walker initial states are then ONE global draw, not one per rank."""
    if context(group, enabled) is None:
        return
    name, key, pos, has_gauss, gauss = np.random.get_state()
    if name != "MT19937":      # pragma: no cover
        raise RuntimeError("unexpected NumPy bit generator %r" % name)
    head = broadcast_bytes(np.array([pos, has_gauss], dtype=np.int64), src, group, device, enabled)
    key = broadcast_bytes(np.asarray(key, dtype=np.uint32), src, group, device, enabled)
    gauss = broadcast_bytes(np.array([gauss], dtype=np.float64), src, group, device, enabled)
    np.random.set_state((name, key, int(head[0]), int(head[1]), float(gauss[0])))


def all_gather_arrays(arrays, group=None, device=None, enabled=None):
    """ONE all-gather of a list of equally-shaped (across ranks) float64 arrays: returns
    ``per_rank[r][k]`` = rank r's k-th array.  Without a process group: ``[arrays]``."""
    import torch
    import torch.distributed as dist
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    if context(group, enabled) is None:
        return [arrs]
    flat = np.concatenate([a.ravel() for a in arrs]) if arrs else np.empty(0)
    mine = torch.from_numpy(flat).to(_carrier(group, device))
    world = dist.get_world_size(group)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    out = []
    for g in gathered:
        h = g.cpu().numpy()
        parts, at = [], 0
        for a in arrs:
            parts.append(h[at:at + a.size].reshape(a.shape).copy())
            at += a.size
        out.append(parts)
    return out


def spread_restarts(n_restarts, run_mine, n_params, group=None, device=None, enabled=None):
    """Optimiser restarts over the ranks (SURVEY.md section 8e, row "GP fit": the fit does not shard,
    its restarts do): restart r belongs to rank ``r % world``; ``run_mine(indices) -> [(mll, p), ...]``
    runs this rank's restarts; ONE all-gather of (2 + P) doubles per restart slot returns
    ``(mll (R,), p (R, P))`` in restart order on every rank.

    The ranks do different work before the collective, so a failure must not leave the others waiting
    in it: an exception in ``run_mine`` is caught, travels as a status word in the same all-gather, and
    then EVERY rank raises (the failing rank its own exception, the others a ``RuntimeError`` naming it)."""
    ctx = context(group, enabled)
    rank, world = ctx if ctx is not None else (0, 1)
    mine = list(range(rank, int(n_restarts), world))
    err = None
    try:
        res = run_mine(mine) if mine else []
        if len(res) != len(mine):
            raise ValueError("run_mine must return one (mll, p) per restart index")
    except Exception as e:       # noqa: BLE001 -- re-raised below, after the collective every rank is about to enter
        if ctx is None:
            raise
        err, res = e, []
    slots = (int(n_restarts) + world - 1) // world
    rec = np.full((slots, 2 + int(n_params)), np.nan)
    rec[:, 0] = 0.0 if err is None else 1.0        # status word of this rank
    for s, (mll, p) in enumerate(res):
        rec[s, 1] = mll
        rec[s, 2:] = np.asarray(p, dtype=np.float64)
    per_rank = all_gather_arrays([rec], group, device, enabled)
    failed = [r for r in range(world) if per_rank[r][0][0, 0] != 0.0] if slots else []
    if err is not None:
        raise err
    if failed:
        raise RuntimeError("optimiser restarts failed on rank(s) %s" % failed)
    mll = np.empty(int(n_restarts))
    ps = np.empty((int(n_restarts), int(n_params)))
    for r in range(int(n_restarts)):
        row = per_rank[r % world][0][r // world]
        mll[r], ps[r] = row[1], row[2:]
    return mll, ps


def raise_together(err, what, src=None, group=None, device=None, enabled=None):
    """Call on EVERY rank after a step only some ranks worked in (``err``: the exception this rank caught,
    or None): one all-gather of a status word; if any rank failed, every rank raises -- the failing
    rank(s) their own exception, the others ``RuntimeError(what ...)`` -- instead of blocking forever in
    the next collective.  Without a process group: re-raises ``err`` if there is one."""
    ctx = context(group, enabled)
    if ctx is None:
        if err is not None:
            raise err
        return
    flags = all_gather_arrays([np.array([0.0 if err is None else 1.0])], group, device, enabled)
    failed = [r for r, f in enumerate(flags) if f[0][0] != 0.0]
    if err is not None:
        raise err
    if failed:
        raise RuntimeError("%s failed on rank(s) %s" % (what, failed))
