"""CPU, world_size 2 over gloo: the candidate-sharded acquisition's collective
(approxposterior_amd.dist).  The per-shard evaluator is the oracle (there is no
HIP on this box); what is tested is the sharding, the 16-byte all-gather record,
the tie-break and that every rank gets the single-process answer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from approxposterior_amd.dist import combine_best, shard_bounds, sharded_acquire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for m in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(m, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            for a, b in zip(spans[:-1], spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_combine_best_rules():
    assert combine_best([(np.inf, -1), (np.nan, 5)]) == (-1, np.inf)
    assert combine_best([(1.0, 9), (1.0, 3), (2.0, 0)]) == (3, 1.0)       # tie -> lowest index
    assert combine_best([(np.nan, 0), (-2.0, 8)]) == (8, -2.0)            # NaN never wins
    assert combine_best([(np.inf, 4), (np.inf, 2)]) == (-1, np.inf)          # +inf never wins


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, u_all, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(len(u_all), world, rank)

    def local(off):
        u = u_all[lo:hi]
        um = np.where(np.isnan(u), np.inf, u)
        if len(u) == 0 or not np.isfinite(um).any():
            return -1, np.inf
        i = int(np.argmin(um))
        return off + i, float(um[i])

    res = sharded_acquire(local, lo)
    np.save(out_path % rank, np.array([res[0], res[1]], dtype=np.float64))
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["random", "tie_across_ranks", "all_inadmissible"])
def test_sharded_acquire_world2_gloo(tmp_path, case):
    rs = np.random.RandomState(5)
    u = rs.normal(size=1001)
    u[17] = np.nan
    if case == "tie_across_ranks":
        u[10] = u[900] = u.min() - 1.0
    if case == "all_inadmissible":
        u[:] = np.inf
    um = np.where(np.isnan(u), np.inf, u)
    want = (int(np.argmin(um)), float(um.min())) if np.isfinite(um).any() else (-1, np.inf)
    port = _free_port()
    out = str(tmp_path / "r%d.npy")
    mp.spawn(_worker, args=(2, port, u, out), nprocs=2, join=True)
    for r in range(2):
        got = np.load(out % r)
        assert int(got[0]) == want[0] and got[1] == want[1]


def _worker8(rank, world, port, cases, out_path):
    """world 8: every case through BOTH record forms -- the host (index, u) pair and the int64[2] tensor that
    GP.acquire(device_record=True) leaves in HBM (a CPU tensor here) -- with the gathered per-rank records kept."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for name, u_all in cases:
        lo, hi = shard_bounds(len(u_all), world, rank)

        def local(off, as_tensor):
            u = u_all[lo:hi]
            um = np.where(np.isnan(u), np.inf, u)
            if len(u) == 0 or not np.isfinite(um).any():
                bi, bu = -1, np.inf
            else:
                i = int(np.argmin(um))
                bi, bu = off + i, float(um[i])
            if as_tensor:
                return torch.tensor([int(np.float64(bu).view(np.int64)), bi], dtype=torch.int64)
            return bi, bu

        for form in (False, True):
            rec = []
            res = sharded_acquire(lambda off: local(off, form), lo, records=rec)
            out["%s_%d" % (name, form)] = np.array([res[0], res[1]] + [v for pair in rec for v in pair], dtype=np.float64)
    np.savez(out_path % rank, **out)
    dist.destroy_process_group()


def test_sharded_acquire_world8_gloo(tmp_path):
    """The collective at the world size it is built for (C4: 8 x MI355X): remainder shards (M % 8 != 0), the
    gathered records in rank order and equal to each shard's own winner, a tie between ranks 0 and 7 (lowest
    global index wins), one rank whose whole shard is inadmissible (index -1 record, never the winner), fewer
    candidates than ranks (empty shards) -- for the host-pair and the device-tensor record alike."""
    rs = np.random.RandomState(11)
    world = 8
    base = rs.normal(size=1003)                                  # 1003 = 8 * 125 + 3: three ranks own one row more
    base[17] = np.nan
    tie = base.copy()
    lo7, hi7 = shard_bounds(len(tie), world, 7)
    tie[5] = tie[lo7 + 3] = np.nanmin(tie) - 1.0                 # the same value on ranks 0 and 7
    dead = base.copy()
    lo3, hi3 = shard_bounds(len(dead), world, 3)
    dead[lo3:hi3] = np.inf                                       # rank 3 has no admissible candidate
    dead[lo3 + 1] = np.nan
    few = np.array([0.5, -0.25, 3.0])                            # 3 candidates on 8 ranks: five empty shards
    cases = [("random", base), ("tie", tie), ("dead", dead), ("few", few), ("none", np.full(40, np.inf))]
    port = _free_port()
    out = str(tmp_path / "w%d.npz")
    mp.spawn(_worker8, args=(world, port, cases, out), nprocs=world, join=True)
    for name, u in cases:
        um = np.where(np.isnan(u), np.inf, u)
        want = (int(np.argmin(um)), float(um.min())) if np.isfinite(um).any() else (-1, np.inf)
        for form in (0, 1):
            for r in range(world):
                got = np.load(out % r)["%s_%d" % (name, form)]
                assert int(got[0]) == want[0] and got[1] == want[1], (name, form, r, got[:2], want)
                recs = got[2:].reshape(world, 2)                 # (u, index) per rank, in rank order
                for q in range(world):
                    lo, hi = shard_bounds(len(u), world, q)
                    uq = um[lo:hi]
                    if len(uq) == 0 or not np.isfinite(uq).any():
                        assert recs[q, 1] == -1 and recs[q, 0] == np.inf
                    else:
                        assert recs[q, 1] == lo + int(np.argmin(uq)) and recs[q, 0] == uq.min()
    assert want == (-1, np.inf)                                  # (the last case: nothing admissible anywhere)
    # the tie went to rank 0's index although rank 7 holds the same utility
    assert int(np.load(out % 0)["tie_1"][0]) == 5


def _replica_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from approxposterior_amd.dist import replicated_ensembles

    def local(seed):          # stand-in for gp.sample_ensemble: the chain depends on the seed only
        rs = np.random.RandomState(seed)
        return rs.normal(size=(7, 4, 3)), rs.normal(size=(7, 4))

    chain, logp = replicated_ensembles(local, seed=100)
    np.savez(out_path % rank, chain=chain, logp=logp)
    dist.destroy_process_group()


def test_replicated_ensembles_world2_gloo(tmp_path):
    """the MCMC does not shard per step; ranks run independent ensembles
    (rank-specific seeds) and gather the chains once at the end."""
    from approxposterior_amd.dist import replicated_ensembles
    port = _free_port()
    out = str(tmp_path / "c%d.npz")
    mp.spawn(_replica_worker, args=(2, port, out), nprocs=2, join=True)
    want_chain, want_logp = [], []
    for r in range(2):
        rs = np.random.RandomState(100 + r)
        want_chain.append(rs.normal(size=(7, 4, 3)))
        want_logp.append(rs.normal(size=(7, 4)))
    want_chain = np.concatenate(want_chain, axis=1)
    want_logp = np.concatenate(want_logp, axis=1)
    for r in range(2):
        got = np.load(out % r)
        assert got["chain"].shape == (7, 8, 3)
        assert np.array_equal(got["chain"], want_chain) and np.array_equal(got["logp"], want_logp)
    # without a process group: the local ensemble, unchanged
    c, l = replicated_ensembles(lambda s: (np.ones((2, 4, 3)) * s, np.zeros((2, 4))), seed=5)
    assert c.shape == (2, 4, 3) and np.all(c == 5)
    with pytest.raises(ValueError):
        replicated_ensembles(lambda s: (np.ones((2, 4)), np.zeros((2, 4))))


def test_replicated_ensembles_world8_gloo(tmp_path):
    """Eight independent ensembles, one per rank, gathered once: walkers concatenated in rank order, every rank
    holds the same arrays."""
    port = _free_port()
    out = str(tmp_path / "c%d.npz")
    mp.spawn(_replica_worker, args=(8, port, out), nprocs=8, join=True)
    want_chain, want_logp = [], []
    for r in range(8):
        rs = np.random.RandomState(100 + r)
        want_chain.append(rs.normal(size=(7, 4, 3)))
        want_logp.append(rs.normal(size=(7, 4)))
    want_chain = np.concatenate(want_chain, axis=1)
    want_logp = np.concatenate(want_logp, axis=1)
    for r in range(8):
        got = np.load(out % r)
        assert got["chain"].shape == (7, 32, 3)
        assert np.array_equal(got["chain"], want_chain) and np.array_equal(got["logp"], want_logp)


def test_bench_candidate_shards_tile_the_global_draw():
    """bench.py's candidate matrix (C3 / C4): rank r's rows are rows
    [r M/world, (r+1) M/world) of ONE NumPy seed-1 draw, for every world size -- so 8 ranks x 1.25e6
    (C4 as written, --total-candidates) sweep exactly the matrix one rank would."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from approxposterior_amd.dist import shard_bounds
    M, D = 100003, 8
    whole = np.random.RandomState(1).uniform(-5.0, 5.0, size=(M, D))
    for world in (1, 2, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(M, world, r)
            mine, allc = bench.candidate_rows(M, D, lo, hi, keep_all=(r == 0))
            assert mine.shape == (hi - lo, D) and mine.flags["C_CONTIGUOUS"]
            if r == 0:
                assert np.array_equal(allc, whole)
            parts.append(mine)
        assert np.array_equal(np.vstack(parts), whole)
