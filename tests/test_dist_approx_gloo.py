"""CPU, world_size 2 over gloo: the multi-GPU wiring INSIDE ``ApproxPosterior`` (BASELINE
config 5 "on 8 GPUs"; /root/reference/approxposterior/approx.py:396-424 outer loop, :664-672 point search,
:839-856 sampler) -- ``findNextPoint`` shards the ``nCandidates`` sweep by rank and all-gathers the winners,
``runMCMC`` runs one replica ensemble per rank and gathers the chains, ``optimizeGP`` spreads its restarts over the
ranks, the forward model runs on rank 0 only, every rank ends with the same training set and hyper-parameters.

The GP is a stub: the oracle's arithmetic behind the members the distributed paths use (``acquire`` with
``idx_offset`` / ``device_record``, ``sample_ensemble``, ``nll_batch``).  No HIP runs here; what is tested is the
host logic and the collectives (gloo)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS = ((-5.0, 5.0), (-5.0, 5.0))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_module():
    """A ``george``-shaped namespace whose GP is the oracle GP + the batched members of the HIP GP."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import george_oracle as go
    from approxposterior_amd import gpUtils, mcmc

    class StubGP(go.GP):
        calls = {"acquire_rows": [], "sample_seeds": [], "box_rows": []}

        def acquire(self, y, t, kind, bounds=None, mask=None, zeta=0.01, return_all=False, idx_offset=0,
                    device_record=False):
            t = self.parse_samples(np.asarray(t)) if len(t) else np.empty((0, self.kernel.ndim))
            StubGP.calls["acquire_rows"].append(len(t))
            bi, bu = -1, np.inf
            if len(t):
                mu, var = self.predict(y, t, return_var=True)
                with np.errstate(all="ignore"):
                    if kind == "agp":
                        u = -(mu + 0.5 * np.log(2.0 * np.pi * np.e * var))
                    elif kind == "bape":
                        u = -((2.0 * mu + var) + var + np.log(1.0 - np.exp(-var)))
                    else:
                        raise ValueError(kind)
                b = np.asarray(bounds, dtype=float)
                ok = np.all((t >= b[:, 0]) & (t <= b[:, 1]), axis=1)
                u = np.where(ok & ~np.isnan(u), u, np.inf)
                if np.isfinite(u).any():
                    i = int(np.argmin(u))
                    bi, bu = idx_offset + i, float(u[i])
            if device_record:
                return torch.tensor([int(np.float64(bu).view(np.int64)), bi], dtype=torch.int64)
            return bi, bu

        def sample_ensemble(self, y, initial_state, iterations, bounds, a=2.0, seed=0, store=True):
            StubGP.calls["sample_seeds"].append(int(seed))
            b = np.asarray(bounds, dtype=float)

            def logp(pts):
                ok = np.all((pts >= b[:, 0]) & (pts <= b[:, 1]), axis=1)
                mu = self.predict(y, pts, return_cov=False, return_var=False)
                return np.where(ok, mu, -np.inf)

            p0 = np.asarray(initial_state, dtype=float)
            s = mcmc.EnsembleSampler(len(p0), p0.shape[1], logp, vectorize=True, a=a, seed=seed)
            s.run_mcmc(p0, iterations)
            return {"chain": s.get_chain(), "log_prob": s.get_log_prob(), "naccept": s._naccepted,
                    "coords": s.get_chain()[-1], "final_log_prob": s.get_log_prob()[-1]}

        def box_candidates(self, m, bounds, seed, idx_offset=0):
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from philox_ref import philox_box_numpy
            b = np.asarray(bounds, dtype=float)
            StubGP.calls["box_rows"].append((int(m), int(idx_offset)))
            return torch.from_numpy(philox_box_numpy(int(m), len(b), b[:, 0], b[:, 1], int(seed), int(idx_offset)))

        def nll_batch(self, P, y):
            saved = self.get_parameter_vector()
            out = np.array([gpUtils._nll(np.array(p), self, y, None) for p in P])
            self.set_parameter_vector(saved)
            return out

    mod = types.SimpleNamespace(GP=StubGP, kernels=go.kernels, ExpSquaredKernel=go.ExpSquaredKernel)
    return mod


def _drive(out_path, rank):
    """The same script on every rank: C5's shape in miniature (run: optGP, m design points by the sharded sweep
    with a re-fit each, an on-device-style MCMC per iteration; then one more point by replicated Nelder-Mead)."""
    from approxposterior_amd import approx, likelihood as lh
    stub = _stub_module()
    approx.george = stub                       # what _absorbPoint builds the grown GP from
    np.random.seed(57 + 1000 * rank)           # DIFFERENT streams per rank: the wiring has to make them agree
    rs = np.random.RandomState(4)
    theta = rs.uniform(-5, 5, size=(24, 2))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    k = stub.ExpSquaredKernel(metric=np.array([3.0, 5.0]), ndim=2)
    gp = stub.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(theta)
    evaluations = []

    def lnlike(t, *a, **kw):
        evaluations.append(np.array(t))
        return lh.rosenbrockLnlike(t) + 1e-3 * np.random.randn()     # a NOISY forward model (draws on rank 0 only)

    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lnlike,
                                priorSample=lh.rosenbrockSample, bounds=BOUNDS, algorithm="alternate")
    runName = os.path.join(os.path.dirname(out_path), "ap_rank%d" % rank)
    with np.errstate(all="ignore"):
        ap.run(m=3, nmax=2, nCandidates=601, nGPRestarts=3, onDevice=True, verbose=False, cache=True,
               runName=runName, gpOptions={"maxiter": 3, "xtol": 1e-2, "ftol": 1e-2},
               samplerKwargs={"nwalkers": 6}, mcmcKwargs={"iterations": 30}, estBurnin=True, thinChains=True)
        chain = ap.sampler.get_chain()
        extra = ap.findNextPoint(nCandidates=None, nMinObjRestarts=2, cache=False, verbose=False,
                                 gpOptions={"maxiter": 2}, minObjOptions={"maxiter": 20})
        dev_pt = ap.findNextPoint(nCandidates=1003, deviceCandidates=True, computeLnLike=False, cache=False, verbose=False)
        best, val = ap.findMAP(nRestarts=2, options={"maxiter": 20, "adaptive": True})
    np.savez(out_path, theta=ap.theta, y=ap.y, p=ap.gp.get_parameter_vector(), chain=chain,
             iburns=ap.iburns, ithins=ap.ithins, nlnlike=len(evaluations), extra=np.asarray(extra[0]),
             best=best, val=val, rows=np.array(stub.GP.calls["acquire_rows"]), dev_pt=np.asarray(dev_pt),
             box_rows=np.array(stub.GP.calls["box_rows"]),
             seeds=np.array(stub.GP.calls["sample_seeds"]),
             wrote_cache=os.path.exists(runName + "APFModelCache.npz"),
             state=np.random.get_state()[1])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _drive(os.path.join(out_dir, "w%d_r%d.npz" % (world, rank)), rank)
    dist.destroy_process_group()


def _single(out_dir):
    sys.path.insert(0, ROOT)
    _drive(os.path.join(out_dir, "single.npz"), 0)


@pytest.mark.parametrize("world", [2, 3])
def test_approxposterior_run_is_rank_consistent(tmp_path, world):
    out = str(tmp_path)
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_single, args=(out,))
    proc.start(); proc.join()
    assert proc.exitcode == 0
    one = np.load(os.path.join(out, "single.npz"))
    ranks = [np.load(os.path.join(out, "w%d_r%d.npz" % (world, r))) for r in range(world)]
    n_new = 3 * 2 + 1
    assert one["theta"].shape == (24 + n_new, 2)
    for r, got in enumerate(ranks):
        # identical training set and hyper-parameters on every rank, equal to the single-process run's
        assert np.array_equal(got["theta"], ranks[0]["theta"]) and np.array_equal(got["y"], ranks[0]["y"])
        assert np.array_equal(got["p"], ranks[0]["p"])
        assert np.array_equal(got["theta"], one["theta"]) and np.array_equal(got["y"], one["y"])
        assert np.array_equal(got["p"], one["p"])
        assert np.array_equal(got["extra"], one["extra"])
        assert np.array_equal(got["best"], one["best"]) and got["val"] == one["val"]
        # the forward model ran on rank 0 only; only rank 0 wrote caches
        assert int(got["nlnlike"]) == (n_new if r == 0 else 0)
        assert bool(got["wrote_cache"]) == (r == 0)
        # every sweep saw this rank's shard of the 601-row draw (the last one: of the 1003 rows drawn "on the device")
        base, rem = divmod(601, world)
        assert set(got["rows"][:-1].tolist()) == {base + (1 if r < rem else 0)}
        b2, r2 = divmod(1003, world)
        lo2 = r * b2 + min(r, r2)
        assert got["rows"][-1] == b2 + (1 if r < r2 else 0)
        # deviceCandidates: this rank generated exactly its rows [lo, hi) of the global matrix, then the winning row alone
        assert got["box_rows"][0].tolist() == [b2 + (1 if r < r2 else 0), lo2] and got["box_rows"][1][0] == 1
        assert np.array_equal(got["dev_pt"], one["dev_pt"])
        # replica ensembles: world x 6 walkers, the same gathered chain everywhere; rank r sampled with base + r
        assert got["chain"].shape == (30, 6 * world, 2)
        assert np.array_equal(got["chain"], ranks[0]["chain"])
        assert np.array_equal(got["seeds"] - r, ranks[0]["seeds"])
        assert np.array_equal(got["iburns"], ranks[0]["iburns"]) and np.array_equal(got["ithins"], ranks[0]["ithins"])
    # rank 0's replica is the single-process chain (same seed, same surrogate)
    assert np.array_equal(ranks[0]["chain"][:, :6], one["chain"])
    assert one["rows"].tolist() == [601] * 6 + [1003]
    # the ranks' NumPy streams were pulled together before every draw that matters: after the last synchronised
    # call (findMAP) no rank has drawn anything rank 0 has not
    for got in ranks[1:]:
        assert np.array_equal(got["state"], ranks[0]["state"])


def _restart_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from approxposterior_amd import dist as apdist
    ran = []

    def run_mine(indices):
        ran.extend(indices)
        return [(-(i - 2.0) ** 2 if i != 4 else np.nan, np.array([i, 10.0 * i, -i])) for i in indices]

    mll, ps = apdist.spread_restarts(5, run_mine, 3)
    np.random.seed(rank)
    apdist.sync_random_state(0)
    draw = np.random.randn(3)
    got = apdist.broadcast_bytes(np.arange(4, dtype=np.int64) * (rank + 1), src=1)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), mll=mll, ps=ps, ran=np.array(ran), draw=draw, got=got)
    dist.destroy_process_group()


def test_spread_restarts_sync_and_broadcast_world2(tmp_path):
    out = str(tmp_path)
    mp.spawn(_restart_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = (np.load(os.path.join(out, "r%d.npz" % r)) for r in range(2))
    assert r0["ran"].tolist() == [0, 2, 4] and r1["ran"].tolist() == [1, 3]      # restart i on rank i % world
    for got in (r0, r1):
        assert np.array_equal(got["mll"][:4], -(np.arange(4) - 2.0) ** 2) and np.isnan(got["mll"][4])
        assert np.array_equal(got["ps"], np.array([[i, 10.0 * i, -i] for i in range(5)]))
        assert np.array_equal(got["got"], np.arange(4) * 2)                      # rank 1's bytes
    np.random.seed(0)
    assert np.array_equal(r0["draw"], np.random.randn(3)) and np.array_equal(r1["draw"], r0["draw"])
    # no process group: everything degenerates to the local call
    from approxposterior_amd import dist as apdist
    mll, ps = apdist.spread_restarts(2, lambda idx: [(1.0 * i, np.array([i])) for i in idx], 1)
    assert mll.tolist() == [0.0, 1.0] and ps.tolist() == [[0.0], [1.0]]
    assert apdist.context() is None and apdist.context(enabled=False) is None
    with pytest.raises(RuntimeError):
        apdist.context(enabled=True)


# ---------------------------------------------------------------------------------------------------------------
# distributed=False inside somebody else's process group: the ranks run DIFFERENT problems,
# a different number of collective-free steps each; nothing may touch the group -- no deadlock, no merged chains, no
# winner from the other rank's candidate matrix.
# ---------------------------------------------------------------------------------------------------------------

def _independent(out_path, rank):
    from approxposterior_amd import approx, likelihood as lh
    stub = _stub_module()
    approx.george = stub
    np.random.seed(11 + rank)
    rs = np.random.RandomState(40 + rank)                  # a different training set per rank
    theta = rs.uniform(-5, 5, size=(20 + 3 * rank, 2))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    k = stub.ExpSquaredKernel(metric=np.array([3.0, 5.0]), ndim=2)
    gp = stub.GP(kernel=k, fit_mean=True, mean=np.median(y), white_noise=-12, fit_white_noise=False)
    gp.compute(theta)
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lh.rosenbrockLnlike,
                                priorSample=lh.rosenbrockSample, bounds=BOUNDS, algorithm="agp", distributed=False)
    assert ap._ranks() is None
    pts = []
    with np.errstate(all="ignore"):
        for _ in range(1 + rank):                           # rank 1 does more (collective-free) steps than rank 0
            pts.append(ap.findNextPoint(nCandidates=307 + rank, deviceCandidates=True, computeLnLike=True, cache=False,
                                        verbose=False, nGPRestarts=2, gpOptions={"maxiter": 2})[0])
        ap.runMCMC(samplerKwargs={"nwalkers": 6}, mcmcKwargs={"iterations": 12 + rank}, onDevice=True, cache=False,
                   estBurnin=False, thinChains=False)
    np.savez(out_path, pts=np.array(pts), chain=ap.sampler.get_chain(), theta=ap.theta, p=ap.gp.get_parameter_vector(),
             rows=np.array(stub.GP.calls["acquire_rows"]), seeds=np.array(stub.GP.calls["sample_seeds"]))


def _independent_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _independent(os.path.join(out_dir, "g_r%d.npz" % rank), rank)
    dist.barrier()                                          # the group is intact: nobody left a half-done collective in it
    dist.destroy_process_group()


def _independent_alone(rank, out_dir):
    sys.path.insert(0, ROOT)
    _independent(os.path.join(out_dir, "a_r%d.npz" % rank), rank)


def test_distributed_false_never_touches_the_process_group(tmp_path):
    out = str(tmp_path)
    mp.spawn(_independent_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ctx = mp.get_context("spawn")
    for r in range(2):
        proc = ctx.Process(target=_independent_alone, args=(r, out))
        proc.start(); proc.join()
        assert proc.exitcode == 0
    for r in range(2):
        grouped, alone = (np.load(os.path.join(out, "%s_r%d.npz" % (tag, r))) for tag in ("g", "a"))
        for key in ("pts", "chain", "theta", "p", "rows", "seeds"):
            assert np.array_equal(grouped[key], alone[key]), (r, key)    # as if the other rank did not exist
        assert grouped["chain"].shape == (12 + r, 6, 2)                  # this rank's 6 walkers, not 12
        assert grouped["rows"].tolist() == [307 + r] * (1 + r)           # the whole matrix, not a shard


# ---------------------------------------------------------------------------------------------------------------
# rank-asymmetric work fails on one rank: every rank raises; nobody waits in a collective.
# ---------------------------------------------------------------------------------------------------------------

def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from approxposterior_amd import approx, dist as apdist, likelihood as lh
    seen = {}

    def run_mine(indices):
        if rank == 1:
            raise np.linalg.LinAlgError("restart blew up on rank 1")
        return [(0.0, np.zeros(2)) for _ in indices]
    try:
        apdist.spread_restarts(4, run_mine, 2)
    except Exception as err:      # noqa: BLE001
        seen["restarts"] = "%s: %s" % (type(err).__name__, err)

    stub = _stub_module()
    approx.george = stub
    rs = np.random.RandomState(4)
    theta = rs.uniform(-5, 5, size=(20, 2))
    y = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    gp = stub.GP(kernel=stub.ExpSquaredKernel(metric=np.array([3.0, 5.0]), ndim=2), fit_mean=True, mean=np.median(y),
                 white_noise=-12, fit_white_noise=False)
    gp.compute(theta)

    def lnlike(t, *a, **kw):
        raise FloatingPointError("the forward model diverged")
    ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lnlike,
                                priorSample=lh.rosenbrockSample, bounds=BOUNDS, algorithm="agp")
    try:
        with np.errstate(all="ignore"):
            ap.findNextPoint(nCandidates=200, cache=False, verbose=False)
    except Exception as err:      # noqa: BLE001
        seen["lnlike"] = "%s: %s" % (type(err).__name__, err)
    dist.barrier()
    with open(os.path.join(out_dir, "fail_r%d.txt" % rank), "w") as f:
        f.write(seen.get("restarts", "-") + "\n" + seen.get("lnlike", "-") + "\n")
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failure_on_one_rank_raises_on_every_rank(tmp_path):
    out = str(tmp_path)
    mp.spawn(_failing_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = (open(os.path.join(out, "fail_r%d.txt" % r)).read().splitlines() for r in range(2))
    assert r1[0].startswith("LinAlgError: restart blew up on rank 1")           # the failing rank: its own exception
    assert r0[0].startswith("RuntimeError: optimiser restarts failed on rank(s) [1]")
    assert r0[1].startswith("FloatingPointError: the forward model diverged")   # rank 0 ran the forward model
    assert r1[1].startswith("RuntimeError: the forward model (lnlike) failed on rank(s) [0]")
