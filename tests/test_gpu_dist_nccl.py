"""MI355X: the multi-GPU paths through the REAL collective library -- a world-size-1 ``nccl`` (= RCCL) process group
opened in-process on the one GPU the driver has.  What it pins: the sweep's int64[2] device
record goes HBM -> ``all_gather`` -> one D2H and equals ``GP.acquire`` bit for bit; ``replicated_ensembles`` over RCCL
equals the local chain; ``ApproxPosterior.run`` under the process group reproduces the run without it (world 1: the
shards, replicas and restarts are the whole job, so every bit must agree).
Reference: /root/reference/approxposterior/approx.py:396-424, :664-672, :839-856."""
import os
import socket

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.fixture(scope="module")
def rccl_world1():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _gp(n, d, seed):
    from approxposterior_amd import gp as agp
    rs = np.random.RandomState(seed)
    X = rs.uniform(-5, 5, size=(n, d))
    y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2, axis=1) / 100.0
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(d, 8.0), ndim=d), fit_mean=True, mean=np.median(y),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    return g, X, y


def test_device_record_through_rccl_equals_acquire(rccl_world1):
    import torch
    from approxposterior_amd import dist as apdist
    assert rccl_world1.get_backend() == "nccl" and apdist.context() == (0, 1)
    g, X, y = _gp(700, 8, 3)
    bounds = [(-5, 5)] * 8
    for kind, m, off in (("agp", 20000, 0), ("bape", 4097, 123456789), ("jones", 64, 7)):
        T = np.random.RandomState(m).uniform(-5.3, 5.3, size=(m, 8))
        want = g.acquire(y, T, kind, bounds=bounds, idx_offset=off)
        rec = []
        got = apdist.sharded_acquire(
            lambda o: g.acquire(y, T, kind, bounds=bounds, idx_offset=o, device_record=True), off, records=rec)
        assert got == want                                          # (index, utility): bit for bit
        assert rec == [(want[1], want[0])]
        # candidates already resident in HBM, record straight from the arg-min kernel's output
        Td = torch.from_numpy(T).cuda()
        raw = g.acquire(y, Td, kind, bounds=bounds, idx_offset=off, device_record=True)
        assert raw.is_cuda and raw.dtype == torch.int64
        assert apdist.sharded_acquire(lambda o: raw, off) == want
    # nothing admissible: index -1, +inf
    T = np.random.RandomState(0).uniform(6, 7, size=(300, 8))
    assert apdist.sharded_acquire(lambda o: g.acquire(y, T, "agp", bounds=bounds, idx_offset=o, device_record=True), 5) \
        == (-1, float("inf"))


def test_replicas_restarts_and_broadcast_through_rccl(rccl_world1):
    from approxposterior_amd import dist as apdist
    g, X, y = _gp(300, 4, 5)
    p0 = np.random.RandomState(1).uniform(-5, 5, size=(16, 4))
    bounds = [(-5, 5)] * 4
    local = g.sample_ensemble(y, p0, 200, bounds, seed=77)

    def sample(seed):
        r = g.sample_ensemble(y, p0, 200, bounds, seed=seed)
        return r["chain"], r["log_prob"], r["naccept"]
    chain, logp, nacc = apdist.replicated_ensembles(sample, seed=77)
    assert np.array_equal(chain, local["chain"]) and np.array_equal(logp, local["log_prob"])
    assert np.array_equal(nacc, local["naccept"])
    mll, ps = apdist.spread_restarts(3, lambda idx: [(-1.0 * i, np.array([i, -0.0, np.pi])) for i in idx], 3)
    assert mll.tolist() == [0.0, -1.0, -2.0] and np.array_equal(ps[:, 2], [np.pi] * 3) and np.signbit(ps[1, 1])
    np.random.seed(9)
    before = np.random.get_state()[1].copy()
    apdist.sync_random_state(0)
    assert np.array_equal(np.random.get_state()[1], before)
    b = apdist.broadcast_bytes(np.array([np.nan, -0.0, 1e-310]))
    assert np.isnan(b[0]) and np.signbit(b[1]) and b[2] == 1e-310     # bytes, not values, travel


@pytest.mark.parametrize("device_cands", [False, True])
def test_run_under_rccl_group_equals_run_without(rccl_world1, tmp_path, monkeypatch, device_cands):
    """C5's shape in miniature (D = 4, m0 = 96, m = 4, nmax = 2, 20,000 candidates, 3 restarts, on-device MCMC),
    once with ``distributed=False`` and once through the process group: identical training sets, hyper-parameters
    and chains -- with the candidates drawn by ``priorSample`` on the host and (``deviceCandidates``) by the counter-based
    generator on the device."""
    monkeypatch.chdir(tmp_path)
    from scipy.optimize import rosen
    from approxposterior_amd import approx, gpUtils
    D = 4
    lnprior = lambda t: 0.0 if np.all(np.abs(np.asarray(t)) <= 5) else -np.inf          # noqa: E731
    sample = lambda n=1: np.random.uniform(-5, 5, size=(n, D))                           # noqa: E731
    lnlike = lambda t, *a, **k: -rosen(np.asarray(t).ravel()) / 100.0                    # noqa: E731
    out = {}
    for mode in (False, None):
        np.random.seed(21)
        theta = sample(96)
        y = np.array([lnlike(t) + lnprior(t) for t in theta])
        gp = gpUtils.defaultGP(theta, y)
        ap = approx.ApproxPosterior(theta=theta, y=y, gp=gp, lnprior=lnprior, lnlike=lnlike, priorSample=sample,
                                    bounds=[(-5, 5)] * D, algorithm="alternate", distributed=mode)
        assert (ap._ranks() is None) == (mode is False)
        with np.errstate(all="ignore"):
            ap.run(m=4, nmax=2, nCandidates=20000, nGPRestarts=3, cache=False, verbose=False, onDevice=True,
                   gpOptions={"maxiter": 4}, mcmcKwargs={"iterations": 300}, samplerKwargs={"nwalkers": 16},
                   deviceCandidates=device_cands)
        out[mode] = (ap.theta.copy(), ap.y.copy(), ap.gp.get_parameter_vector().copy(), ap.sampler.get_chain().copy(),
                     ap.sampler.get_log_prob().copy(), np.random.get_state()[1].copy())
    for a, b in zip(out[False], out[None]):
        assert np.array_equal(a, b)
    assert out[None][0].shape == (96 + 8, D)


def test_run_loop_two_ranks_sharing_the_gpu_equals_one_rank(tmp_path):
    """Round 6: the product loop under a REAL multi-rank process group on the one GPU the driver has -- two processes,
    both on cuda:0, host collectives (``tools/run_c5_dist.py --backend gloo --share-device``): the sweep sharded by rank
    over the HIP kernels' device records, the restarts spread, one replica ensemble per rank, the forward model on rank 0.
    Both ranks must end with the same training set and hyper-parameters (``ranks_agree``), equal to the single-rank run's
    (``digest``), with twice the walkers in the gathered chain.  approx.py:396-424, :664-672, :839-856."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--points", "3", "--nmax", "2", "--m0", "200", "--candidates", "20000", "--iterations", "300", "--walkers", "16",
             "--restarts", "2", "--backend", "gloo", "--share-device"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(root, "tools", "run_c5_dist.py")] + small,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=800)
    assert two.returncode == 0, two.stderr[-3000:]
    rec2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    env1 = dict(env, MASTER_PORT=str(port + 1))
    one = subprocess.run([sys.executable, os.path.join(root, "tools", "run_c5_dist.py")] + small,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env1, timeout=800)
    assert one.returncode == 0, one.stderr[-3000:]
    rec1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec2["world"] == 2 and rec2["ranks_agree"] and rec1["world"] == 1
    assert rec2["n_train"] == rec1["n_train"] == 206
    assert rec2["digest"] == rec1["digest"]                       # same design points, values and hyper-parameters, every bit
    assert rec2["chain_walkers"] == 2 * rec1["chain_walkers"] == 32
