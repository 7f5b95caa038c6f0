"""MI355X, round 5: the fit path checked DIRECTLY against LAPACK / the oracle (not against another HIP path), the
invariants the in-place refactorisation relies on, the back-off after a persistent launch that gave up, and the
optimiser-loop memo.  Everything goes through the C ABI (``apgp_nll_eval``) or the ``george.GP``-shaped object over it.
Reference: /root/reference/approxposterior/gpUtils.py:46-80 (``_nll`` -> george ``GP.log_likelihood`` ->
``BasicSolver.compute`` = scipy ``cholesky`` + ``2 sum log diag``), doc/faq.rst:6-7."""
import ctypes
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import george_oracle as go
    from approxposterior_amd import gp as agp
    return go, agp


def _case(n, D, seed, metric=8.0):
    rs = np.random.RandomState(seed)
    X = rs.uniform(-5, 5, size=(n, D))
    y = -np.sum(100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1 - X[:, :-1]) ** 2, axis=1) / 100.0 \
        if D > 1 else np.sin(X[:, 0])
    return X, y


@pytest.mark.parametrize("n,D,mode", [(1152, 8, 0), (3000, 5, 0), (4096, 8, 0), (4096, 8, 1), (700, 2, 0), (64, 3, 0)])
def test_nll_eval_against_lapack_and_oracle(n, D, mode):
    """``apgp_nll_eval`` as shipped (mode 0: persistent launch up to n = 3200, hybrid above; mode 1: launch per step):
    the factor against ``scipy.linalg.cholesky`` of the oracle's Gram matrix, z against ``solve_triangular``, the
    log-determinant and z.z against the same, and the log-likelihood against the oracle GP.
    Tolerances (fp64, cond(K) ~ 1e3 at these hyper-parameters): |dL| <= 1e-11 max|L|, |dz| <= 1e-9 max|z|,
    logdet / z.z / ll to 1e-11 relative."""
    import torch
    from scipy.linalg import cholesky, solve_triangular
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, D, n + D)
    mean = float(np.median(y))
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=mean, white_noise=-12,
               fit_white_noise=False)
    g._x = X; g._yerr2 = 0.0
    ks = g._kernel_struct()
    X_d, y_d = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    K = torch.zeros((n, n), dtype=torch.float64, device="cuda")
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    info = torch.empty(1, dtype=torch.int32, device="cuda")
    o5 = torch.empty(5, dtype=torch.float64, device="cuda")
    o = np.empty(5)
    lib.apgp_potrf_mode(mode)
    try:
        fb = lib.apgp_potrf_fallbacks()
        rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), mean, K.data_ptr(), z.data_ptr(),
                               info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
        assert rc == 0, lib.apgp_last_error()
        torch.cuda.synchronize()
        assert lib.apgp_potrf_fallbacks() == fb
    finally:
        lib.apgp_potrf_mode(0)
    ko = go.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
    Ko = ko.get_value(X)
    Ko[np.diag_indices(n)] += np.exp(-12.0)
    Lo = cholesky(Ko, lower=True)
    zo = solve_triangular(Lo, y - mean, lower=True)
    Lh, zh = torch.tril(K).cpu().numpy(), z.cpu().numpy()
    assert int(info.item()) == 0 and o[4] == 0.0
    assert np.abs(Lh - Lo).max() <= 1e-11 * np.abs(Lo).max()
    assert np.abs(zh - zo).max() <= 1e-9 * np.abs(zo).max()
    logdet = 2.0 * np.sum(np.log(np.diag(Lo)))
    assert abs(o[0] - logdet) <= 1e-11 * abs(logdet)
    assert abs(o[3] - zo @ zo) <= 1e-11 * (zo @ zo)
    assert o[1] == pytest.approx(np.diag(Lo).min(), rel=1e-10) and o[2] == pytest.approx(np.diag(Lo).max(), rel=1e-10)
    gpo = go.GP(kernel=ko, fit_mean=True, mean=mean, white_noise=-12, fit_white_noise=False)
    gpo.compute(X)
    ll = -0.5 * (n * np.log(2 * np.pi) + o[0]) - 0.5 * o[3]
    assert abs(ll - gpo.log_likelihood(y)) <= 1e-11 * abs(ll)
    # the strict upper triangle of the work matrix is never written (LAPACK's dpotrf contract; GP._factor relies on it)
    assert float(torch.triu(K, 1).abs().max()) == 0.0


@pytest.mark.parametrize("n", [500, 1152, 3400])
def test_compute_and_recompute_take_the_persistent_plan(n):
    """``GP.compute`` / ``recompute`` (gpUtils.py:178,244,254; approx.py:717) run the same plan as an ``_nll``
    evaluation (round 4 sent them through a launch per 64-column step): same factor bits as the launch-per-step
    path, log-determinant and predictions equal to the oracle's."""
    import torch
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, 8, 3)

    def build():
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.compute(X)
        return g
    lib.apgp_potrf_mode(1)
    try:
        g1 = build()
        L1 = torch.tril(g1._L).clone()
    finally:
        lib.apgp_potrf_mode(0)
    fb = lib.apgp_potrf_fallbacks()
    g0 = build()
    assert lib.apgp_potrf_fallbacks() == fb
    assert torch.equal(torch.tril(g0._L), L1) and g0.log_determinant == g1.log_determinant
    p = g0.get_parameter_vector()
    p[1:] += 0.3
    g0.set_parameter_vector(p)
    assert not g0.computed
    g0.recompute()
    gpo = go.GP(kernel=go.ExpSquaredKernel(np.exp(p[1:]), ndim=8), fit_mean=True, mean=float(p[0]), white_noise=-12,
                fit_white_noise=False)
    gpo.compute(X)
    assert g0.computed and abs(g0.log_determinant - gpo.log_determinant) <= 1e-10 * abs(gpo.log_determinant)
    T = np.random.RandomState(1).uniform(-5, 5, size=(200, 8))
    mu, var = g0.predict(y, T, return_var=True)
    mo, vo = gpo.predict(y, T, return_var=True)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    assert np.abs(mu - mo).max() <= 1e-9 * asum and np.abs(var - vo).max() <= 1e-9


def test_reused_factor_buffer_keeps_a_clean_upper_triangle():
    """``GP._factor`` refactorises an optimiser's evaluations in place WITHOUT zeroing the buffer: sound only while no
    kernel ever writes above the diagonal.  300 evaluations at changing hyper-parameters on the persistent path, the
    hybrid, the launch-per-step path and through a forced give-up + fallback: the strict upper triangle stays exactly
    zero and every value equals a fresh-buffer evaluation."""
    import torch
    from approxposterior_amd import _lib, gpUtils
    go, agp = _mods()
    lib = _lib.load()
    rs = np.random.RandomState(5)
    for n, modes in ((1152, (0, 2, 1)), (3400, (0, 1))):
        X, y = _case(n, 8, n)
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.compute(X)
        p0 = g.get_parameter_vector()
        for mode in modes:
            lib.apgp_potrf_mode(mode)
            try:
                for it in range(100 if n == 1152 else 12):
                    p = p0 + np.concatenate([[0.0], rs.uniform(-0.5, 0.5, size=8)])
                    g._nllMemo = None
                    v = gpUtils._nll(p, g, y, None)
                    assert g._nll_owned and np.isfinite(v)
                    if it % 25 == 0:
                        assert float(torch.triu(g._L, 1).abs().max()) == 0.0
                        fresh = agp.GP(kernel=agp.ExpSquaredKernel(np.exp(p[1:]), ndim=8), fit_mean=True, mean=float(p[0]),
                                       white_noise=-12, fit_white_noise=False)
                        fresh._x = X
                        fresh._factor(y, upload_x=True)
                        assert -(fresh._const - 0.5 * fresh._ztz_host) == v
                assert float(torch.triu(g._L, 1).abs().max()) == 0.0
            finally:
                lib.apgp_potrf_mode(0)


def test_backoff_after_a_contended_persistent_launch():
    """a persistent launch that cannot get its workgroups resident costs the 50 ms timeout plus
    the re-run -- and did so on EVERY call while the contention lasted.  Now the give-up starts a back-off: the next
    evaluations go straight to the launch-per-step path.  Foreign kernels on a second stream hold most compute units
    while ``_nll`` is evaluated 200 times on the first: the average must stay under 2 ms -- not 50 -- and every value
    equals the uncontended one bit for bit."""
    import threading
    import torch
    from approxposterior_amd import _lib, gpUtils
    go, agp = _mods()
    lib = _lib.load()
    n = 1152
    X, y = _case(n, 8, 9)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    p0 = g.get_parameter_vector()
    ps = [p0 + np.concatenate([[0.0], np.random.RandomState(k).uniform(-0.3, 0.3, size=8)]) for k in range(200)]
    lib.apgp_potrf_mode(0)
    want = []
    for p in ps:
        g._nllMemo = None
        want.append(gpUtils._nll(p, g, y, None))
    # the foreign work: 220 long-running single-workgroup kernels (the on-device ensemble sampler, one workgroup per
    # ensemble, LDS-resident) on another stream -- 220 of the 256 compute units are taken, the persistent launch (18 row +
    # ~45 update workgroups, a whole CU's LDS each) cannot become resident in full, the launch-per-step path still runs
    Xb, yb = _case(600, 8, 2)
    hog = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(yb)),
                 white_noise=-12, fit_white_noise=False)
    side = torch.cuda.Stream()
    stop = threading.Event()

    def foreign():
        with torch.cuda.stream(side):
            hog.compute(Xb)
            p0 = np.random.RandomState(3).uniform(-5, 5, size=(220, 16, 8))
            while not stop.is_set():
                hog.sample_ensemble(yb, p0, 4000, [(-5, 5)] * 8, seed=1, store=False)
    th = threading.Thread(target=foreign)
    th.start()
    try:
        time.sleep(1.0)              # (the first sweep is running)
        fb, sk = lib.apgp_potrf_fallbacks(), lib.apgp_potrf_backoff_skips()
        t0 = time.perf_counter()
        got = []
        for p in ps:
            g._nllMemo = None
            got.append(gpUtils._nll(p, g, y, None))
        dt = (time.perf_counter() - t0) / len(ps)
    finally:
        stop.set()
        th.join()
        lib.apgp_potrf_mode(0)       # (ends the back-off for the tests that follow)
    assert got == want
    gave_up = lib.apgp_potrf_fallbacks() - fb
    skipped = lib.apgp_potrf_backoff_skips() - sk
    print("contended _nll: %.3f ms on average, %d give-ups, %d evaluations skipped the persistent launch" % (dt * 1e3, gave_up, skipped))
    assert dt < 2e-3
    if gave_up:
        assert skipped >= 64 and gave_up <= 4


@pytest.mark.parametrize("n,D,B", [(300, 3, 4), (1152, 8, 5), (1152, 8, 6), (1700, 4, 3), (2048, 8, 2), (3100, 2, 2)])
def test_side_by_side_batch_against_lapack_and_oracle(n, D, B):
    """``apgp_nll_eval_batch`` on its side-by-side path (round 6: ``potrf_persist_batch_kernel``, gridDim.y = B) DIRECTLY
    against LAPACK and the oracle -- not only against the single-evaluation HIP path it shares its code with: every
    matrix's factor against ``scipy.linalg.cholesky`` of the oracle's Gram matrix at THAT matrix's hyper-parameters, z
    against ``solve_triangular``, log-determinant / z.z / log-likelihood to 1e-11 (times the condition estimate over 1e3)."""
    import torch
    from scipy.linalg import cholesky, solve_triangular
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, D, n + B)
    rs = np.random.RandomState(B * n)
    metrics = [np.exp(rs.uniform(np.log(4.0), np.log(12.0), size=D)) for _ in range(B)]
    means = np.array([float(np.median(y)) + 0.1 * b for b in range(B)])
    karr = (_lib.KernelStruct * B)()
    for b in range(B):
        g = agp.GP(kernel=agp.ExpSquaredKernel(metrics[b], ndim=D), fit_mean=True, mean=means[b], white_noise=-12,
                   fit_white_noise=False)
        g._x = X; g._yerr2 = 0.0
        karr[b] = g._kernel_struct()
    X_d, y_d = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    K = torch.zeros((B, n, n), dtype=torch.float64, device="cuda")
    z = torch.empty((B, n), dtype=torch.float64, device="cuda")
    info = torch.empty(B, dtype=torch.int32, device="cuda")
    o5 = torch.empty((B, 5), dtype=torch.float64, device="cuda")
    o = np.empty((B, 5))
    lib.apgp_potrf_mode(0)
    sb, fb = lib.apgp_nll_side_batches(), lib.apgp_potrf_fallbacks()
    rc = lib.apgp_nll_eval_batch(X_d.data_ptr(), n, B, ctypes.addressof(karr), y_d.data_ptr(), means.ctypes.data, K.data_ptr(),
                                 z.data_ptr(), info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
    assert rc == 0, lib.apgp_last_error()
    torch.cuda.synchronize()
    assert lib.apgp_nll_side_batches() == sb + 1 and lib.apgp_potrf_fallbacks() == fb      # served side by side, nobody gave up
    for b in range(B):
        ko = go.ExpSquaredKernel(metrics[b], ndim=D)
        Ko = ko.get_value(X)
        Ko[np.diag_indices(n)] += np.exp(-12.0)
        Lo = cholesky(Ko, lower=True)
        zo = solve_triangular(Lo, y - means[b], lower=True)
        Lh, zh = torch.tril(K[b]).cpu().numpy(), z[b].cpu().numpy()
        assert int(info[b].item()) == 0 and o[b, 4] == 0.0
        assert np.abs(Lh - Lo).max() <= 1e-11 * np.abs(Lo).max()
        assert np.abs(zh - zo).max() <= 1e-9 * np.abs(zo).max()
        logdet = 2.0 * np.sum(np.log(np.diag(Lo)))
        # (random metrics 4 .. 12: at D = 2 the 3,100 points lie dense and cond(K) is ~1e4 times the D = 8 cases')
        tol = 1e-11 * max(1.0, (o[b, 2] / o[b, 1]) ** 2 / 1e3)
        assert abs(o[b, 0] - logdet) <= tol * abs(logdet)
        assert abs(o[b, 3] - zo @ zo) <= tol * (zo @ zo)
        gpo = go.GP(kernel=ko, fit_mean=True, mean=means[b], white_noise=-12, fit_white_noise=False)
        gpo.compute(X)
        ll = -0.5 * (n * np.log(2.0 * np.pi) + o[b, 0]) - 0.5 * o[b, 3]
        assert abs(ll - gpo.log_likelihood(y)) <= tol * abs(gpo.log_likelihood(y))


def test_side_by_side_batch_under_contention_equals_single_evaluations():
    """Round 6: ``apgp_nll_eval_batch`` runs 2 .. 6 persistent factorisations side by side in one launch; when foreign
    kernels hold most compute units its workgroups cannot all be resident, a matrix gives up, and the whole batch is
    redone on the batched launch-per-step path (then the back-off of the single call applies).  Whatever path served it,
    every value equals the single evaluation's, bit for bit."""
    import threading
    import torch
    from approxposterior_amd import _lib, gpUtils
    go, agp = _mods()
    lib = _lib.load()
    n = 900
    X, y = _case(n, 8, 19)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    g.lookahead = 0
    p0 = g.get_parameter_vector()
    P = np.array([p0 + np.concatenate([[0.0], np.random.RandomState(k).uniform(-0.3, 0.3, size=8)]) for k in range(5)])
    lib.apgp_potrf_mode(0)
    want = []
    for p in P:
        g._nllMemo = None
        want.append(gpUtils._nll(p, g, y, None))
    want = np.array(want)
    assert np.array_equal(g.nll_batch(P, y), want)
    Xb, yb = _case(600, 8, 2)
    hog = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(yb)),
                 white_noise=-12, fit_white_noise=False)
    side = torch.cuda.Stream()
    stop = threading.Event()

    def foreign():
        with torch.cuda.stream(side):
            hog.compute(Xb)
            q0 = np.random.RandomState(3).uniform(-5, 5, size=(220, 16, 8))
            while not stop.is_set():
                hog.sample_ensemble(yb, q0, 4000, [(-5, 5)] * 8, seed=1, store=False)
    th = threading.Thread(target=foreign)
    th.start()
    try:
        time.sleep(1.0)
        fb, sb, sk = lib.apgp_potrf_fallbacks(), lib.apgp_nll_side_batches(), lib.apgp_potrf_backoff_skips()
        t0 = time.perf_counter()
        for rep in range(120):
            got = g.nll_batch(P[:2 + rep % 4], y)
            assert np.array_equal(got, want[:2 + rep % 4]), rep
        dt = (time.perf_counter() - t0) / 120
    finally:
        stop.set()
        th.join()
        lib.apgp_potrf_mode(0)
    print("contended batches: %.3f ms on average, %d served side by side, %d gave up, %d skipped the persistent launch"
          % (dt * 1e3, lib.apgp_nll_side_batches() - sb, lib.apgp_potrf_fallbacks() - fb, lib.apgp_potrf_backoff_skips() - sk))
    assert dt < 4e-3


def test_multi_workgroup_sampler_gives_up_cleanly_under_contention():
    """a workgroup of the multi-workgroup sampler that times out used to mark the failure with NaN in
    ``logp`` only, which the ensemble's workgroup 0 could overwrite a moment later.  Now it raises a sticky word of the launch
    and a one-workgroup kernel behind it turns that into NaN for every log-probability; ``GP.sample_ensemble`` re-runs on the
    single-workgroup kernel, chosen for that call only (``apgp_ensemble_sample_ex``: no process-wide switch is flipped).
    Foreign single-workgroup kernels hold 240 of the 256 compute units while one ensemble of 64 walkers asks for 32: the
    result must be the single-workgroup kernel's, complete and finite, and the process-wide mode untouched."""
    import threading
    import torch
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(700, 8, 5)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    p0 = np.random.RandomState(8).uniform(-5, 5, size=(1, 64, 8))
    bounds = [(-5, 5)] * 8
    prev = lib.apgp_ensemble_mode(1)
    try:
        single = g.sample_ensemble(y, p0, 300, bounds, seed=5)
    finally:
        lib.apgp_ensemble_mode(prev)
    Xb, yb = _case(600, 8, 2)
    hog = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(yb)),
                 white_noise=-12, fit_white_noise=False)
    side = torch.cuda.Stream()
    stop = threading.Event()

    def foreign():
        with torch.cuda.stream(side):
            hog.compute(Xb)
            q0 = np.random.RandomState(3).uniform(-5, 5, size=(240, 16, 8))
            while not stop.is_set():
                hog.sample_ensemble(yb, q0, 4000, bounds, seed=1, store=False)
    th = threading.Thread(target=foreign)
    th.start()
    try:
        time.sleep(1.0)
        runs = []
        for attempt in range(40):        # (a call may fall into the gap between two foreign launches: until one gives up)
            before = getattr(g, "ensemble_fallbacks", 0)
            t0 = time.perf_counter()
            got = g.sample_ensemble(y, p0, 300, bounds, seed=5)
            dt = time.perf_counter() - t0
            runs.append((dt, got, getattr(g, "ensemble_fallbacks", 0) > before))
            if runs[-1][2]:
                break
    finally:
        stop.set()
        th.join()
    assert lib.apgp_ensemble_mode(-1) == 0                               # nobody flipped the process-wide switch
    gave_up = sum(1 for _, _, fell_back in runs if fell_back)
    print("contended multi-workgroup sampler: %d calls, %d gave up and re-ran on one workgroup (%s ms)"
          % (len(runs), gave_up, ", ".join("%.1f" % (dt * 1e3) for dt, _, _ in runs[-3:])))
    for dt, got, fell_back in runs:
        for key in ("chain", "log_prob", "coords", "final_log_prob"):
            assert not np.any(np.isnan(got[key])), key                    # complete: no row left unwritten, no marker left
        if fell_back:
            for key in ("chain", "log_prob", "coords", "final_log_prob", "naccept"):
                assert np.array_equal(got[key], single[key]), key         # the single-workgroup kernel's result, every bit
        else:
            assert np.allclose(got["chain"], single["chain"], rtol=1e-9, atol=1e-9)
            assert np.array_equal(got["naccept"], single["naccept"])


def test_nll_memo_answers_exact_repeats_only():
    """gpUtils._nll keeps the last 64 evaluated points of THIS training set and y (Powell re-asks f at the head of every
    line search): an exact repeat costs no device work and returns the same float; a different y, a new training set or a
    changed fixed white noise never hit."""
    from approxposterior_amd import gpUtils, _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(300, 4, 1)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(4, 8.0), ndim=4), fit_mean=True, mean=float(np.median(y)),
               white_noise=-12, fit_white_noise=False)
    g.compute(X)
    p = g.get_parameter_vector() + 0.1
    calls = []
    real = g.log_likelihood
    g.log_likelihood = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    v1 = gpUtils._nll(p, g, y, gpUtils.defaultHyperPrior)
    v2 = gpUtils._nll(p.copy(), g, y, gpUtils.defaultHyperPrior)
    assert v1 == v2 and len(calls) == 1
    assert np.array_equal(g.get_parameter_vector(), p)             # (the hit still left p set, as george would)
    gpUtils._nll(p + 1e-16 * np.abs(p).max(), g, y, None)
    v3 = gpUtils._nll(p, g, y + 1.0, None)                         # another y: new table
    assert len(calls) == 3 and v3 != v1
    g.white_noise.value = -10.0
    v4 = gpUtils._nll(p, g, y + 1.0, None)
    assert len(calls) == 4 and v4 != v3
    g.compute(X[:250])
    assert g._nllMemo is None
    # Powell at C5's first size: the memo changes nothing but the number of device evaluations
    from scipy.optimize import minimize
    X, y = _case(512, 8, 2)
    res = []
    for use in (True, False):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(8, 8.0), ndim=8), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.compute(X)
        if not use:
            del g._nllMemo                                          # (an object without the slot is never memoised)
        n_dev = []
        real = g.log_likelihood
        g.log_likelihood = lambda *a, _r=real, **k: (n_dev.append(1), _r(*a, **k))[1]
        with np.errstate(all="ignore"):
            sol = minimize(gpUtils._nll, g.get_parameter_vector(), args=(g, y, gpUtils.defaultHyperPrior),
                           method="powell", options={"maxiter": 3})
        res.append((sol["x"], sol["fun"], sol["nfev"], len(n_dev)))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert res[0][3] < res[1][3]
    print("Powell, 3 iterations at N = 512: %d objective calls, %d device evaluations with the memo" % (res[0][2], res[0][3]))


@pytest.mark.parametrize("n,D,W,E", [(1152, 8, 64, 1), (300, 2, 8, 3), (700, 5, 40, 8), (2500, 8, 64, 2)])
def test_multi_workgroup_sampler_against_oracle_and_single_workgroup(n, D, W, E):
    """Round 5: one ensemble over several workgroups (csrc/ensemble.hip, ensemble_mw_kernel; BASELINE config 5's MCMC --
    approx.py:839-856 -- was one workgroup on one of 256 compute units).  (1) every stored log-probability is the oracle's GP
    mean at the stored coordinates (|d| <= 1e-9 sum|alpha|), (2) the chain equals the single-workgroup kernel's -- same
    counter-based RNG streams and proposals; the GP means are summed in another order, so to rounding (1e-9), not bit for
    bit -- including accepted-move counts, (3) the training stream outside LDS (n = 2500: 200 KB) takes the same path."""
    import torch
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, D, n + W)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-12,
               fit_white_noise=False)
    g.compute(X)
    gpo = go.GP(kernel=go.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-12,
                fit_white_noise=False)
    gpo.compute(X)
    p0 = np.random.RandomState(2).uniform(-5, 5, size=(E, W, D))
    p0[0, 0] = 7.0                                   # a walker that starts outside the prior: -inf until it moves
    bounds = [(-5, 5)] * D
    iters = 400
    assert lib.apgp_ensemble_mode(-1) == 0
    t0 = time.perf_counter()
    multi = g.sample_ensemble(y, p0, iters, bounds, seed=99)
    t_multi = time.perf_counter() - t0
    prev = lib.apgp_ensemble_mode(1)
    try:
        t0 = time.perf_counter()
        single = g.sample_ensemble(y, p0, iters, bounds, seed=99)
        t_single = time.perf_counter() - t0
    finally:
        lib.apgp_ensemble_mode(prev)
    print("n = %d, %d ensemble(s) x %d walkers x %d iterations: %.1f ms on several workgroups, %.1f ms on one"
          % (n, E, W, iters, t_multi * 1e3, t_single * 1e3))
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    for it in (0, iters // 2, iters - 1):
        want = gpo.predict(y, multi["chain"][it], return_cov=False)
        lp = multi["log_prob"][it]
        inside = np.all(np.abs(multi["chain"][it]) <= 5, axis=1)
        assert np.all(np.isneginf(lp[~inside])) and np.abs(lp[inside] - want[inside]).max() <= 1e-9 * asum
    assert np.allclose(multi["chain"], single["chain"], rtol=1e-9, atol=1e-9)
    fin = np.isfinite(single["log_prob"])
    assert np.array_equal(fin, np.isfinite(multi["log_prob"]))
    assert np.allclose(multi["log_prob"][fin], single["log_prob"][fin], rtol=1e-9, atol=1e-9 * asum)
    assert np.array_equal(multi["naccept"], single["naccept"])
    assert np.allclose(multi["coords"], single["coords"], rtol=1e-9, atol=1e-9)
    assert 0.05 < multi["naccept"].sum() / (iters * E * W) < 0.95


@pytest.mark.parametrize("n,D", [(300, 24), (700, 17), (64, 32)])
def test_more_than_sixteen_dimensions(n, D):
    """VERDICT round 4, missing 3: the reference (through george) has no dimension limit (gpUtils.py:150-161); round 4 raised
    above D = 16.  Every kernel that evaluates k(x, x') is now also instantiated for D padded to 32: log-likelihood, gradient,
    predictions with variance (batched, single candidate, mean only), the fused acquisition and the on-device sampler at
    D = 17 / 24 / 32 against the oracle.  D = 33 still raises."""
    go, agp = _mods()
    rs = np.random.RandomState(D)
    X = rs.uniform(-2, 2, size=(n, D))
    y = np.sin(X[:, 0]) + 0.5 * np.cos(X[:, 1:]).sum(axis=1) + 0.01 * rs.normal(size=n)
    metric = rs.uniform(4.0, 12.0, size=D) * D / 8.0
    mean = float(np.median(y))
    g = agp.GP(kernel=2.5 * agp.ExpSquaredKernel(metric, ndim=D), fit_mean=True, mean=mean, white_noise=-8, fit_white_noise=False)
    o = go.GP(kernel=2.5 * go.ExpSquaredKernel(metric, ndim=D), fit_mean=True, mean=mean, white_noise=-8, fit_white_noise=False)
    g.compute(X); o.compute(X)
    assert len(g.get_parameter_vector()) == D + 2 and g.get_parameter_names() == o.get_parameter_names()
    llo = o.log_likelihood(y)
    assert abs(g.log_likelihood(y) - llo) <= 1e-10 * abs(llo)
    p = g.get_parameter_vector() + 0.05
    g.set_parameter_vector(p); o.set_parameter_vector(p)
    llo = o.log_likelihood(y)
    assert abs(g.log_likelihood(y) - llo) <= 1e-10 * abs(llo)          # (the _nll path: factorisation with y riding along)
    gg, og = g.grad_log_likelihood(y), o.grad_log_likelihood(y)
    assert np.abs(gg - og).max() <= 1e-7 * max(1.0, np.abs(og).max())
    T = rs.uniform(-2.2, 2.2, size=(1500, D))
    mo, vo = o.predict(y, T, return_var=True)
    mu, var = g.predict(y, T, return_var=True)
    asum = np.abs(o._compute_alpha(y, False)).sum()
    assert np.abs(mu - mo).max() <= 1e-9 * asum and np.abs(var - vo).max() <= 1e-9 * 2.5
    m1, v1 = g.predict(y, T[7:8], return_var=True)                     # the scalar utilities' single-candidate path
    assert abs(m1[0] - mo[7]) <= 1e-9 * asum and abs(v1[0] - vo[7]) <= 1e-9 * 2.5
    assert np.abs(g.predict(y, T[:40], return_cov=False) - mo[:40]).max() <= 1e-9 * asum
    bounds = [(-2, 2)] * D
    bi, bu, u, mu2, var2 = g.acquire(y, T, "bape", bounds=bounds, return_all=True)
    with np.errstate(all="ignore"):
        uo = -((2.0 * mo + vo) + vo + np.log(1.0 - np.exp(-vo)))
    uo = np.where(np.all(np.abs(T) <= 2, axis=1), uo, np.inf)
    fin = np.isfinite(uo)
    assert np.array_equal(np.isfinite(u), fin) and np.abs(u[fin] - uo[fin]).max() <= 1e-7 * np.abs(uo[fin]).max()
    assert bi == int(np.argmin(uo)) or abs(uo[bi] - uo.min()) <= 1e-9 * abs(uo.min())
    if n > 64:
        W = 2 * D + 2
        res = g.sample_ensemble(y, rs.uniform(-2, 2, size=(W, D)), 60, bounds, seed=3)
        want = o.predict(y, res["chain"][-1], return_cov=False)
        assert np.abs(res["log_prob"][-1] - want).max() <= 1e-9 * asum
    if D == 32:
        with pytest.raises(ValueError):
            agp.GP(kernel=agp.ExpSquaredKernel(np.ones(33), ndim=33), fit_mean=True, mean=0.0, white_noise=-8,
                   fit_white_noise=False).compute(np.zeros((40, 33)))


from philox_ref import philox_box_numpy as _philox_box_numpy  # noqa: E402


def test_device_candidates_are_a_function_of_seed_and_row():
    """``deviceCandidates`` (round 5): the candidate matrix of the sweep drawn on the device by counter-based Philox -- the
    batched counterpart of the ``sampleFn`` draws utility.minimizeObjective starts from (utility.py:334-338) when the prior is
    the box.  (1) the matrix equals a NumPy restatement of the generator (to 1 ulp: the kernel fuses lo + span * u),
    (2) rows generated with an offset are exactly the rows of the whole matrix -- what lets every rank of a sharded sweep
    generate its own shard, (3) uniform in the box, (4) ``findNextPoint(nCandidates=..., deviceCandidates=True)`` picks the
    arg-min of the oracle's utility over that matrix."""
    from approxposterior_amd import approx, gpUtils, likelihood as lh, utility as ut
    go, agp = _mods()
    D = 5
    X, y = _case(200, D, 4)
    g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-12,
               fit_white_noise=False)
    g.compute(X)
    bounds = [(-5.0, 5.0), (-1.0, 3.0), (0.0, 0.5), (-5.0, 5.0), (2.0, 2.0 + 1e-3)]
    lo, hi = np.array([b[0] for b in bounds]), np.array([b[1] for b in bounds])
    seed = 123456789123
    T = g.box_candidates(100003, bounds, seed).cpu().numpy()
    want = _philox_box_numpy(100003, D, lo, hi, seed, 0)
    assert np.abs(T - want).max() <= 4e-16 * np.abs(want).max()
    part = g.box_candidates(777, bounds, seed, idx_offset=50000).cpu().numpy()
    assert np.array_equal(part, T[50000:50777])
    big = g.box_candidates(3, bounds, seed, idx_offset=2 ** 33 + 5).cpu().numpy()
    assert np.abs(big - _philox_box_numpy(3, D, lo, hi, seed, 2 ** 33 + 5)).max() <= 4e-16 * 5
    assert np.all(T >= lo) and np.all(T <= hi)
    u = (T - lo) / (hi - lo)
    assert np.abs(u.mean(axis=0) - 0.5).max() < 0.01 and np.abs(u.var(axis=0) - 1.0 / 12).max() < 0.005
    assert g.box_candidates(0, bounds, seed).shape == (0, D)
    # through ApproxPosterior: the winner is the oracle's arg-min over the generated matrix
    D = 2
    np.random.seed(57)
    theta = np.array(lh.rosenbrockSample(60))
    yy = np.array([lh.rosenbrockLnlike(t) + lh.rosenbrockLnprior(t) for t in theta])
    gp = gpUtils.defaultGP(theta, yy)
    ap = approx.ApproxPosterior(theta=theta, y=yy, gp=gp, lnprior=lh.rosenbrockLnprior, lnlike=lh.rosenbrockLnlike,
                                priorSample=lh.rosenbrockSample, bounds=((-5, 5), (-5, 5)), algorithm="bape")
    state = np.random.get_state()
    pt = ap.findNextPoint(computeLnLike=False, nCandidates=50000, deviceCandidates=True)
    np.random.set_state(state)
    s2 = int(np.random.randint(0, 2 ** 31 - 1))
    M = _philox_box_numpy(50000, 2, np.array([-5.0, -5.0]), np.array([5.0, 5.0]), s2, 0)
    o = go.GP(kernel=go.ExpSquaredKernel(np.exp(gp.get_parameter_vector()[1:]), ndim=2), fit_mean=True,
              mean=float(gp.get_parameter_vector()[0]), white_noise=-12, fit_white_noise=False)
    o.compute(theta)
    mo, vo = o.predict(yy, M, return_var=True)
    with np.errstate(all="ignore"):
        uo = -((2.0 * mo + vo) + vo + np.log(1.0 - np.exp(-vo)))
    best = int(np.nanargmin(uo))
    assert np.abs(pt - M[best]).max() <= 1e-12 or abs(uo[best] - np.sort(uo[np.isfinite(uo)])[0]) < 1e-9
    assert ap.deviceCandidates is True


def test_repeated_evaluation_plan_equals_the_generic_path():
    """``GP._factor_again`` (the optimiser loop's short path: same training set, y, stream -> the previous call's
    argument list with the kernel struct refilled) against the generic ``_factor``: same values, same factor bits, same
    object state afterwards; anything that differs (y's contents, the current stream, a failed factorisation, an
    intervening prediction) falls back or recovers as the generic path does."""
    import torch
    from approxposterior_amd import gpUtils
    go, agp = _mods()
    n, D = 700, 5
    X, y = _case(n, D, 9)
    rs = np.random.RandomState(0)

    def make():
        g = gpUtils.defaultGP(X, y, fitAmp=True)
        g.compute(X)
        return g
    ga, gb = make(), make()
    p0 = ga.get_parameter_vector()
    T = rs.uniform(-5, 5, size=(40, D))
    used = 0
    for it in range(12):
        p = p0 + rs.normal(0, 0.2, size=p0.shape)
        ga._nllMemo = gb._nllMemo = None
        used += ga._nll_plan is not None
        gb._nll_plan = None                                   # generic path every time
        a, b = gpUtils._nll(p, ga, y, None), gpUtils._nll(p, gb, y, None)
        assert a == b and np.isfinite(a)
        assert ga.computed and gb.computed
        assert torch.equal(torch.tril(ga._L), torch.tril(gb._L)) and torch.equal(ga._z, gb._z)
        for f in ("log_determinant", "cond_estimate", "_const", "_ztz_host", "_alpha_mean", "_factored_key"):
            assert getattr(ga, f) == getattr(gb, f), f
        if it % 4 == 3:
            # consumers of the factor in between (alpha, L^-1, packed operands): dropped again by the next evaluation
            ma, va = ga.predict(y, T, return_var=True)
            mb, vb = gb.predict(y, T, return_var=True)
            assert np.array_equal(ma, mb) and np.array_equal(va, vb)
    assert used >= 10
    # y with other contents (same object, mutated in place): not the plan's y any more
    y2 = y.copy()
    gpUtils._nll(p0, ga, y2, None)
    assert ga._nll_plan is not None
    y2[3] += 1.0
    ga._nllMemo = gb._nllMemo = None
    gb._nll_plan = None
    assert gpUtils._nll(p0 + 0.01, ga, y2, None) == gpUtils._nll(p0 + 0.01, gb, y2, None)
    # another current stream: fresh buffers, as the generic path
    K_before = ga._L
    with torch.cuda.stream(torch.cuda.Stream()):
        v = gpUtils._nll(p0 + 0.02, ga, y2, None)
        torch.cuda.current_stream().synchronize()
    assert ga._L is not K_before
    gb._nll_plan = None
    assert v == gpUtils._nll(p0 + 0.02, gb, y2, None)
    # a Gram matrix that is not positive definite on the short path: +inf, reset state, and the next evaluation is fine
    gpUtils._nll(p0, ga, y, None)
    assert ga._nll_plan is not None
    bad = p0.copy(); bad[2:] = 19.0; bad[1] = 19.5            # (huge length scales and amplitude: rank-deficient in fp64)
    vb_ = gpUtils._nll(bad, ga, y, None)
    gb._nll_plan = None
    assert vb_ == gpUtils._nll(bad, gb, y, None)
    if not np.isfinite(vb_):
        assert not ga.computed and ga._L is None
    ga._nllMemo = gb._nllMemo = None
    gb._nll_plan = None
    assert gpUtils._nll(p0, ga, y, None) == gpUtils._nll(p0, gb, y, None)


def test_compute_owns_its_training_set():
    """``GP.compute(x)`` keeps a COPY of x (george: the object owns its training set): editing the caller's array in
    place and computing again must be seen as a new training set (the device copy of x is re-uploaded), and edits
    without a compute must not leak into later evaluations."""
    go, agp = _mods()
    n, D = 300, 3
    X, y = _case(n, D, 21)

    def fresh(Xc):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.compute(Xc)
        return g
    Xu = np.ascontiguousarray(X.copy())
    g = fresh(Xu)
    ll0 = g.log_likelihood(y)
    Xu[5] += 0.25                                   # in place, no compute: the GP still describes the old set
    p = g.get_parameter_vector()
    g.set_parameter_vector(p)                        # (dirty: the next log_likelihood refactorises from the GP's own x)
    assert abs(g.log_likelihood(y) - ll0) <= 1e-12 * abs(ll0)    # (z rides along the factorisation here: last bits may differ)
    g.compute(Xu)                                    # the same object, new contents
    assert g.log_likelihood(y) == fresh(Xu.copy()).log_likelihood(y) != ll0


@pytest.mark.parametrize("n,D", [(65, 2), (66, 3), (90, 2), (100, 8), (127, 5), (128, 16), (97, 24), (128, 1)])
def test_two_block_column_fused_evaluation(n, D):
    """64 < n <= 128 (the README example's sizes, N = 50 -> 90): ``apgp_nll_eval`` is ONE single-workgroup launch (Gram tiles in
    LDS, both block columns by the panel step's code, the update by the launch-per-step product, the summary in the finish
    kernel's order).  Bit-identical to the separate launches (mode 1): factor, z, record, LAPACK info -- also for a matrix that
    is not positive definite in either block column -- and equal to LAPACK / the oracle within fp64 tolerances."""
    import torch
    from scipy.linalg import cholesky, solve_triangular
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, D, 7 * n + D)
    mean = float(np.median(y))

    def run(mode, Xc, wn=-12.0):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 8.0), ndim=D), fit_mean=True, mean=mean, white_noise=wn,
                   fit_white_noise=False)
        g._x = Xc; g._yerr2 = 0.0
        ks = g._kernel_struct()
        X_d, y_d = torch.from_numpy(Xc).cuda(), torch.from_numpy(y).cuda()
        K = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        z = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
        info = torch.empty(1, dtype=torch.int32, device="cuda")
        o5 = torch.empty(5, dtype=torch.float64, device="cuda")
        o = np.full(5, np.nan)
        lib.apgp_potrf_mode(mode)
        try:
            for _ in range(2):                   # (twice in the same buffers: nothing may leak from one call into the next)
                rc = lib.apgp_nll_eval(X_d.data_ptr(), n, ctypes.byref(ks), y_d.data_ptr(), mean, K.data_ptr(), z.data_ptr(),
                                       info.data_ptr(), o5.data_ptr(), o.ctypes.data, None)
                assert rc == 0, lib.apgp_last_error()
            torch.cuda.synchronize()
        finally:
            lib.apgp_potrf_mode(0)
        assert o.tobytes() == o5.cpu().numpy().tobytes()
        return K.clone(), z.clone(), o, int(info.item())
    K0, z0, o0, i0 = run(0, X)
    K1, z1, o1, i1 = run(1, X)
    assert i0 == i1 == 0 and o0.tobytes() == o1.tobytes()
    assert torch.equal(torch.tril(K0), torch.tril(K1)) and torch.equal(z0, z1)
    assert float(torch.triu(K0, 1).abs().max()) == 0.0
    ko = go.ExpSquaredKernel(np.full(D, 8.0), ndim=D)
    Ko = ko.get_value(X)
    Ko[np.diag_indices(n)] += np.exp(-12.0)
    Lo = cholesky(Ko, lower=True)
    zo = solve_triangular(Lo, y - mean, lower=True)
    assert np.abs(torch.tril(K0).cpu().numpy() - Lo).max() <= 1e-11 * np.abs(Lo).max()
    assert np.abs(z0.cpu().numpy() - zo).max() <= 1e-9 * np.abs(zo).max()
    assert abs(o0[0] - 2.0 * np.sum(np.log(np.diag(Lo)))) <= 1e-11 * abs(o0[0]) and abs(o0[3] - zo @ zo) <= 1e-11 * (zo @ zo)
    # (nearly) singular: a tripled point in block column 0, then in block column 1 (no white noise to speak of)
    for dup in (10, n - 2):
        Xd = X.copy(); Xd[dup] = Xd[dup - 1]; Xd[dup + 1] = Xd[dup - 1]
        a0, a1 = run(0, Xd, wn=-60.0), run(1, Xd, wn=-60.0)
        assert a0[3] == a1[3] and a0[2][4] == a1[2][4] == a0[3]        # (the same LAPACK info, whatever rounding makes of it)
        assert a0[2].tobytes() == a1[2].tobytes()


def test_repeated_mean_prediction_plan_equals_the_generic_path():
    """``GP.predict(y, t, return_cov=False, return_var=False)`` for a few points at a time (the walker ensembles of the host-loop
    sampler) re-uses the previous call's arguments (``_predict_mean_again``): same values as the generic path, and anything
    that changes the model, y, the mean or the stream goes back to it."""
    import torch
    go, agp = _mods()
    n, D = 90, 2
    X, y = _case(n, D, 13)
    rs = np.random.RandomState(2)

    def make():
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 3.0), ndim=D), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.compute(X)
        return g
    ga, gb = make(), make()
    used = 0
    for it in range(8):
        T = rs.uniform(-5, 5, size=(10 if it % 2 else 7, D))
        used += ga._mean_plan is not None
        gb._mean_plan = None
        assert np.array_equal(ga.predict(y, T, return_cov=False, return_var=False),
                              gb.predict(y, T, return_cov=False, return_var=False))
    assert used >= 7
    gpo = go.GP(kernel=go.ExpSquaredKernel(np.full(D, 3.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-12,
                fit_white_noise=False)
    gpo.compute(X)
    T = rs.uniform(-5, 5, size=(10, D))
    mo = gpo.predict(y, T, return_cov=False, return_var=False)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    assert ga._mean_plan is not None and np.abs(ga.predict(y, T, return_cov=False, return_var=False) - mo).max() <= 1e-10 * asum
    # another y: not the plan's
    y2 = y + 0.5
    gb._mean_plan = None
    assert np.array_equal(ga.predict(y2, T, return_cov=False, return_var=False), gb.predict(y2, T, return_cov=False, return_var=False))
    # new hyper-parameters: the model is dirty, the generic path refactorises
    p = ga.get_parameter_vector() + 0.1
    ga.set_parameter_vector(p); gb.set_parameter_vector(p)
    gb._mean_plan = None
    m1, m2 = ga.predict(y, T, return_cov=False, return_var=False), gb.predict(y, T, return_cov=False, return_var=False)
    assert np.array_equal(m1, m2) and ga.computed
    # a changed mean alone (george: a ConstantModel value), lists and float32 points: generic path, same values
    ga.mean.value += 1.0; gb.mean.value += 1.0
    gb._mean_plan = None
    assert np.array_equal(ga.predict(y, T, return_cov=False, return_var=False), gb.predict(y, T, return_cov=False, return_var=False))
    assert np.array_equal(ga.predict(y, T.tolist(), return_cov=False, return_var=False), ga.predict(y, T, return_cov=False, return_var=False))
    # another current stream
    with torch.cuda.stream(torch.cuda.Stream()):
        m3 = ga.predict(y, T, return_cov=False, return_var=False)
        torch.cuda.current_stream().synchronize()
    assert np.array_equal(m3, ga.predict(y, T, return_cov=False, return_var=False))


def test_repeated_single_candidate_prediction_plan_equals_the_generic_path():
    """``GP.predict(y, t, return_var=True)`` for ONE point at a time (the reference's scalar utilities under Nelder-Mead) re-uses
    the previous call's arguments (``_predict_one_again``), through the dense inverse or -- ``variance_mode = "solve"`` -- the
    substitution against L: same (mu, var) as the generic path; a switch of the form, new hyper-parameters or another y go
    back to it."""
    go, agp = _mods()
    n, D = 90, 2
    X, y = _case(n, D, 17)
    rs = np.random.RandomState(3)

    def make(mode):
        g = agp.GP(kernel=agp.ExpSquaredKernel(np.full(D, 3.0), ndim=D), fit_mean=True, mean=float(np.median(y)),
                   white_noise=-12, fit_white_noise=False)
        g.variance_mode = mode
        g.compute(X)
        return g
    for mode in (None, "solve"):
        ga, gb = make(mode), make(mode)
        used = 0
        for it in range(6):
            t = rs.uniform(-5, 5, size=(1, D))
            used += ga._one_plan is not None
            gb._one_plan = None
            ma, va = ga.predict(y, t, return_var=True)
            mb, vb = gb.predict(y, t, return_var=True)
            assert np.array_equal(ma, mb) and np.array_equal(va, vb) and ma.shape == va.shape == (1,)
        assert used >= 5
        # the other variance form: not the plan's
        ga.variance_mode = gb.variance_mode = "inverse" if mode == "solve" else "solve"
        gb._one_plan = None
        t = rs.uniform(-5, 5, size=(1, D))
        ra, rb = ga.predict(y, t, return_var=True), gb.predict(y, t, return_var=True)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        # new hyper-parameters, another y, two points at once
        p = ga.get_parameter_vector() + 0.05
        ga.set_parameter_vector(p); gb.set_parameter_vector(p)
        gb._one_plan = None
        ra, rb = ga.predict(y, t, return_var=True), gb.predict(y, t, return_var=True)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        y2 = y - 0.25
        gb._one_plan = None
        ra, rb = ga.predict(y2, t, return_var=True), gb.predict(y2, t, return_var=True)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        T2 = rs.uniform(-5, 5, size=(2, D))
        ra = ga.predict(y2, T2, return_var=True)
        assert ra[0].shape == (2,)
    gpo = go.GP(kernel=go.ExpSquaredKernel(np.exp(p[1:]), ndim=D), fit_mean=True, mean=float(p[0]), white_noise=-12,
                fit_white_noise=False)
    gpo.compute(X)
    mo, vo = gpo.predict(y, t, return_var=True)
    ma, va = ga.predict(y, t, return_var=True)
    ma, va = ga.predict(y, t, return_var=True)          # (the second call is the short path)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    assert abs(ma[0] - mo[0]) <= 1e-10 * asum and abs(va[0] - vo[0]) <= 1e-9


@pytest.mark.parametrize("n,D", [(1, 1), (50, 2), (64, 3), (90, 2), (129, 8), (200, 16), (256, 5), (257, 2)])
def test_single_candidate_prediction_in_one_launch(n, D):
    """``apgp_predict1_host`` through the dense inverse at n <= 256 (the README configuration's point search: one candidate per
    call under Nelder-Mead) is ONE single-workgroup launch; every sum is formed in the order of the three launches it replaces
    (mode 1 keeps those): (mu, sigma^2) bit-identical, and equal to the oracle's within fp64 tolerances."""
    from approxposterior_amd import _lib
    go, agp = _mods()
    lib = _lib.load()
    X, y = _case(n, D, 31 * n + D) if n > 1 else (np.array([[0.3]]), np.array([0.7]))
    rs = np.random.RandomState(n)
    g = agp.GP(kernel=2.5 * agp.ExpSquaredKernel(np.full(D, 4.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-10,
               fit_white_noise=False)
    g.variance_mode = "inverse"
    g.compute(X)
    gpo = go.GP(kernel=2.5 * go.ExpSquaredKernel(np.full(D, 4.0), ndim=D), fit_mean=True, mean=float(np.median(y)), white_noise=-10,
                fit_white_noise=False)
    gpo.compute(X)
    asum = np.abs(gpo._compute_alpha(y, False)).sum()
    for it in range(6):
        t = rs.uniform(-5, 5, size=(1, D))
        if it == 5:
            t[0, 0] = np.nan
        out = {}
        for mode in (0, 1):
            lib.apgp_potrf_mode(mode)
            try:
                g._one_plan = None
                out[mode] = g.predict(y, t, return_var=True)
                again = g.predict(y, t, return_var=True)             # (the repeated-call plan)
                assert np.array_equal(again[0], out[mode][0], equal_nan=True) and np.array_equal(again[1], out[mode][1], equal_nan=True)
            finally:
                lib.apgp_potrf_mode(0)
        assert out[0][0].tobytes() == out[1][0].tobytes() and out[0][1].tobytes() == out[1][1].tobytes()
        if it < 5:
            mo, vo = gpo.predict(y, t, return_var=True)
            assert abs(out[0][0][0] - mo[0]) <= 1e-10 * max(asum, 1.0) and abs(out[0][1][0] - vo[0]) <= 1e-9
        else:
            assert np.isnan(out[0][0][0]) and np.isnan(out[0][1][0])
