"""Edge cases through the C-ABI: empty and minimal inputs, capacity limits, ragged batches."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32


def _bfa():
    import bayesianfiltering_amd as bfa
    return bfa, bfa.nonlinearities


def test_empty_inputs_are_rejected_loudly():
    bfa, nl = _bfa()
    a = cm.cv_model_arrays()
    p = cm.product_params(a)
    for bad in (np.zeros((0, 2), F32), np.zeros((0, 5, 2), F32), np.zeros((3, 0, 2), F32)):
        with pytest.raises(ValueError):
            bfa.kalman_filter(p, bad) if bad.ndim == 3 else bfa.gaussian_sum_filter(p, bad, 2)
    with pytest.raises(ValueError):
        bfa.gaussian_sum_filter(p, np.zeros((4, 3), F32), 2)          # wrong emission dimension
    with pytest.raises(ValueError):
        bfa.gaussian_sum_filter(p, np.zeros((4, 2), F32), 0)          # no components


def test_single_step_single_trajectory():
    bfa, nl = _bfa()
    a = cm.cv_model_arrays()
    po, pp = cm.oracle_params(a), cm.product_params(a)
    y = np.array([[0.3, -0.2]], F32)
    init = np.array([[0.1, 0.0, -0.1, 0.0]], F32)
    ref = go.gaussian_sum_filter(po, y, 1, initial_means=init)
    post = bfa.gaussian_sum_filter(pp, y, 1, 1, initial_means=init)
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        assert tuple(getattr(post, k).shape) == getattr(ref, k).shape
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 1e-5, k
    kf = bfa.kalman_filter(pp, y[None], initial_means=init)
    assert cm.rel_err(kf.means.cpu().numpy()[0], ref.means) < 1e-5


@pytest.mark.parametrize("B", [1, 31, 33, 95, 257])
def test_ragged_batches(B):
    """Batches that do not fill the last wave / workgroup: the tail goes through the strided path."""
    bfa, nl = _bfa()
    a = cm.cv_model_arrays()
    po, pp = cm.oracle_params(a), cm.product_params(a)
    T = 12
    ys = cm.simulate_batch(a, B, T, seed=B)
    init = np.tile(a["m0"], (B, 1)).astype(F32)
    post = bfa.kalman_filter(pp, ys, initial_means=init)
    ref = cm.oracle_kalman_batch(a, ys[[0, B - 1]], init[[0, B - 1]])
    for k in ("means", "covariances", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy()[[0, B - 1]], ref[k]) < 1e-5, k
    gsf = bfa.gaussian_sum_filter(pp, ys, 4, 1, initial_means=np.repeat(init[:, None], 4, axis=1))
    assert cm.rel_err(gsf.means.cpu().numpy()[:, 0], post.means.cpu().numpy()[:, 0]) < 2e-5


def test_capacity_limits():
    bfa, nl = _bfa()
    # 256 components = every lane of a workgroup (scalar state: one lane per chain)
    po = go.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), om.Sine(1, 1.5), np.zeros(1, F32), 0.1 * np.eye(1, dtype=F32),
                        om.Quadratic(1, 0.5), np.zeros(1, F32), 0.5 * np.eye(1, dtype=F32))
    pp = bfa.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), nl.sine(1, 1.5), np.zeros(1, F32), 0.1 * np.eye(1, dtype=F32),
                         nl.quadratic(1, 0.5), np.zeros(1, F32), 0.5 * np.eye(1, dtype=F32))
    ys = go.sample_ssm(po, otf.PRNGKey(1), 6)[1]
    init = np.linspace(-2, 2, 256, dtype=F32).reshape(256, 1)
    ref = go.gaussian_sum_filter(po, ys, 256, initial_means=init)
    post = bfa.gaussian_sum_filter(pp, ys, 256, 1, initial_means=init)
    assert cm.rel_err(post.means.cpu().numpy(), ref.means) < 1e-4
    assert np.max(np.abs(post.weights.cpu().numpy() - ref.weights)) < 1e-5
    # 257 components no longer fit the register kernel's workgroup: the run-time-dimension kernel takes them in turns
    init2 = np.linspace(-2, 2, 257, dtype=F32).reshape(257, 1)
    ref2 = go.gaussian_sum_filter(po, ys, 257, initial_means=init2)
    post2 = bfa.gaussian_sum_filter(pp, ys, 257, 1, initial_means=init2)
    assert cm.rel_err(post2.means.cpu().numpy(), ref2.means) < 1e-4
    assert np.max(np.abs(post2.weights.cpu().numpy() - ref2.weights)) < 1e-5
    # 1024 leaves = the largest augmented tree; 1025 is refused
    a = cm.cv_model_arrays()
    p4 = cm.product_params(a)
    y4 = np.zeros((3, 2), F32)
    post, _ = bfa.speedy_augmented_gaussian_sum_filter(p4, y4, (16, 8, 8))
    assert tuple(post.means.shape) == (16, 3, 4) and bool(np.isfinite(post.means.cpu().numpy()).all())
    with pytest.raises(bfa.BayesFiltError):
        bfa.speedy_augmented_gaussian_sum_filter(p4, y4, (41, 5, 5))
    # the in-register particle capacities: 4096 in general, 16384 for small states; beyond, the particles live in HBM
    bp = bfa.ParamsBPF(*p4, nl.gaussian_log_prob(p4.emission_function, a["R"]))
    out = bfa.bootstrap_particle_filter(bp, y4, 16384, bfa.PRNGKey(0), output="summary")
    assert tuple(out["mean"].shape) == (3, 4) and bool(np.isfinite(out["mean"].cpu().numpy()).all())
    out = bfa.bootstrap_particle_filter(bp, y4, 16385, bfa.PRNGKey(0), output="summary")
    assert bool(np.isfinite(out["mean"].cpu().numpy()).all())
    with pytest.raises(bfa.BayesFiltError):
        bfa.bootstrap_particle_filter(bp, y4, (1 << 20) + 1, bfa.PRNGKey(0))


@pytest.mark.parametrize("n,m", [(4, 2), (12, 4), (20, 8), (32, 16), (48, 20), (64, 32)])
def test_nan_observation_poisons_the_trajectory_and_only_it(n, m):
    """A NaN observation at step t makes the innovation, the log-likelihood, the weight and the mean NaN from t on
    (inference.py:102-104, :347-350: nothing in the reference guards it) -- in every Kalman kernel family (register,
    run-time-dimension, one-wave matrix-core, padded and native (64, 32)): the launch terminates, the steps before t and the
    other trajectories are untouched."""
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(n, m, seed=n * 7 + m, bias=True)
    B, T, tb, bb = 3, 14, 5, 1
    ys = cm.simulate_batch(a, B, T, seed=2)
    init = np.tile(a["m0"], (B, 1))
    p = cm.product_params(a)
    clean, ll_clean = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True)
    bad = ys.copy()
    bad[bb, tb, 0] = np.nan
    post, ll = bfa.kalman_filter(p, bad, initial_means=init, return_loglik=True)
    mm, mc = post.means.cpu().numpy(), clean.means.cpu().numpy()
    assert np.isnan(mm[bb, 0, tb:]).all() and np.isnan(ll.cpu().numpy()[bb, 0, tb:]).all()
    assert np.isnan(post.weights.cpu().numpy()[bb, 0, tb:]).all()
    assert np.array_equal(mm[bb, 0, :tb], mc[bb, 0, :tb])
    for b in (0, 2):
        assert np.array_equal(mm[b], mc[b]) and np.array_equal(ll.cpu().numpy()[b], ll_clean.cpu().numpy()[b])
    # the covariance recursion does not depend on the data: it stays finite
    assert np.isfinite(post.covariances.cpu().numpy()).all()


def test_call_options_apply_to_one_call_only():
    """bf_set_call_option (Python: options={...}): a tuning option armed for ONE call on the calling thread -- the next filter
    entry point sees it, the one after does not, and the process-wide default (bf_set_option) is never touched: a library's
    choice of kernel variant cannot change another caller's results-to-rounding."""
    import threading
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    a = cm.random_stable_lgssm(4, 2, seed=3, bias=True)
    p = cm.product_params(a)
    ys = cm.simulate_batch(a, 7, 50, seed=3)
    default = bfa.kalman_filter(p, ys, return_loglik=True)
    once = bfa.kalman_filter(p, ys, return_loglik=True, options={"force_generic": 1})      # the run-time-dimension kernel, this call only
    after = bfa.kalman_filter(p, ys, return_loglik=True)
    _lib.check(lib.bf_set_option(b"force_generic", 1))
    try:
        glob = bfa.kalman_filter(p, ys, return_loglik=True)
    finally:
        _lib.check(lib.bf_set_option(b"force_generic", 0))
    for k in bfa.FULL5:
        assert torch.equal(getattr(once[0], k), getattr(glob[0], k)), k
        assert torch.equal(getattr(after[0], k), getattr(default[0], k)), k
    assert torch.equal(once[1], glob[1]) and torch.equal(after[1], default[1])
    assert not torch.equal(once[0].covariances, default[0].covariances)       # (the two kernels round differently: the option did act)
    # an override armed on ANOTHER thread is that thread's: this thread's next call is the default
    t = threading.Thread(target=lambda: _lib.check(lib.bf_set_call_option(b"force_generic", 1)))
    t.start(); t.join()
    again = bfa.kalman_filter(p, ys, return_loglik=True)
    assert torch.equal(again[0].covariances, default[0].covariances)
    # armed overrides are dropped even when the call fails
    _lib.check(lib.bf_set_call_option(b"force_generic", 1))
    assert lib.bf_kalman_filter_f32(None, None, 1, 1, None, None, None) == _lib.BF_EINVAL
    again = bfa.kalman_filter(p, ys, return_loglik=True)
    assert torch.equal(again[0].covariances, default[0].covariances)
    assert lib.bf_set_call_option(b"no_such_option", 1) == _lib.BF_EINVAL


def test_constant_cache_under_eviction_pressure():
    """csrc/const_cache.hip keeps at most 128 device-resident constant blocks and evicts least-recently-used -- never a block a
    call in progress was handed (it is pinned until the entry point returns).  150 distinct models, each with per-step
    covariance tables (three blocks per call), then the first one again: same bits as its first run."""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    rng = np.random.default_rng(0)
    T, B, n, m = 6, 4, 4, 2
    ys = torch.randn((B, T, m), device="cuda")

    def model(i):
        r = np.random.default_rng(i)
        A = (0.9 * np.eye(n) + 0.01 * r.normal(size=(n, n))).astype(F32)
        H = r.normal(size=(m, n)).astype(F32)
        Qt = np.stack([(0.05 + 0.01 * t) * np.eye(n) for t in range(T)]).astype(F32) * (1 + 0.001 * i)
        Rt = np.stack([(0.3 + 0.01 * t) * np.eye(m) for t in range(T)]).astype(F32)
        return bfa.ParamsNLSSM(np.zeros(n, F32), np.eye(n, dtype=F32), nl.linear_dynamics(A, np.eye(n, dtype=F32)), np.zeros(n, F32), Qt,
                               nl.linear_emission(H), np.zeros(m, F32), Rt)
    first = bfa.gaussian_sum_filter(model(0), ys, 2, 1, initial_means=np.zeros((B, 2, n), F32))
    keep = first.covariances.clone()
    for i in range(1, 150):
        out = bfa.gaussian_sum_filter(model(i), ys, 2, 1, initial_means=np.zeros((B, 2, n), F32))
    assert torch.isfinite(out.means).all()
    again = bfa.gaussian_sum_filter(model(0), ys, 2, 1, initial_means=np.zeros((B, 2, n), F32))
    assert torch.equal(again.covariances, keep)


def test_concurrent_callers_with_different_options_and_cache_pressure():
    """Four host threads call the library at once for a while: each with its own model, its own per-call options (two of them the
    run-time-dimension kernel), its own torch stream, and per-step covariance tables that churn the constant cache (more distinct
    blocks than it holds, so evictions run while other threads' calls are in flight).  Every result equals the same call made
    alone: per-call options are thread-local, cache entries are pinned while a call holds them, the entry points are re-entrant."""
    import threading
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, B, n, m, K = 8, 6, 4, 2, 3
    ys = torch.randn((B, T, m), device="cuda")
    im = np.zeros((B, K, n), F32)

    def model(i):
        r = np.random.default_rng(1000 + i)
        A = (0.9 * np.eye(n) + 0.02 * r.normal(size=(n, n))).astype(F32)
        H = r.normal(size=(m, n)).astype(F32)
        Qt = np.stack([(0.05 + 0.01 * t) * np.eye(n) for t in range(T)]).astype(F32) * (1 + 0.001 * i)
        return bfa.ParamsNLSSM(np.zeros(n, F32), np.eye(n, dtype=F32), nl.linear_dynamics(A), np.zeros(n, F32), Qt, nl.linear_emission(H),
                               np.zeros(m, F32), 0.3 * np.eye(m, dtype=F32))
    per_thread, rounds = 60, 2
    opts = [None, {"force_generic": 1}, None, {"force_generic": 1}]
    expected = {}
    for th in range(4):
        for i in range(per_thread):
            expected[(th, i)] = bfa.gaussian_sum_filter(model(th * per_thread + i), ys, K, 1, initial_means=im, options=opts[th]).covariances.clone()
    torch.cuda.synchronize()
    errors = []

    def worker(th):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    for i in range(per_thread):
                        out = bfa.gaussian_sum_filter(model(th * per_thread + i), ys, K, 1, initial_means=im, options=opts[th])
                        stream.synchronize()
                        if not torch.equal(out.covariances, expected[(th, i)]):
                            errors.append((th, i))
        except Exception as e:      # noqa: BLE001
            errors.append((th, repr(e)))
    threads = [threading.Thread(target=worker, args=(th,)) for th in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
