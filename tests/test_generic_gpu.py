"""The run-time-dimension kernel (csrc/generic_scan.hip): every (state_dim, obs_dim, components) the compile-time
instances do not cover -- the reference's _predict / _condition_on are dimension-generic (gaussfiltax/inference.py:51-105)
-- against the oracle at 1e-5, and against the compiled instances where both exist."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf, c_oracle
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32
FIELDS = ("means", "covariances", "predicted_means", "predicted_covariances")
# n = 9 ... 32 (every state dimension), obs_dim below, equal to and ABOVE the compiled limit of 4, m > n, and the larger shapes
DIMS = [(n, 1 + (5 * n) % min(n, 12)) for n in range(9, 33)] + [(2, 3), (3, 5), (8, 8), (6, 5), (16, 8), (32, 16), (40, 24), (48, 20), (64, 16), (96, 8)]


class _forced:
    def __enter__(self):
        from bayesianfiltering_amd import _lib
        self.lib = _lib.load()
        _lib.check(self.lib.bf_set_option(b"force_generic", 1))

    def __exit__(self, *exc):
        self.lib.bf_set_option(b"force_generic", 0)


@pytest.mark.parametrize("n,m", DIMS)
def test_kalman_any_dimension(n, m):
    import bayesianfiltering_amd as bfa
    dq, dr = max(1, n - (n + m) % 3), max(1, m - (n % 2))
    a = cm.random_stable_lgssm(n, m, seed=100 * n + m, dq=dq, dr=dr, bias=True)
    B, T = 5, 12 + (n % 5)
    ys = cm.simulate_batch(a, B, T, seed=n + m)
    init = np.tile(a["m0"], (B, 1)) + 0.1 * np.arange(B, dtype=F32)[:, None]
    ref = c_oracle.kalman_filter(a, ys, init)
    for layout in ("reference", "batch_inner"):
        post, ll = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, layout=layout, return_loglik=True)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < 1e-5, (layout, k, cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]))
        assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 5e-5, layout
        assert bool((post.weights == 1.0).all())


@pytest.mark.parametrize("n,m", [(4, 2), (8, 4), (5, 3), (1, 1)])
def test_generic_equals_compiled_instances(n, m):
    """Where both exist, the run-time-dimension kernel and the register kernel agree to rounding (and with the oracle)."""
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(n, m, seed=7 * n + m, bias=True)
    B, T = 9, 33
    ys = cm.simulate_batch(a, B, T, seed=3)
    init = np.tile(a["m0"], (B, 1))
    fast, llf = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, return_loglik=True)
    with _forced():
        slow, lls = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, return_loglik=True)
    ref = cm.oracle_kalman_batch(a, ys, init)
    for k in FIELDS:
        assert cm.rel_err(getattr(slow, k).cpu().numpy(), ref[k]) < 1e-5, k
        assert cm.rel_err(getattr(slow, k).cpu().numpy(), getattr(fast, k).cpu().numpy()) < 1e-5, k
    assert cm.rel_err(lls.cpu().numpy(), llf.cpu().numpy()) < 5e-5


def test_chunked_carry_equals_one_shot_any_dimension():
    import torch
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(12, 5, seed=4)
    ys = cm.simulate_batch(a, 6, 40, seed=4)
    p = cm.product_params(a)
    one, c1 = bfa.kalman_filter(p, ys, return_carry=True)
    h1, c = bfa.kalman_filter(p, ys[:, :17], return_carry=True)
    h2, c2 = bfa.kalman_filter(p, ys[:, 17:], carry=c, return_carry=True)
    for k in FIELDS:
        assert torch.equal(torch.cat([getattr(h1, k), getattr(h2, k)], dim=2), getattr(one, k)), k
    assert torch.equal(c2.covariances, c1.covariances) and torch.equal(c2.means, c1.means)


def test_gsf_300_components_lorenz63():
    """K = 300 components of the Lorenz-63 EKF bank exceed one workgroup's lanes (K x lanes <= 256 in gsf_scan.hpp): the
    components take turns in the LDS tile; weights in the oracle's adjacent-pair tree order."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    K, T = 300, 10
    Q, R = 0.1 * np.eye(3, dtype=F32), 1.0 * np.eye(1, dtype=F32)
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    po = go.ParamsNLSSM(m0, P0, om.Lorenz63(), np.zeros(3, F32), Q, om.Quadratic(3, 0.05), np.zeros(1, F32), R)
    pp = bfa.ParamsNLSSM(m0, P0, nl.lorenz63(), np.zeros(3, F32), Q, nl.quadratic(3, 0.05), np.zeros(1, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(2)])
    im = (m0 + 0.5 * np.random.default_rng(0).normal(size=(2, K, 3))).astype(F32)
    post, carry = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im, return_carry=True)
    for b in range(2):
        ref = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b])
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 2e-5, (b, k)
        assert np.max(np.abs(post.weights[b].cpu().numpy() - ref.weights)) < 2e-5, b
        assert cm.rel_err(carry.means[b].cpu().numpy(), ref.predicted_means[:, -1]) < 2e-5
    # in two chunks through the carry: bit for bit
    import torch
    h1, c = bfa.gaussian_sum_filter(pp, ys[:, :4], K, 1, initial_means=im, return_carry=True)
    h2 = bfa.gaussian_sum_filter(pp, ys[:, 4:], K, 1, carry=c)
    assert torch.equal(torch.cat([h1.weights, h2.weights], dim=2), post.weights)
    assert torch.equal(torch.cat([h1.covariances, h2.covariances], dim=2), post.covariances)


@pytest.mark.parametrize("n,K", [(16, 4), (12, 7), (40, 2)])
def test_gsf_lorenz96_larger_states(n, K):
    """Lorenz-96 (gaussfiltax/nonlinearities.py:37-52) beyond state_dim 8: the cfg4 model's dimensions as an EKF bank."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    m, T = n // 2, 8
    Q, R = 1e-2 * np.eye(n, dtype=F32), 1e-1 * np.eye(m, dtype=F32)
    m0 = 8 * np.ones(n, F32)
    po = go.ParamsNLSSM(m0, np.eye(n, dtype=F32), om.Lorenz96(n), np.zeros(n, F32), Q, om.PickEven(n), np.zeros(m, F32), R)
    pp = bfa.ParamsNLSSM(m0, np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), Q, nl.pick_even(n), np.zeros(m, F32), R)
    ys = go.sample_ssm(po, otf.PRNGKey(n), T)[1]
    im = (m0 + np.random.default_rng(n).normal(size=(K, n))).astype(F32)
    post = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im)
    ref = go.gaussian_sum_filter(po, ys, K, initial_means=im)
    for k in FIELDS:
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 2e-5, k
    assert np.max(np.abs(post.weights.cpu().numpy() - ref.weights)) < 5e-5


@pytest.mark.parametrize("n,m", [(12, 3), (64, 32)])
def test_time_varying_covariances_any_dimension(n, m):
    """(T, d, d) covariances (the `_get_params(x, 2, t)` rule, inference.py:21, :337-340) beyond the compiled instances --
    including (64, 32), where the MFMA kernel takes constant covariances only."""
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(n, m, seed=n)
    B, T = 2, 9
    ys = cm.simulate_batch(a, B, T, seed=1)
    rng = np.random.default_rng(5)
    Qt = np.stack([(1 + 0.5 * rng.random()) * a["Q"] for _ in range(T)]).astype(F32)
    Rt = np.stack([(1 + 0.5 * rng.random()) * a["R"] for _ in range(T)]).astype(F32)
    pp = cm.product_params(a)._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt)
    po = cm.oracle_params(a)._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt)
    init = np.tile(a["m0"], (B, 1))
    post = bfa.kalman_filter(pp, ys, initial_means=init)
    for b in range(B):
        ref = go.gaussian_sum_filter(po, ys[b], 1, initial_means=init[b].reshape(1, -1))
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 1e-5, (b, k)


@pytest.mark.parametrize("n,dq,m,dr", [(10, 10, 4, 4), (64, 64, 32, 32), (5, 3, 2, 3)])
def test_data_generator_any_dimension(n, dq, m, dr):
    """NonlinearSSM.sample (gaussfiltax/models.py:240-289) at shapes the compiled sampler has no instance for."""
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(n, m, seed=n + m, dq=dq, dr=dr, bias=True)
    T = 7
    keys = otf.split(otf.PRNGKey(9), 3)
    xs, ys = bfa.NonlinearSSM(n, dq, m, dr).sample(cm.product_params(a), keys, T)
    for b in range(3):
        rx, ry = go.sample_ssm(cm.oracle_params(a), keys[b], T)
        assert cm.rel_err(xs[b].cpu().numpy(), rx) < 2e-6 and cm.rel_err(ys[b].cpu().numpy(), ry) < 2e-6


def test_data_generator_lorenz96_n40():
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    n, m, T = 40, 20, 6
    Q, R = 1e-2 * np.eye(n, dtype=F32), 1e-1 * np.eye(m, dtype=F32)
    po = go.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), om.Lorenz96(n), np.zeros(n, F32), Q, om.PickEven(n), np.zeros(m, F32), R)
    pp = bfa.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), Q, nl.pick_even(n), np.zeros(m, F32), R)
    xs, ys = bfa.NonlinearSSM(n, n, m, m).sample(pp, otf.PRNGKey(3), T)
    rx, ry = go.sample_ssm(po, otf.PRNGKey(3), T)
    assert cm.rel_err(xs.cpu().numpy(), rx) < 2e-6 and cm.rel_err(ys.cpu().numpy(), ry) < 2e-6


@pytest.mark.parametrize("n,m", [(9, 9), (12, 30), (16, 8), (17, 3), (24, 5), (32, 16), (32, 32), (33, 32), (40, 24), (48, 20),
                                 (64, 16), (64, 1), (64, 32)])
def test_models_padded_into_the_matrix_core_kernel(n, m):
    """Kalman models ride zero-padded in matrix-core tiles (unit-noise dummy observations, their log N(0; 0, 1) taken off
    the log-likelihood): up to n = 32 on the one-wave-per-trajectory kernel (single 32 x 32 tiles), from there to n = 64 in
    the (64, 32) kernel -- against the oracle at 1e-5, against the run-time-dimension kernel, and chunked through the carry
    == one shot bit for bit."""
    import bayesianfiltering_amd as bfa
    dq, dr = max(1, n - 3), max(1, m - (n % 2))
    a = cm.random_stable_lgssm(n, m, seed=11 * n + m, dq=dq, dr=dr, bias=True)
    B, T = 6, 40
    ys = cm.simulate_batch(a, B, T, seed=abs(n - m) + 1)
    init = np.tile(a["m0"], (B, 1)) + 0.05 * np.arange(B, dtype=F32)[:, None]
    ref = c_oracle.kalman_filter(a, ys, init)
    p = cm.product_params(a)
    fast, llf, carry = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True, return_carry=True)
    with _forced():
        slow, lls = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True)
    for k in FIELDS:
        assert cm.rel_err(getattr(fast, k).cpu().numpy(), ref[k]) < 1e-5, (k, cm.rel_err(getattr(fast, k).cpu().numpy(), ref[k]))
        assert cm.rel_err(getattr(fast, k).cpu().numpy(), getattr(slow, k).cpu().numpy()) < 1e-5, k
    assert cm.rel_err(llf.cpu().numpy(), ref["loglik"]) < 5e-5
    assert cm.rel_err(llf.cpu().numpy(), lls.cpu().numpy()) < 5e-5
    # two chunks through the carry
    h1, l1, c1 = bfa.kalman_filter(p, ys[:, :17], initial_means=init, return_loglik=True, return_carry=True)
    h2, l2, c2 = bfa.kalman_filter(p, ys[:, 17:], carry=c1, return_loglik=True, return_carry=True)
    for k in FIELDS:
        whole = getattr(fast, k).cpu().numpy()
        assert np.array_equal(np.concatenate([getattr(h1, k).cpu().numpy(), getattr(h2, k).cpu().numpy()], axis=2), whole), k
    assert np.array_equal(c2.covariances.cpu().numpy(), carry.covariances.cpu().numpy())


@pytest.mark.parametrize("n,m", [(16, 8), (24, 12), (32, 16)])
def test_small_mode_off_routes_to_the_other_kernels(n, m):
    """bf_set_option("kf_small_mode", 0): the same models on the (64, 32) kernel (n >= 24) or the run-time-dimension kernel."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    a = cm.random_stable_lgssm(n, m, seed=n + m, bias=True)
    B, T = 4, 25
    ys = cm.simulate_batch(a, B, T, seed=1)
    init = np.tile(a["m0"], (B, 1))
    ref = c_oracle.kalman_filter(a, ys, init)
    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_small_mode", 0))
    try:
        post, ll = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, return_loglik=True)
    finally:
        _lib.check(lib.bf_set_option(b"kf_small_mode", 1))
    for k in FIELDS:
        assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < 1e-5, k
    assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 5e-5


@pytest.mark.parametrize("n,m,K,tv", [(16, 8, 4, False), (32, 16, 32, False), (24, 12, 5, True), (32, 16, 4, True), (12, 10, 3, True),
                                       (32, 32, 64, False), (64, 32, 4, False), (64, 32, 32, True), (48, 24, 5, True), (33, 7, 2, False)])
def test_gaussian_sum_of_a_linear_model_on_the_matrix_cores(n, m, K, tv):
    """_predict / _condition_on are vmapped over the components and read per-step covariances (inference.py:21,51-105,337-353):
    for LINEAR models of 9 <= n <= 64 the K components take turns on the matrix-core kernels (bf16 three-term products: one
    wave per trajectory up to n = 32, four waves up to (64, 32)), per-step Q_t / R_t tables included.  Against the oracle at 1e-5 (weights absolutely), against the
    run-time-dimension kernel, and two chunks through the carry == one shot bit for bit."""
    import torch
    import bayesianfiltering_amd as bfa
    a = cm.random_stable_lgssm(n, m, seed=n + m + K, dq=max(1, n - 2), dr=m, bias=True)
    B, T = 3, 14
    ys = cm.simulate_batch(a, B, T, seed=K)
    rng = np.random.default_rng(K)
    pp, po = cm.product_params(a), cm.oracle_params(a)
    if tv:
        Qt = np.stack([(0.6 + rng.random()) * a["Q"] for _ in range(T)]).astype(F32)
        Rt = np.stack([(0.6 + rng.random()) * a["R"] for _ in range(T)]).astype(F32)
        pp = pp._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt)
        po = po._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt)
    im = (a["m0"] + 0.5 * rng.normal(size=(B, K, n))).astype(F32)
    post, ll, carry = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im, return_loglik=True, return_carry=True)
    with _forced():
        slow = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im)
    assert not torch.equal(post.covariances, slow.covariances)          # (another kernel did run: the two round differently)
    for b in range(B):
        ref, rll = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b], return_ll=True)
        for k in FIELDS:
            e = cm.both_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k), k)
            assert e[0] < 1e-5, (b, k, e)
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(slow, k)[b].cpu().numpy()) < 1e-5, (b, k)
        assert np.max(np.abs(post.weights[b].cpu().numpy() - ref.weights)) < 2e-5, b
        assert cm.rel_err(ll[b].cpu().numpy(), rll) < 2e-5, b
        assert cm.rel_err(carry.means[b].cpu().numpy(), ref.predicted_means[:, -1]) < 1e-5
        assert np.max(np.abs(carry.weights[b].cpu().numpy() - ref.weights[:, -1])) < 2e-5
    # two chunks through the carry (the per-step tables are sliced with the observations)
    cut = 6
    p1 = pp if not tv else pp._replace(dynamics_noise_covariance=Qt[:cut], emission_noise_covariance=Rt[:cut])
    p2 = pp if not tv else pp._replace(dynamics_noise_covariance=Qt[cut:], emission_noise_covariance=Rt[cut:])
    h1, c1 = bfa.gaussian_sum_filter(p1, ys[:, :cut], K, 1, initial_means=im, return_carry=True)
    h2, c2 = bfa.gaussian_sum_filter(p2, ys[:, cut:], K, 1, carry=c1, return_carry=True)
    for k in FIELDS + ("weights",):
        assert torch.equal(torch.cat([getattr(h1, k), getattr(h2, k)], dim=2), getattr(post, k)), k
    for x, y_ in zip(c2, carry):
        assert torch.equal(x, y_)


def test_time_varying_kalman_on_the_matrix_cores():
    """bf_kalman_filter_f32 with (T, d, d) covariances at (32, 16): the one-wave matrix-core kernel reads per-step tables."""
    import torch
    import bayesianfiltering_amd as bfa
    n, m, B, T = 32, 16, 4, 20
    a = cm.random_stable_lgssm(n, m, seed=77, bias=True)
    ys = cm.simulate_batch(a, B, T, seed=7)
    rng = np.random.default_rng(7)
    Qt = np.stack([(0.6 + rng.random()) * a["Q"] for _ in range(T)]).astype(F32)
    Rt = np.stack([(0.6 + rng.random()) * a["R"] for _ in range(T)]).astype(F32)
    init = np.tile(a["m0"], (B, 1))
    for kw in ({"dynamics_noise_covariance": Qt, "emission_noise_covariance": Rt}, {"dynamics_noise_covariance": Qt}, {"emission_noise_covariance": Rt}):
        pp, po = cm.product_params(a)._replace(**kw), cm.oracle_params(a)._replace(**kw)
        post, ll = bfa.kalman_filter(pp, ys, initial_means=init, return_loglik=True)
        with _forced():
            slow = bfa.kalman_filter(pp, ys, initial_means=init)
        assert not torch.equal(post.covariances, slow.covariances)
        for b in range(B):
            ref, rll = go.gaussian_sum_filter(po, ys[b], 1, initial_means=init[b].reshape(1, -1), return_ll=True)
            for k in FIELDS:
                assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 1e-5, (b, k)
            assert cm.rel_err(ll[b].cpu().numpy(), rll) < 2e-5


@pytest.mark.parametrize("model,n,K,tv", [("lorenz96", 16, 4, False), ("lorenz96", 32, 32, False), ("lorenz96", 24, 5, True), ("lorenz96_as_written", 32, 3, False),
                                          ("sine", 16, 6, False), ("lorenz96", 20, 1, False), ("lorenz96", 40, 2, False), ("lorenz96", 64, 4, True),
                                          ("sine", 48, 3, False)])
def test_extended_kalman_chains_on_the_matrix_cores(model, n, K, tv):
    """Gaussian-sum filters of NONLINEAR registry dynamics with a linear emission (Lorenz-96 with the even-state emission,
    gaussfiltax/nonlinearities.py:37-52, at 16 <= n <= 64; the sine map) on the matrix-core kernels (one wave per chain up to
    n = 32, four beyond): the Jacobian's
    row is evaluated analytically at the filtered mean every step (inference.py:328, :61-62) and re-split into bf16 terms in
    the operand registers.  Against the oracle at 1e-5, against the run-time-dimension kernel, chunked == one shot."""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    m, T, B = n // 2, 14, 3
    rng = np.random.default_rng(n + K)
    Q, R = (1e-2 * np.eye(n)).astype(F32), (1e-1 * np.eye(m)).astype(F32)
    q0 = (0.01 * rng.normal(size=n)).astype(F32)
    if model.startswith("lorenz96"):
        mode = "as_written" if model.endswith("as_written") else "matrix_power"
        m0 = 8 * np.ones(n, F32)
        fo, fp = om.Lorenz96(n, mode=mode), nl.lorenz96(n, mode=mode)
    else:
        m0 = np.zeros(n, F32)
        fo, fp = om.Sine(n, 1.5), nl.sine(n, 1.5)
    po = go.ParamsNLSSM(m0, np.eye(n, dtype=F32), fo, q0, Q, om.PickEven(n), np.zeros(m, F32), R)
    pp = bfa.ParamsNLSSM(m0, np.eye(n, dtype=F32), fp, q0, Q, nl.pick_even(n), np.zeros(m, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    if tv:
        Qt = np.stack([(0.6 + rng.random()) * Q for _ in range(T)]).astype(F32)
        pp, po = pp._replace(dynamics_noise_covariance=Qt), po._replace(dynamics_noise_covariance=Qt)
    im = (m0 + 0.5 * rng.normal(size=(B, K, n))).astype(F32)
    post, ll, carry = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im, return_loglik=True, return_carry=True)
    with _forced():
        slow = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im)
    assert not torch.equal(post.covariances, slow.covariances)          # (another kernel did run: the two round differently)
    for b in range(B):
        ref, rll = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b], return_ll=True)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 1e-5, (b, k, cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)))
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(slow, k)[b].cpu().numpy()) < 1e-5, (b, k)
        assert np.max(np.abs(post.weights[b].cpu().numpy() - ref.weights)) < 2e-5, b
        assert cm.rel_err(ll[b].cpu().numpy(), rll) < 2e-5, b
    h1, c1 = bfa.gaussian_sum_filter(pp if not tv else pp._replace(dynamics_noise_covariance=Qt[:6]), ys[:, :6], K, 1, initial_means=im, return_carry=True)
    h2, c2 = bfa.gaussian_sum_filter(pp if not tv else pp._replace(dynamics_noise_covariance=Qt[6:]), ys[:, 6:], K, 1, carry=c1, return_carry=True)
    for k in FIELDS + ("weights",):
        assert torch.equal(torch.cat([getattr(h1, k), getattr(h2, k)], dim=2), getattr(post, k)), k
    for x, y_ in zip(c2, carry):
        assert torch.equal(x, y_)


@pytest.mark.parametrize("case", ["kalman_odd_batch", "gaussian_sum_tv", "lorenz96_chains"])
def test_two_chains_per_wave_variant_of_the_one_wave_kernel(case):
    """kf_small_mode = 2: the one-wave matrix-core kernel with a second chain in the upper half-wave (one factorization for two
    chains, column broadcasts by ds_swizzle).  Same operations on the same values: every output bit equal to the default
    variant's, odd chain counts included.  (Measured slower -- one wave per SIMD loses more than the shared factorization wins --
    so it stays an option)"""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    if case == "lorenz96_chains":
        n, K, B, T = 24, 5, 33, 20
        m = n // 2
        p = bfa.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), (1e-2 * np.eye(n)).astype(F32),
                            nl.pick_even(n), np.zeros(m, F32), (1e-1 * np.eye(m)).astype(F32))
        y = cm.device_observations(p, (n, n, m, m), B, T, seed=n)
        init = 8.0 + torch.randn((B, K, n), device="cuda")
    else:
        n, m, K, B, T = (32, 16, 1, 77, 25) if case == "kalman_odd_batch" else (20, 12, 3, 41, 20)
        a = cm.random_stable_lgssm(n, m, seed=n)
        p = cm.product_params(a)
        if case == "gaussian_sum_tv":
            rng = np.random.default_rng(1)
            p = p._replace(dynamics_noise_covariance=np.stack([(0.6 + rng.random()) * a["Q"] for _ in range(T)]).astype(F32))
        y = cm.device_observations(cm.product_params(a), (n, n, m, m), B, T, seed=n)
        init = torch.as_tensor(a["m0"], device="cuda") + 0.3 * torch.randn((B, K, n), device="cuda")
    outs = [bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, return_carry=True, options={"kf_small_mode": mode}) for mode in (1, 2)]
    for k in ("weights", "means", "covariances", "predicted_means", "predicted_covariances"):
        x, z = getattr(outs[0][0], k), getattr(outs[1][0], k)
        assert bool(((x == z) | (torch.isnan(x) & torch.isnan(z))).all()), k
    for x, z in zip(outs[0][1], outs[1][1]):
        assert bool(((x == z) | (torch.isnan(x) & torch.isnan(z))).all())
