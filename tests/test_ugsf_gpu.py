"""GPU parity of the unscented Gaussian-sum filter (bf_ugsf_ukf_f32; gaussfiltax/inference.py:379-456,
146-174, 198-224, utils.py:247-254) against the NumPy oracle.  Tolerance: 2e-5 relative on means and
covariances (north_star: 1e-5; the symmetric matrix square root is recomputed twice per step on both
sides -- float64 eigh in the oracle, float32 Jacobi on the device -- and measured errors are 2-4e-6)."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32
FIELDS = ("means", "covariances", "predicted_means", "predicted_covariances")


def _nl():
    import bayesianfiltering_amd as bfa
    return bfa, bfa.nonlinearities


def _check(post, ref, tol, wtol=5e-5):
    for k in FIELDS:
        got, exp = getattr(post, k).cpu().numpy(), getattr(ref, k) if hasattr(ref, "_fields") else ref[k]
        assert got.shape == exp.shape, k
        assert cm.rel_err(got, exp) < tol, (k, cm.rel_err(got, exp))
    we = ref.weights if hasattr(ref, "_fields") else ref["weights"]
    assert np.max(np.abs(post.weights.cpu().numpy() - we)) < wtol


def _oracle_batch(p, up, ys, K, init, inputs=None):
    outs = {k: [] for k in ("weights",) + FIELDS}
    lls = []
    for b in range(ys.shape[0]):
        post, ll = go.unscented_gaussian_sum_filter(p, up, ys[b], K, initial_means=init[b], inputs=inputs, return_ll=True)
        for k in outs:
            outs[k].append(getattr(post, k))
        lls.append(ll)
    return {k: np.stack(v) for k, v in outs.items()}, np.stack(lls)


@pytest.mark.parametrize("uparams", [(1.0, 0.0, 0.0), (0.5, 2.0, 1.0)])
def test_linear_model_matches_oracle_and_the_kalman_filter(uparams):
    """The unscented transform is exact for linear f, h: besides the oracle, the extended filter is a
    second reference (different algebra, same posterior)."""
    bfa, nl = _nl()
    a = cm.cv_model_arrays()
    po, pp = cm.oracle_params(a), cm.product_params(a)
    T, B, K = 30, 5, 3
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    init = np.random.default_rng(0).normal(size=(B, K, 4)).astype(F32)
    up = bfa.ParamsUKF(*uparams)
    ref, ref_ll = _oracle_batch(po, go.ParamsUKF(*uparams), ys, K, init)
    post, ll = bfa.unscented_gaussian_sum_filter(pp, up, ys, K, 1, initial_means=init, return_loglik=True)
    assert tuple(post.means.shape) == (B, K, T, 4) and tuple(post.covariances.shape) == (B, K, T, 4, 4)
    _check(post, ref, tol=1e-5)
    assert cm.rel_err(ll.cpu().numpy(), ref_ll) < 5e-5
    ekf = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=init)
    for k in FIELDS:
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ekf, k).cpu().numpy()) < 5e-5, k
    # unbatched call: reference shapes (K, T, ...)
    one = bfa.unscented_gaussian_sum_filter(pp, up, ys[0], K, 1, initial_means=init[0])
    assert tuple(one.means.shape) == (K, T, 4)
    assert np.array_equal(one.means.cpu().numpy(), post.means[0].cpu().numpy())


def test_bearings_only_tracking_with_inputs():
    """BOT_Experiment_script.py:108-110: ParamsUKF(1, 0, 0), manoeuvring target, bearing + range."""
    bfa, nl = _nl()
    rng = np.random.default_rng(0)
    T, K, B = 24, 5, 3
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32)
    po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R)
    pp = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(10 + b), T, inputs.reshape(T, 1))[1] for b in range(B)])
    init = (mu0 + 0.05 * rng.normal(size=(B, K, 4))).astype(F32)
    ref, _ = _oracle_batch(po, go.ParamsUKF(1, 0, 0), ys, K, init, inputs.reshape(T, 1))
    post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, K, 1, inputs, initial_means=init)
    _check(post, ref, tol=1e-5)
    # the scan in two chunks through the carry reproduces the single scan bit for bit
    p1, c1 = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys[:, :10], K, 1, inputs[:10], initial_means=init,
                                               return_carry=True)
    p2 = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys[:, 10:], K, 1, inputs[10:], carry=c1)
    for k in FIELDS + ("weights",):
        cat = np.concatenate([getattr(p1, k).cpu().numpy(), getattr(p2, k).cpu().numpy()], axis=2)
        assert cm.rel_err(cat, getattr(post, k).cpu().numpy()) < 1e-6, k


def test_lorenz63_quadratic_and_scalar_models():
    bfa, nl = _nl()
    T, K = 20, 4
    po = go.ParamsNLSSM(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), om.Lorenz63(), np.zeros(3, F32),
                        0.1 * np.eye(3, dtype=F32), om.Quadratic(3, 0.05), np.zeros(1, F32), np.eye(1, dtype=F32))
    pp = bfa.ParamsNLSSM(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(), np.zeros(3, F32),
                         0.1 * np.eye(3, dtype=F32), nl.quadratic(3, 0.05), np.zeros(1, F32), np.eye(1, dtype=F32))
    ys = go.sample_ssm(po, otf.PRNGKey(4), T)[1]
    init = (np.array([0.0, 1.0, 1.05], F32) + np.random.default_rng(3).normal(size=(K, 3))).astype(F32)
    for up in ((1.0, 0.0, 0.0), (1.0, 2.0, 0.5)):
        ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(*up), ys, K, initial_means=init)
        _check(bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(*up), ys, K, 1, initial_means=init), ref, tol=1e-5)
    # scalar state: sin dynamics with the quadratic emission (f1 / g1 of Experiment_TSP_2023.ipynb, w0 = 1.5)
    po = go.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), om.Sine(1, 1.5), np.zeros(1, F32), 0.1 * np.eye(1, dtype=F32),
                        om.Quadratic(1, 0.5), np.zeros(1, F32), 0.5 * np.eye(1, dtype=F32))
    pp = bfa.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), nl.sine(1, 1.5), np.zeros(1, F32), 0.1 * np.eye(1, dtype=F32),
                         nl.quadratic(1, 0.5), np.zeros(1, F32), 0.5 * np.eye(1, dtype=F32))
    ys = go.sample_ssm(po, otf.PRNGKey(5), T)[1]
    init = np.array([[0.4], [-0.6], [1.0]], F32)
    ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(1, 0, 0), ys, 3, initial_means=init)
    _check(bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, 3, 1, initial_means=init), ref, tol=1e-5)


def test_stochastic_volatility_non_additive_noise():
    """h(x, r, u) = u beta exp(x / sigma) r + (1 - u)(c x + r): the emission noise enters multiplicatively,
    the case the augmented (non-additive) sigma points exist for (adaptive_experiment.py:51-54)."""
    bfa, nl = _nl()
    T, K = 20, 4
    Phi = 0.8 * np.eye(2, dtype=F32)
    Q, R = 0.5 * np.eye(2, dtype=F32), 1e-1 * np.eye(2, dtype=F32)
    inputs = np.array([0] * 10 + [1] * 10, F32)
    r0 = np.array([0.1, -0.2], F32)
    po = go.ParamsNLSSM(np.zeros(2, F32), np.eye(2, dtype=F32), om.Linear(Phi), np.zeros(2, F32), Q, om.StochVol(2), r0, R)
    pp = bfa.ParamsNLSSM(np.zeros(2, F32), np.eye(2, dtype=F32), nl.linear_dynamics(Phi), np.zeros(2, F32), Q,
                         nl.stoch_vol(2), r0, R)
    ys = go.sample_ssm(po, otf.PRNGKey(3), T, inputs.reshape(T, 1))[1]
    init = np.random.default_rng(1).normal(size=(K, 2)).astype(F32)
    ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(1, 0, 0), ys, K, initial_means=init, inputs=inputs.reshape(T, 1))
    post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, K, 1, inputs, initial_means=init)
    _check(post, ref, tol=1e-5)


def test_lorenz96_eight_states_and_many_components():
    bfa, nl = _nl()
    T, B = 12, 2
    po = go.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), om.Lorenz96(8), np.zeros(8, F32),
                        1e-2 * np.eye(8, dtype=F32), om.PickEven(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    pp = bfa.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32),
                         1e-2 * np.eye(8, dtype=F32), nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    for K in (2, 100):   # K = 100: 128 lanes per trajectory, the reweight continues through LDS
        init = np.random.default_rng(K).normal(size=(B, K, 8)).astype(F32)
        ref, _ = _oracle_batch(po, go.ParamsUKF(1, 0, 0), ys, K, init)
        _check(bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, K, 1, initial_means=init), ref, tol=2e-5)   # measured 1.8e-5 at K = 100 (unscented transform of the chaotic map, 12 steps)


def test_errors():
    bfa, nl = _nl()
    a = cm.cv_model_arrays()
    pp = cm.product_params(a)
    ys = np.zeros((6, 2), F32)
    with pytest.raises(bfa.BayesFiltError):    # alpha must be positive
        bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(0.0, 2, 0), ys, 2)
    with pytest.raises(bfa.BayesFiltError):    # more components than one workgroup holds
        bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, 300)
    Qt = np.stack([a["Q"]] * 5)
    with pytest.raises(bfa.BayesFiltError):    # time-varying covariances: one matrix per step
        bfa.unscented_gaussian_sum_filter(pp._replace(dynamics_noise_covariance=Qt), bfa.ParamsUKF(1, 0, 0), ys, 2)


def test_golden_bearings_only_fixture(golden_dir):
    bfa, nl = _nl()
    d = np.load(f"{golden_dir}/ugsf_bot_K4_T24.npz")
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    pp = bfa.ParamsNLSSM(mu0, np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32), nl.maneuver_bot(), np.zeros(2, F32),
                         1e-3 * np.eye(2, dtype=F32), nl.bearing_range(), np.zeros(2, F32), np.diag([1e-3, 1e-2]).astype(F32))
    post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(*d["uparams"]), d["emissions"], 4, 1, d["inputs"],
                                             initial_means=d["initial_means"])
    _check(post, d, tol=1e-5)


def test_time_varying_covariances():
    """(T, d, d) noise covariances: the step's Q_t / R_t are picked by _get_params (inference.py:411-417) before
    the unscented transforms, whose sigma points then use sqrtm(blockdiag(P, R_t)) / sqrtm(blockdiag(P, Q_t)).
    Linear model (the extended filter with the same tables is a second reference), non-diagonal tables, the
    bearings-only model with a non-identity F_q, and the multiplicative-noise emission."""
    bfa, nl = _nl()
    rng = np.random.default_rng(5)
    T, B, K = 22, 3, 3
    a = cm.cv_model_arrays()
    po, pp = cm.oracle_params(a), cm.product_params(a)

    def spd_table(base, T):
        d = base.shape[0]
        out = []
        for t in range(T):
            w = rng.normal(size=(d, d)) * 0.3
            out.append(base * (0.5 + rng.uniform()) + 0.2 * np.mean(np.diag(base)) * (w @ w.T))
        return np.stack(out).astype(F32)

    Qt, Rt = spd_table(a["Q"], T), spd_table(a["R"], T)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    init = rng.normal(size=(B, K, 4)).astype(F32)
    up = (1.0, 0.0, 0.0)
    for kw in ({"dynamics_noise_covariance": Qt, "emission_noise_covariance": Rt}, {"dynamics_noise_covariance": Qt},
               {"emission_noise_covariance": Rt}):
        ref, ref_ll = _oracle_batch(po._replace(**kw), go.ParamsUKF(*up), ys, K, init)
        post, ll = bfa.unscented_gaussian_sum_filter(pp._replace(**kw), bfa.ParamsUKF(*up), ys, K, 1, initial_means=init,
                                                     return_loglik=True)
        _check(post, ref, tol=1e-5)
        assert cm.rel_err(ll.cpu().numpy(), ref_ll) < 1e-5
        ekf = bfa.gaussian_sum_filter(pp._replace(**kw), ys, K, 1, initial_means=init)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ekf, k).cpu().numpy()) < 1e-4, k
    # and they matter: the constant-covariance posterior is a different one
    const = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(*up), ys, K, 1, initial_means=init)
    assert cm.rel_err(const.covariances.cpu().numpy(), ref["covariances"]) > 1e-2

    # multiplicative emission noise with a per-step R (no constant H_r needed here: R only enters the sigma points)
    T2, K2 = 20, 4
    Phi = 0.8 * np.eye(2, dtype=F32)
    Q, R = 0.5 * np.eye(2, dtype=F32), 1e-1 * np.eye(2, dtype=F32)
    Rt2 = spd_table(R, T2)
    inputs = np.array([0] * 10 + [1] * 10, F32)
    r0 = np.array([0.1, -0.2], F32)
    po2 = go.ParamsNLSSM(np.zeros(2, F32), np.eye(2, dtype=F32), om.Linear(Phi), np.zeros(2, F32), Q, om.StochVol(2), r0, Rt2)
    pp2 = bfa.ParamsNLSSM(np.zeros(2, F32), np.eye(2, dtype=F32), nl.linear_dynamics(Phi), np.zeros(2, F32), Q,
                          nl.stoch_vol(2), r0, Rt2)
    ys2 = go.sample_ssm(po2._replace(emission_noise_covariance=R), otf.PRNGKey(3), T2, inputs.reshape(T2, 1))[1]
    init2 = rng.normal(size=(K2, 2)).astype(F32)
    ref2 = go.unscented_gaussian_sum_filter(po2, go.ParamsUKF(1, 0, 0), ys2, K2, initial_means=init2, inputs=inputs.reshape(T2, 1))
    _check(bfa.unscented_gaussian_sum_filter(pp2, bfa.ParamsUKF(1, 0, 0), ys2, K2, 1, inputs, initial_means=init2), ref2, tol=1e-5)
