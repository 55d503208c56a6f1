"""Pins the NumPy oracle (and its C port) of the filter recursions without the reference:
fp64 textbook Kalman filter, discrete Riccati steady state, finite-difference Jacobians, and
the committed golden fixtures."""
import numpy as np
import pytest
from scipy.linalg import solve_discrete_are

from oracle import gaussfilt_oracle as go, models as om, threefry as otf, c_oracle
from tests import common as cm

F32 = np.float32


def test_oracle_matches_textbook_kalman_fp64():
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, 1, 300, seed=0)[0]
    post, ll = go.gaussian_sum_filter(cm.oracle_params(a), ys, 1, initial_means=a["m0"][None], return_ll=True)
    GQG = a["G"] @ a["Q"] @ a["G"].T
    fm, fP, pm, pP, l64 = go.textbook_kalman_f64(a["A"], GQG, a["H"], a["R"], a["m0"], a["P0"], ys.astype(np.float64))
    # the reference's 1e-6 jitter on S (utils.py:258) is a ~1e-5 relative perturbation at R = 0.1
    assert cm.rel_err(post.means[0], fm) < 3e-5
    assert cm.rel_err(post.covariances[0], fP) < 3e-5
    assert cm.rel_err(post.predicted_covariances[0], pP) < 3e-5
    assert cm.rel_err(ll[0], l64) < 5e-5
    assert np.all(post.weights == 1.0)


def test_steady_state_matches_riccati():
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, 1, 400, seed=1)[0]
    post = go.gaussian_sum_filter(cm.oracle_params(a), ys, 1, initial_means=a["m0"][None])
    GQG = (a["G"] @ a["Q"] @ a["G"].T).astype(np.float64)
    Pinf = solve_discrete_are(a["A"].T.astype(np.float64), a["H"].T.astype(np.float64), GQG, a["R"].astype(np.float64))
    assert cm.rel_err(post.predicted_covariances[0, -1], Pinf) < 1e-4


def test_psd_solve_is_lu_of_jittered_matrix():
    rng = np.random.default_rng(0)
    for m in (1, 2, 3, 4, 8):
        L = rng.normal(size=(m, m))
        S = (L @ L.T + 0.1 * np.eye(m)).astype(F32)
        b = rng.normal(size=(m, 5)).astype(F32)
        x = go.psd_solve(S, b)
        assert np.allclose(x, go.lu_solve_explicit(S + F32(1e-6), b), rtol=2e-5, atol=1e-6)
        assert np.allclose((S.astype(np.float64) + 1e-6) @ x, b, atol=1e-4)


def test_mvn_log_prob_against_scipy():
    from scipy.stats import multivariate_normal
    rng = np.random.default_rng(1)
    L = rng.normal(size=(3, 3)); S = (L @ L.T + 0.5 * np.eye(3)).astype(F32)
    mu, y = rng.normal(size=3).astype(F32), rng.normal(size=3).astype(F32)
    assert abs(go.mvn_log_prob(mu, S, y) - multivariate_normal(mu, S).logpdf(y)) < 1e-4


@pytest.mark.parametrize("fn,n,nw,u", [
    (om.Lorenz96(8), 8, 8, [0]), (om.Lorenz96(8, mode="as_written"), 8, 8, [0]), (om.Lorenz63(), 3, 3, [0]),
    (om.ManeuverBOT(), 4, 2, [0]), (om.ManeuverBOT(), 4, 2, [1]), (om.ManeuverBOT(), 4, 2, [2]),
    (om.BearingRange(), 4, 2, [0]), (om.Bearing(), 4, 1, [0]), (om.Sine(3), 3, 3, [0]), (om.Quadratic(3, 0.5), 3, 1, [0]),
    (om.Growth(), 1, 1, [0.3]), (om.StochVol(3), 3, 3, [1]), (om.StochVol(3), 3, 3, [0]), (om.PickEven(8), 8, 4, [0])])
def test_analytic_jacobians_match_finite_differences(fn, n, nw, u):
    rng = np.random.default_rng(n + nw)
    x = rng.normal(size=n).astype(F32)
    w = (0.1 * rng.normal(size=nw)).astype(F32)
    uu = np.array(u, F32)
    Jx, Jw = om.finite_difference_jacobians(fn, x, w, uu)
    assert np.allclose(fn.jac_x(x, w, uu), Jx, atol=5e-6 * max(1.0, np.abs(Jx).max()))
    assert np.allclose(fn.jac_noise(x, w, uu), Jw, atol=5e-6 * max(1.0, np.abs(Jw).max()))
    assert np.allclose(fn.value(x, w, uu), om._eval64(fn, x.astype(float), w.astype(float), u), atol=5e-6)


def test_lorenz96_as_written_is_linear():
    f = om.Lorenz96(8, mode="as_written")
    x = np.arange(8, dtype=F32)
    assert np.allclose(f.value(x, np.zeros(8, F32), None), x + 0.01 * (-x + 8.0))


def test_reweight_and_collapse():
    w = go.reweight(np.array([-1000.0, -1001.0], F32), np.array([0.5, 0.5], F32))
    assert np.allclose(w, [1 / (1 + np.exp(-1)), np.exp(-1) / (1 + np.exp(-1))], rtol=1e-6)
    with np.errstate(all="ignore"):
        assert np.isnan(go.reweight(np.array([-np.inf, -np.inf], F32), np.array([0.5, 0.5], F32))).all()
    means = np.array([[0.0, 0.0], [2.0, 0.0]]); covs = np.stack([np.eye(2)] * 2); wv = np.array([0.5, 0.5])
    mu, cov = go.collapse(means, covs, wv)
    assert np.allclose(mu, [1, 0]) and np.allclose(cov, [[2, 0], [0, 1]])


def test_c_port_matches_numpy_oracle():
    for a, B, T in [(cm.cv_model_arrays(), 6, 80), (cm.random_stable_lgssm(8, 4, 3, bias=True), 4, 50),
                    (cm.random_stable_lgssm(3, 3, 5, dq=2, dr=3, bias=True), 4, 50)]:
        ys = cm.simulate_batch(a, B, T, seed=2)
        init = np.tile(a["m0"], (B, 1))
        ref = cm.oracle_kalman_batch(a, ys, init)
        got = c_oracle.kalman_filter(a, ys, init)
        for k in got:
            assert cm.rel_err(got[k], ref[k]) < 3e-6, k


def test_golden_kalman_fixtures_reproduce(golden_dir):
    for name in ("kalman_cv_n4_m2_T64", "kalman_random_n3_m3_T40"):
        d = np.load(f"{golden_dir}/{name}.npz")
        a = {k: d[k] for k in ("A", "G", "H", "D", "Q", "R", "m0", "P0", "q0", "r0")}
        ref = cm.oracle_kalman_batch(a, d["emissions"], d["initial_means"])
        got = c_oracle.kalman_filter(a, d["emissions"], d["initial_means"])
        for k in ref:
            assert np.array_equal(ref[k], d["out_" + k]), (name, k)      # oracle is deterministic
            assert cm.rel_err(got[k], d["out_" + k]) < 3e-6, (name, k)   # C port agrees


def test_golden_gsf_and_bpf_fixtures_reproduce(golden_dir):
    d = np.load(f"{golden_dir}/gsf_lorenz96_matrix_power_n8_K4_T32.npz")
    p = go.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), om.Lorenz96(8), np.zeros(8, F32),
                       1e-2 * np.eye(8, dtype=F32), om.PickEven(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    post = go.gaussian_sum_filter(p, d["emissions"], 4, initial_means=d["initial_means"])
    assert np.array_equal(post.means, d["means"]) and np.array_equal(post.weights, d["weights"])
    assert post.means.shape == (4, 32, 8) and post.covariances.shape == (4, 32, 8, 8)   # (K, T, ...) like :372
    d = np.load(f"{golden_dir}/bpf_lorenz63_N64_T16.npz")
    h = om.Linear(np.eye(3, dtype=F32)); R = 0.5 * np.eye(3, dtype=F32)
    pb = go.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), om.Lorenz63(), np.zeros(3, F32),
                      0.1 * np.eye(3, dtype=F32), h, np.zeros(3, F32), R, go.GaussianEmissionLogProb(h, R))
    out, dbg = go.bootstrap_particle_filter(pb, d["emissions"], 64, key=d["key"], debug=True)
    assert np.array_equal(dbg["ancestors"], d["ancestors"]) and np.array_equal(out["particles"], d["particles"])
    assert out["weights"].shape == (64, 16) and out["particles"].shape == (64, 16, 3)   # (N, T, ...) like :1378


def test_gsf_single_component_equals_kalman_quirks():
    """K = 1 still runs the weight update: weights stay exactly 1; num_iter is ignored (:307)."""
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, 1, 20, seed=3)[0]
    p = cm.oracle_params(a)
    p1 = go.gaussian_sum_filter(p, ys, 1, 1, initial_means=a["m0"][None])
    p5 = go.gaussian_sum_filter(p, ys, 1, 5, initial_means=a["m0"][None])
    assert np.array_equal(p1.means, p5.means) and np.all(p1.weights == 1)
    # default initial means are a PRNGKey(0) draw (:367), not m0
    pd = go.gaussian_sum_filter(p, ys, 1)
    assert not np.array_equal(pd.means[0, 0], p1.means[0, 0])


def test_unscented_filter_is_exact_on_linear_models():
    """The unscented transform reproduces linear maps exactly: on a linear-Gaussian model the oracle's
    unscented Gaussian-sum filter (inference.py:379-456) must agree with its extended one, for any
    ParamsUKF, and with the float64 textbook Kalman filter."""
    a = cm.cv_model_arrays()
    po = cm.oracle_params(a)
    ys = go.sample_ssm(po, otf.PRNGKey(0), 25)[1]
    init = np.zeros((1, 4), np.float32)
    ekf = go.gaussian_sum_filter(po, ys, 1, initial_means=init)
    for up in ((1.0, 0.0, 0.0), (0.5, 2.0, 1.0), (1.0, 2.0, 0.0)):
        ukf = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(*up), ys, 1, initial_means=init)
        for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
            assert cm.rel_err(getattr(ukf, k), getattr(ekf, k)) < 2e-5, (up, k)
    # sigma points: 2 L rows, symmetric about the mean, spread sqrt(L + lambda) * sqrtm(P)
    P = np.array([[2.0, 0.3], [0.3, 1.0]], np.float32)
    m = np.array([1.0, -1.0], np.float32)
    sp = go._get_sigma_points(m, P, 1.0)
    assert sp.shape == (4, 2) and np.allclose(sp[:2] + sp[2:], 2 * m, atol=1e-6)
    L = (sp[:2] - m) / np.sqrt(3.0)
    assert np.allclose(L @ L, P, atol=1e-5) and np.allclose(L, L.T, atol=1e-6)


def _bot_oracle_params():
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], np.float32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(np.float32)
    Q, R = 1e-3 * np.eye(2, dtype=np.float32), np.diag([1e-3, 1e-2]).astype(np.float32)
    return go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, np.float32), Q, om.BearingRange(), np.zeros(2, np.float32), R)


def test_golden_unscented_and_augmented_fixtures_reproduce(golden_dir):
    p = _bot_oracle_params()
    d = np.load(f"{golden_dir}/ugsf_bot_K4_T24.npz")
    post = go.unscented_gaussian_sum_filter(p, go.ParamsUKF(*d["uparams"]), d["emissions"], 4, inputs=d["inputs"],
                                            initial_means=d["initial_means"])
    for k in ("weights", "means", "covariances", "predicted_means", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k), d[k]) < 1e-6, k       # float64 eigh inside: allow the last bit
    assert post.means.shape == (4, 24, 4)
    d = np.load(f"{golden_dir}/agsf_bot_322_T24.npz")
    post, aux = go.speedy_augmented_gaussian_sum_filter(p, d["emissions"], (3, 2, 2), d["key"], 1, tuple(d["opt_args"]),
                                                        d["inputs"], initial_means=d["initial_means"], debug=True)
    assert np.array_equal(post.means, d["means"]) and np.array_equal(post.covariances, d["covariances"])
    assert np.array_equal(aux["pre_weights"], d["pre_weights"])
    assert post.means.shape == (3, 24, 4) and np.allclose(post.weights, 1.0 / 3.0)      # :765 weights = ones / N0
    assert np.allclose(aux["pre_weights"].sum(axis=1), 1.0, atol=1e-6)


def test_c_ports_of_the_gaussian_sum_and_particle_filters_follow_the_numpy_oracle():
    """oracle/c/kf_oracle.c: oracle_gsf_lorenz96_f32 / oracle_bpf_lorenz96_f32 -- the timed CPU baselines of bench.py's
    configs[2] and [3] -- against the NumPy oracle on the same inputs: the EKF bank to rounding; the particle filter with
    identical resampling decisions and ancestors-to-rounding (its sums run sequentially, the NumPy oracle's in tree order)."""
    n, m, K, T, B = 8, 4, 5, 30, 2
    Q, R = 1e-2 * np.eye(n, dtype=F32), 1e-1 * np.eye(m, dtype=F32)
    q0, r0 = np.zeros(n, F32), np.zeros(m, F32)
    fo = om.Lorenz96(n)
    th = np.asarray([fo.alpha, fo.beta, fo.gamma, fo.dt, 1.0], F32)
    po = go.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), fo, q0, Q, om.PickEven(n), r0, R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    im = (8 + np.random.default_rng(0).normal(size=(B, K, n))).astype(F32)
    H = np.zeros((m, n), F32)
    H[np.arange(m), 2 * np.arange(m)] = 1
    got = c_oracle.gsf_lorenz96(th, H, Q, R, q0, r0, ys, im, np.eye(n, dtype=F32))
    for b in range(B):
        ref = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b])
        assert cm.rel_err(got["means"][b], ref.means) < 2e-6 and cm.rel_err(got["covariances"][b], ref.covariances) < 2e-6
        assert np.max(np.abs(got["weights"][b] - ref.weights)) < 1e-5
    # particle filter
    N = 256
    Rl = 0.5 * np.eye(m, dtype=F32)
    pb = go.ParamsBPF(*po[:8], go.GaussianEmissionLogProb(om.PickEven(n), Rl))
    key = np.array([0, 7], np.uint32)
    out = c_oracle.bpf_lorenz96(th, q0, np.diag(Q), np.diag(Rl), 8 * np.ones(n, F32), np.ones(n, F32), ys, N, key)
    for b in range(B):
        ref, dbg = go.bootstrap_particle_filter(pb, ys[b], N, key=key, debug=True)
        assert np.array_equal(out["resampled"][b] > 0.5, dbg["resampled"]) and dbg["resampled"].any()
        mean = np.einsum("itd,it->td", ref["particles"], ref["weights"])
        t_ok = int(np.argmax(np.abs(out["mean"][b] - mean).max(axis=1) > 1e-3)) if (np.abs(out["mean"][b] - mean).max(axis=1) > 1e-3).any() else T
        assert t_ok >= 5                      # (a draw within an ulp of a CDF step flips one ancestor; the runs then part ways)
        assert cm.rel_err(out["mean"][b][:t_ok], mean[:t_ok]) < 1e-4
