"""User-defined f / h from SOURCE TEXT (bf_user_model_create: hiprtc + forward-mode dual numbers), the engine's
counterpart of the arbitrary Python callables and `jacfwd` of the reference (gaussfiltax/models.py:46-49,
gaussfiltax/inference.py:328-329):
* source twins of registry functions (Lorenz-63, the growth model) against the registry functions themselves -- bit for
  bit where the expressions are the same, to rounding where dual-number and hand-derived Jacobians differ in form;
* a model OUTSIDE the registry (pendulum with state-dependent observation noise) against the oracle with analytic Jacobians."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32
FIELDS = ("weights", "means", "covariances", "predicted_means", "predicted_covariances")

L63_SRC = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  const float s = th[0], r = th[1], b = th[2], dt = th[3];
  out[0] = dt * s * (x[1] - x[0]) + x[0] + q[0];
  out[1] = dt * (x[0] * r - x[1] - x[0] * x[2]) + x[1] + q[1];
  out[2] = dt * (x[0] * x[1] - b * x[2]) + x[2] + q[2];
}
"""
QUAD_SRC = """
template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out) {
  T s = x[0] * x[0];
  for (int i = 1; i < BF_N; ++i) s = s + x[i] * x[i];
  out[0] = th[0] * s + r[0];
}
"""
GROWTH_SRC = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  out[0] = x[0] / 2.0f + 25.0f * x[0] / (1.0f + x[0] * x[0]) + u + q[0];
}
"""


class _forced_generic:
    def __enter__(self):
        from bayesianfiltering_amd import _lib
        self.lib = _lib.load()
        _lib.check(self.lib.bf_set_option(b"force_generic", 1))

    def __exit__(self, *exc):
        self.lib.bf_set_option(b"force_generic", 0)


def _bits(t):
    return np.ascontiguousarray(t.cpu().numpy(), F32).view(np.uint32)


def test_lorenz63_source_twin_matches_registry_bit_for_bit():
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    K, T = 3, 40
    Q, R = 0.1 * np.eye(3, dtype=F32), 1.0 * np.eye(1, dtype=F32)
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    th = [10.0, 28.0, 2.667, 0.01]
    reg = bfa.ParamsNLSSM(m0, P0, nl.lorenz63(*th), np.zeros(3, F32), Q, nl.quadratic(3, 0.05), np.zeros(1, F32), R)
    usr = bfa.ParamsNLSSM(m0, P0, nl.user_dynamics(L63_SRC, 3, theta=th), np.zeros(3, F32), Q, nl.quadratic(3, 0.05), np.zeros(1, F32), R)
    po = go.ParamsNLSSM(m0, P0, om.Lorenz63(), np.zeros(3, F32), Q, om.Quadratic(3, 0.05), np.zeros(1, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(2)])
    im = (m0 + 0.5 * np.random.default_rng(0).normal(size=(2, K, 3))).astype(F32)
    with _forced_generic():     # the registry function through the SAME (run-time-dimension) kernel, built ahead of time
        a, la = bfa.gaussian_sum_filter(reg, ys, K, 1, initial_means=im, return_loglik=True)
    b, lb = bfa.gaussian_sum_filter(usr, ys, K, 1, initial_means=im, return_loglik=True)
    for k in FIELDS:
        assert np.array_equal(_bits(getattr(a, k)), _bits(getattr(b, k))), k
    assert np.array_equal(_bits(la), _bits(lb))
    # ... and the register kernel (default path of the registry function) and the oracle to rounding
    c = bfa.gaussian_sum_filter(reg, ys, K, 1, initial_means=im)
    for bb in range(2):
        ref = go.gaussian_sum_filter(po, ys[bb], K, initial_means=im[bb])
        for k in FIELDS:
            assert cm.rel_err(getattr(b, k)[bb].cpu().numpy(), getattr(ref, k)) < 2e-5, k
            assert cm.rel_err(getattr(b, k)[bb].cpu().numpy(), getattr(c, k)[bb].cpu().numpy()) < 2e-5, k
    # user dynamics AND user emission together
    usr2 = usr._replace(emission_function=nl.user_emission(QUAD_SRC, 3, 1, theta=[0.05]))
    # (the source sums x_i^2 with separate multiplies and adds, the registry function with fused multiply-adds: rounding,
    # which 40 steps of the chaotic Lorenz-63 map amplify to ~1e-5 in the mixture weights)
    d = bfa.gaussian_sum_filter(usr2, ys, K, 1, initial_means=im)
    for k in FIELDS:
        assert cm.rel_err(getattr(d, k).cpu().numpy(), getattr(b, k).cpu().numpy()) < (1e-4 if k == "weights" else 2e-5), k


def test_growth_source_twin_matches_registry():
    """f3 of docs/notebooks/Experiment_TSP_2023.ipynb cell 2 (x/2 + 25 x/(1 + x^2) + u + q): same VALUES bit for bit; the
    dual-number derivative 1/2 + 25 (1 + x^2 - 2 x^2) / (1 + x^2)^2 and the hand-derived 25 (1 - x^2) / (1 + x^2)^2 differ
    in form, so covariances agree to rounding."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, K = 60, 4
    Q, R = 1.0 * np.eye(1, dtype=F32), 1.0 * np.eye(1, dtype=F32)
    h = nl.linear_emission(0.8 * np.eye(1, dtype=F32))
    reg = bfa.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), nl.growth(), np.zeros(1, F32), Q, h, np.zeros(1, F32), R)
    usr = reg._replace(dynamics_function=nl.user_dynamics(GROWTH_SRC, 1))
    u = (8.0 * np.cos(1.2 * np.arange(T))).astype(F32)
    po = go.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), om.Growth(), np.zeros(1, F32), Q, om.Linear(0.8 * np.eye(1, dtype=F32)),
                        np.zeros(1, F32), R)
    ys = go.sample_ssm(po, otf.PRNGKey(3), T, u.reshape(T, 1))[1]
    im = np.linspace(-1, 1, K, dtype=F32).reshape(K, 1)
    a = bfa.gaussian_sum_filter(reg, ys, K, 1, u, initial_means=im)
    b = bfa.gaussian_sum_filter(usr, ys, K, 1, u, initial_means=im)
    ref = go.gaussian_sum_filter(po, ys, K, inputs=u.reshape(T, 1), initial_means=im)
    # the values agree bit for bit on the first step (same kernel, same prior, same expression) ...
    with _forced_generic():
        ag = bfa.gaussian_sum_filter(reg, ys, K, 1, u, initial_means=im)
        bg = bfa.gaussian_sum_filter(usr, ys, K, 1, u, initial_means=im)   # (by default a source model of this size runs in registers)
    assert np.array_equal(_bits(ag.predicted_means[:, 0]), _bits(bg.predicted_means[:, 0]))
    for k in FIELDS:
        assert cm.rel_err(getattr(bg, k).cpu().numpy(), getattr(b, k).cpu().numpy()) < 5e-4, k
    # ... and the runs to the conditioning of this model: |f'| reaches 25 and the posterior is bimodal, so the last-bit
    # difference between the two derivative expressions grows over 60 steps
    for k in FIELDS:
        assert cm.rel_err(getattr(b, k).cpu().numpy(), getattr(a, k).cpu().numpy()) < 5e-4, k
        assert cm.rel_err(getattr(b, k).cpu().numpy(), getattr(ref, k)) < 1e-3, k


PENDULUM_DYN = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  const float dt = th[0], g = th[1];
  out[0] = x[0] + dt * x[1] + q[0];
  out[1] = x[1] - dt * g * sin(x[0]) + (1.0f + 0.5f * cos(x[0])) * q[1];      // noise enters through a state-dependent gain
}
"""
PENDULUM_EMI = """
template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out) {
  out[0] = sin(x[0]) + exp(th[0] * x[1]) * r[0];                                // multiplicative observation noise
}
"""


class _PendulumDyn(om.Fn):
    def __init__(self, dt, g):
        self.dt, self.g = F32(dt), F32(g)
        self.out_dim = self.noise_dim = 2

    def value(self, x, w, u):
        return np.array([x[0] + self.dt * x[1] + w[0],
                         x[1] - self.dt * self.g * np.sin(x[0]) + (F32(1) + F32(0.5) * np.cos(x[0])) * w[1]], F32)

    def jac_x(self, x, w, u):
        return np.array([[1, self.dt], [-self.dt * self.g * np.cos(x[0]) - F32(0.5) * np.sin(x[0]) * w[1], 1]], F32)

    def jac_noise(self, x, w, u):
        return np.array([[1, 0], [0, F32(1) + F32(0.5) * np.cos(x[0])]], F32)


class _PendulumEmi(om.Fn):
    def __init__(self, c):
        self.c = F32(c)
        self.out_dim = self.noise_dim = 1

    def value(self, x, w, u):
        return np.array([np.sin(x[0]) + np.exp(self.c * x[1]) * w[0]], F32)

    def jac_x(self, x, w, u):
        return np.array([[np.cos(x[0]), self.c * np.exp(self.c * x[1]) * w[0]]], F32)

    def jac_noise(self, x, w, u):
        return np.array([[np.exp(self.c * x[1])]], F32)


@pytest.mark.parametrize("K", [1, 5])
def test_model_outside_the_registry_against_the_oracle(K):
    """A pendulum with state-dependent process- and observation-noise gains and non-zero noise biases: F_x, F_q, H_x, H_r
    all come from dual numbers and F_q Q F_q^T / H_r R H_r^T are formed on the device every step."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, B = 50, 3
    Q = np.array([[1e-3, 2e-4], [2e-4, 2e-2]], F32)
    R = 5e-2 * np.eye(1, dtype=F32)
    q0, r0 = np.array([0.0, 0.05], F32), np.array([0.1], F32)
    m0, P0 = np.array([1.0, 0.0], F32), 0.1 * np.eye(2, dtype=F32)
    fo, ho = _PendulumDyn(0.05, 9.81), _PendulumEmi(0.2)
    po = go.ParamsNLSSM(m0, P0, fo, q0, Q, ho, r0, R)
    pp = bfa.ParamsNLSSM(m0, P0, nl.user_dynamics(PENDULUM_DYN, 2, theta=[0.05, 9.81], host_fn=lambda x, q, u: fo.value(x, q, u)), q0, Q,
                         nl.user_emission(PENDULUM_EMI, 2, 1, theta=[0.2], host_fn=lambda x, r, u: ho.value(x, r, u)), r0, R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(10 + b), T)[1] for b in range(B)])
    im = (m0 + 0.3 * np.random.default_rng(1).normal(size=(B, K, 2))).astype(F32)
    post, ll = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=im, return_loglik=True)
    for b in range(B):
        ref, rll = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b], return_ll=True)
        for k in FIELDS[1:]:
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 2e-5, (b, k)
        assert np.max(np.abs(post.weights[b].cpu().numpy() - ref.weights)) < 5e-5
        assert cm.rel_err(ll[b].cpu().numpy(), rll) < 5e-5
    # the host twin is callable like a reference lambda
    assert np.allclose(pp.dynamics_function(m0, q0, 0.0), fo.value(m0, q0, np.zeros(1, F32)))
    # per-step covariances (inference.py:21,337-340) on the register kernel of the source model, and two chunks through the carry
    rng = np.random.default_rng(8)
    Qt = np.stack([(0.6 + rng.random()) * Q for _ in range(T)]).astype(F32)
    Rt = np.stack([(0.6 + rng.random()) * R for _ in range(T)]).astype(F32)
    ptv, otv = pp._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt), po._replace(dynamics_noise_covariance=Qt, emission_noise_covariance=Rt)
    tv = bfa.gaussian_sum_filter(ptv, ys, K, 1, initial_means=im)
    for b in range(B):
        ref = go.gaussian_sum_filter(otv, ys[b], K, initial_means=im[b])
        for k in FIELDS[1:]:
            assert cm.rel_err(getattr(tv, k)[b].cpu().numpy(), getattr(ref, k)) < 2e-5, (b, k)
    h1, c1 = bfa.gaussian_sum_filter(pp, ys[:, :20], K, 1, initial_means=im, return_carry=True)
    h2 = bfa.gaussian_sum_filter(pp, ys[:, 20:], K, 1, carry=c1)
    import torch
    assert torch.equal(torch.cat([h1.means, h2.means], dim=2), post.means) and torch.equal(torch.cat([h1.weights, h2.weights], dim=2), post.weights)


def test_user_model_errors_and_cache():
    import time
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    nl = bfa.nonlinearities
    R = np.eye(1, dtype=F32)
    base = bfa.ParamsNLSSM(np.zeros(1, F32), np.eye(1, dtype=F32), nl.growth(), np.zeros(1, F32), np.eye(1, dtype=F32),
                           nl.linear_emission(np.eye(1, dtype=F32)), np.zeros(1, F32), R)
    ys = np.zeros((4, 1), F32)
    bad = "template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) { out[0] = nosuchfn(x[0]); }"
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.gaussian_sum_filter(base._replace(dynamics_function=nl.user_dynamics(bad, 1)), ys, 1, initial_means=np.zeros((1, 1), F32))
    assert e.value.code == _lib.BF_EINVAL and "nosuchfn" in str(e.value)
    with pytest.raises(TypeError):                # a Python callable that leaves NumPy-recordable operations cannot run on the device
        bfa.gaussian_sum_filter(base._replace(dynamics_function=lambda x, q, u: np.asarray([float(x[0])])), ys, 1)
    f = nl.user_dynamics(GROWTH_SRC, 1)
    bfa.gaussian_sum_filter(base._replace(dynamics_function=f), ys, 1, initial_means=np.zeros((1, 1), F32))
    t0 = time.perf_counter()
    for _ in range(20):                                               # compiled once: later calls find the module
        bfa.gaussian_sum_filter(base._replace(dynamics_function=nl.user_dynamics(GROWTH_SRC, 1)), ys, 1, initial_means=np.zeros((1, 1), F32))
    assert time.perf_counter() - t0 < 2.0
    # the augmented filter's extended-Kalman nodes: beside a function from source only the registry's LINEAR one can stand
    bfa.speedy_augmented_gaussian_sum_filter(base._replace(dynamics_function=f), ys, (2, 2, 2), initial_means=np.zeros((2, 1), F32))
    with pytest.raises(_lib.BayesFiltError, match="BOTH functions"):
        bfa.speedy_augmented_gaussian_sum_filter(base._replace(dynamics_function=f, emission_function=nl.quadratic(1, 0.05)), ys, (2, 2, 2),
                                                 initial_means=np.zeros((2, 1), F32))


LINEAR_SRC = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  for (int i = 0; i < BF_N; ++i) {
    T s = th[i * BF_N] * x[0];
    for (int j = 1; j < BF_N; ++j) s = s + th[i * BF_N + j] * x[j];
    out[i] = s + q[i];
  }
}
"""


def test_user_model_with_more_than_64_kib_of_lds():
    """n = 48: the dual-number scratch + the covariance tiles need ~ 91 KiB of dynamic LDS -- the module-function launch above
    the 64 KiB default (launch_user_kernel).  A dense linear map written as source against the oracle's C port."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    n, m, B, T = 48, 4, 3, 12
    a = cm.random_stable_lgssm(n, m, seed=48)
    a["G"] = np.eye(n, dtype=F32)
    from oracle import c_oracle
    ys = cm.simulate_batch(a, B, T, seed=48)
    init = np.tile(a["m0"], (B, 1))
    ref = c_oracle.kalman_filter(a, ys, init)
    reg = cm.product_params(a)
    usr = reg._replace(dynamics_function=nl.user_dynamics(LINEAR_SRC, n, theta=a["A"].reshape(-1)))
    post = bfa.gaussian_sum_filter(usr, ys, 1, 1, initial_means=init.reshape(B, 1, n))
    for k in FIELDS[1:]:
        assert max(cm.both_err(getattr(post, k).cpu().numpy(), ref[k])) < 1e-5, k


def test_user_model_handle_must_match_the_model(monkeypatch):
    """bf_model.user is compiled for fixed (n, dq, m, dr) and for the functions it was given: a handle that does not match
    the bf_model it is attached to is refused with BF_EINVAL before anything is launched (the JIT kernel indexes LDS with
    its compile-time dimensions)."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib, inference
    nl = bfa.nonlinearities
    Q, R = 0.1 * np.eye(3, dtype=F32), np.eye(1, dtype=F32)
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    usr = bfa.ParamsNLSSM(m0, P0, nl.user_dynamics(L63_SRC, 3, theta=[10.0, 28.0, 2.667, 0.01]), np.zeros(3, F32), Q,
                          nl.quadratic(3, 0.05), np.zeros(1, F32), R)
    ys = np.zeros((6, 1), F32)
    im = np.zeros((1, 3), F32)
    bfa.gaussian_sum_filter(usr, ys, 1, initial_means=im)                      # the matching handle runs
    real = inference._compile_user_model
    # (a) a handle compiled for other dimensions
    monkeypatch.setattr(inference, "_compile_user_model", lambda d, e, n, dq, m, dr, lp=None: real(d, e, n + 1, dq + 1, m, dr))
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.gaussian_sum_filter(usr, ys, 1, initial_means=im)
    assert e.value.code == _lib.BF_EINVAL and "compiled for" in str(e.value)
    # (b) dyn_id = BF_FN_USER on a handle that holds an emission only
    monkeypatch.setattr(inference, "_compile_user_model", lambda d, e, n, dq, m, dr, lp=None: real(None, QUAD_SRC, n, dq, m, dr))
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.gaussian_sum_filter(usr, ys, 1, initial_means=im)
    assert e.value.code == _lib.BF_EINVAL and "without dynamics source" in str(e.value)
    # (c) a handle that holds a function the model does not ask for
    monkeypatch.setattr(inference, "_compile_user_model", lambda d, e, n, dq, m, dr, lp=None: real(L63_SRC, QUAD_SRC, n, dq, m, dr))
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.gaussian_sum_filter(usr, ys, 1, initial_means=im)
    assert e.value.code == _lib.BF_EINVAL and "emi_id must be BF_FN_USER" in str(e.value)


def test_corrupt_jit_cache_file_is_rebuilt(tmp_path):
    """A truncated code object under the cache's final name (a crashed writer, a stale file from another ROCm) is deleted
    and recompiled, not reported for good.  Child processes: the cache directory is read from the environment at first use."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import ctypes as C, sys
sys.path.insert(0, {root!r})
from bayesianfiltering_amd import _lib
lib = _lib.require_gpu()
src = b"template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {{ out[0] = x[0] * 0.5f + q[0]; }}"
h = C.c_void_p()
rc = lib.bf_user_model_create(src, None, 1, 1, 1, 1, C.byref(h))
assert rc == _lib.BF_OK, (rc, lib.bf_last_error())
print("created")
"""
    env = dict(os.environ, BAYESFILT_CACHE_DIR=str(tmp_path))
    run = lambda: subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    r = run()
    assert r.returncode == 0 and "created" in r.stdout, (r.stdout, r.stderr)
    files = [f for f in os.listdir(tmp_path) if f.endswith(".co")]
    assert len(files) == 1 and not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    good = (tmp_path / files[0]).read_bytes()
    (tmp_path / files[0]).write_bytes(good[: len(good) // 3])
    r = run()
    assert r.returncode == 0 and "created" in r.stdout, (r.stdout, r.stderr)
    assert (tmp_path / files[0]).read_bytes() == good


# ---------------------------------------------------------------------------------------------------------------------
# The bootstrap particle filter with functions from source (bf_user_model_create_lp): x' = f(x, q, u) and
# emission_distribution_log_prob(x', y, u) are arbitrary callables in the reference (gaussfiltax/models.py:73-84,
# inference.py:1344-1349).

# fBOT / gBOT of docs/experiments/BOT_Experiment_script.py:31-44, operation for operation as the registry functions evaluate
# them (csrc/ssm_device.hpp: DYN_MANEUVER_BOT, EMI_BEARING_RANGE) -- in the particle-filter kernels sin / cos / atan2 / sqrt
# of a source function ARE the canonical arithmetic, so the twin reproduces the registry function's bits
BOT_DYN_SRC = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u0, const float* th, T* out) {
  const float dt = th[0], acc = th[1];
  const T c0 = 0.5f * (u0 - 1.f) * (u0 - 2.f), c1 = -u0 * (u0 - 2.f), c2 = 0.5f * u0 * (u0 - 1.f);
  T Mx[16] = {c0, c0 * dt, 0, 0, 0, c0, 0, 0, 0, 0, c0, c0 * dt, 0, 0, 0, c0};
  const T nrm = sqrt(x[1] * x[1] + x[3] * x[3]);
  T sn0, cs0;
  sincos(dt * (0.1f * acc / nrm), &sn0, &cs0);
  for (int sgn = 0; sgn < 2; ++sgn) {
    const T cc = sgn == 0 ? c1 : c2;
    const T om = 0.1f * (sgn == 0 ? acc : -acc) / nrm;
    const T sn = sgn == 0 ? sn0 : -sn0, cs = cs0;
    const T so = sn / om, co = (1.f - cs) / om;
    const T Fm[16] = {1, so, 0, -co, 0, cs, 0, -sn, 0, co, 1, so, 0, sn, 0, cs};
    for (int i = 0; i < 16; ++i) Mx[i] += cc * Fm[i];
  }
  const float G[8] = {0.5f, 0, 1, 0, 0, 0.5f, 0, 1};
  for (int i = 0; i < 4; ++i) {
    T s = Mx[i * 4] * x[0];
    for (int k = 1; k < 4; ++k) s = fma(Mx[i * 4 + k], x[k], s);
    T g = 0.f;
    for (int k = 0; k < 2; ++k) g = fma(G[i * 2 + k], q[k], g);
    out[i] = s + g;
  }
}
"""
BOT_EMI_SRC = """
template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out) {
  out[0] = atan2(x[2], x[0]) + r[0];
  out[1] = sqrt(x[0] * x[0] + x[2] * x[2]) + r[1];
}
"""
L63_LP_SRC = """
template <class T> __device__ T log_prob(const T* x, const float* y, T u, const float* th) {   // Laplace density around c |x|^2
  T s = x[0] * x[0];
  for (int i = 1; i < BF_N; ++i) s = fma(x[i], x[i], s);
  const T d = abs(y[0] - th[0] * s);
  return -d / th[1] - log(2.0f * th[1]);
}
"""


def _bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a.cpu().numpy(), F32).view(np.uint32), np.ascontiguousarray(b.cpu().numpy(), F32).view(np.uint32))


@pytest.mark.parametrize("N", [100, 1000, 4096, 10000, 70000])
def test_particle_filter_bot_source_twins_match_the_registry_bit_for_bit(N):
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, B = 24, 2
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    r0 = np.array([0.01, -0.02], F32)
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32)
    po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), r0, R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(10 + b), T, inputs.reshape(T, 1))[1] for b in range(B)])
    g_reg = nl.bearing_range()
    reg = bfa.ParamsBPF(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, g_reg, r0, R, nl.gaussian_log_prob(g_reg, R, r0))
    f_usr = nl.user_dynamics(BOT_DYN_SRC, 4, noise_dim=2, theta=nl.maneuver_bot().theta)
    g_usr = nl.user_emission(BOT_EMI_SRC, 4, 2)
    key = np.array([0, 5], np.uint32)
    # (bpf_hbm_mode = 1: the registry run on the one-workgroup-per-trajectory kernel too -- for few trajectories with thousands
    # of particles it would otherwise spread the particles over the chip, where the unpinned mean summary sums in another order)
    a = bfa.bootstrap_particle_filter(reg, ys, N, key, inputs, output="both", return_ancestors=True, options={"bpf_hbm_mode": 1})
    assert float(a["resampled"].mean()) > 0
    for usr in (reg._replace(dynamics_function=f_usr),                                                               # f from source
                reg._replace(emission_function=g_usr, emission_distribution_log_prob=nl.gaussian_log_prob(g_usr, R, r0)),   # h from source
                reg._replace(dynamics_function=f_usr, emission_function=g_usr,
                             emission_distribution_log_prob=nl.gaussian_log_prob(g_usr, R, r0))):                  # both
        b_ = bfa.bootstrap_particle_filter(usr, ys, N, key, inputs, output="both", return_ancestors=True)
        for k in ("weights", "particles", "mean", "ess", "logz", "resampled"):
            if k == "mean" and N > 4096:   # (beyond 4 096 particles the build from source keeps them in HBM -- the reference's 10^4 ... 5 10^5
                # particle runs -- while the registry twin may still fit its registers: the summary's sum runs in another order)
                assert cm.rel_err(b_[k].cpu().numpy(), a[k].cpu().numpy()) < 1e-6
                continue
            assert _bits_eq(a[k], b_[k]), k
        assert np.array_equal(a["ancestors"].cpu().numpy(), b_["ancestors"].cpu().numpy())


def test_particle_filter_with_a_log_density_from_source():
    """A non-Gaussian emission density (Laplace around c |x|^2) on Lorenz-63 dynamics written as source: against a NumPy
    restatement of the recursion run on the engine's own particles (weights one step at a time), and chunked == one shot."""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    N, T, B = 512, 12, 2
    Q = 0.1 * np.eye(3, dtype=F32)
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    th = [10.0, 28.0, 2.667, 0.01]
    c, scale = 0.05, 0.7
    f_usr = nl.user_dynamics(L63_SRC, 3, theta=th)
    lp = nl.user_log_prob(L63_LP_SRC, theta=[c, scale])
    pp = bfa.ParamsBPF(m0, P0, f_usr, np.zeros(3, F32), Q, nl.quadratic(3, c), np.zeros(1, F32), np.eye(1, dtype=F32), lp)
    po = go.ParamsNLSSM(m0, P0, om.Lorenz63(), np.zeros(3, F32), Q, om.Quadratic(3, c), np.zeros(1, F32), np.eye(1, dtype=F32))
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    key = np.array([0, 9], np.uint32)
    out, carry = bfa.bootstrap_particle_filter(pp, ys, N, key, output="both", ess_threshold=0.0, return_carry=True)   # no resampling: weights compound
    x = out["particles"].cpu().numpy().astype(np.float64)       # (B, N, T, 3)
    w = out["weights"].cpu().numpy().astype(np.float64)         # (B, N, T)
    for b in range(B):
        wprev = np.full(N, 1.0 / N)
        for t in range(T):
            s = np.sum(x[b, :, t] ** 2, axis=1)
            ll = -np.abs(ys[b, t, 0] - c * s) / scale - np.log(2 * scale)
            wn = np.exp(ll - ll.max()) * wprev
            wn /= wn.sum()
            assert np.max(np.abs(wn - w[b, :, t])) < 2e-5 * max(1.0, w[b, :, t].max() * N) / N + 1e-7, (b, t)
            wprev = w[b, :, t]
    # the registry Lorenz-63 with the same density gives the same bits (the source twin of f)
    reg = pp._replace(dynamics_function=nl.lorenz63(*th))
    out2 = bfa.bootstrap_particle_filter(reg, ys, N, key, output="both", ess_threshold=0.0)
    assert _bits_eq(out["particles"], out2["particles"]) and _bits_eq(out["weights"], out2["weights"])
    # with resampling, in two chunks through the carry
    one, c1 = bfa.bootstrap_particle_filter(pp, ys, N, key, output="both", return_carry=True)
    h1, cc = bfa.bootstrap_particle_filter(pp, ys[:, :5], N, key, output="both", return_carry=True)
    h2, c2 = bfa.bootstrap_particle_filter(pp, ys[:, 5:], N, None, output="both", carry=cc, return_carry=True)
    assert float(one["resampled"].mean()) > 0
    assert torch.equal(torch.cat([h1["weights"], h2["weights"]], dim=2), one["weights"])
    assert torch.equal(torch.cat([h1["particles"], h2["particles"]], dim=2), one["particles"])
    assert torch.equal(c2.particles, c1.particles)


QUAD_FMA_SRC = """
template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out) {   // c |x|^2 + r, the registry's fma chain
  T s = 0.f;
  for (int i = 0; i < BF_N; ++i) s = fma(x[i], x[i], s);
  out[0] = th[0] * s + r[0];
}
"""


@pytest.mark.parametrize("K", [1, 5, 100])
def test_unscented_filter_with_functions_from_source(K):
    """unscented_gaussian_sum_filter (gaussfiltax/inference.py:379-456) pushes sigma points through ARBITRARY f(x, q, u),
    h(x, r, u) (:146-224): Lorenz-63 dynamics and the quadratic emission of docs/experiments/exp_lorentz63.py as source
    strings, compiled at run time into the unscented kernel, against their registry twins and against the oracle.
    * twins agree to the last few ulps (<= 1e-5 relative over 30 steps of the chaotic map, measured 6e-6), not bit for bit: the unscented kernel's sigma-point algebra is compiled with
      floating-point contraction, and the compiler's fusion choices depend on the code inlined into it (measured: keeping f out
      of line makes dynamics twins bit-identical over 6 steps at 1.5 x the registry path's run time; the particle filter,
      whose weight path is contraction-free by definition, IS bit-identical to its twins);
    * per-step covariances and two chunks through the carry on the source build."""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, B = 30, 2
    Q, R = 0.1 * np.eye(3, dtype=F32), 1.0 * np.eye(1, dtype=F32)
    m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
    th = [10.0, 28.0, 2.667, 0.01]
    reg = bfa.ParamsNLSSM(m0, P0, nl.lorenz63(*th), np.zeros(3, F32), Q, nl.quadratic(3, 0.05), np.zeros(1, F32), R)
    po = go.ParamsNLSSM(m0, P0, om.Lorenz63(), np.zeros(3, F32), Q, om.Quadratic(3, 0.05), np.zeros(1, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
    im = (m0 + 0.5 * np.random.default_rng(K).normal(size=(B, K, 3))).astype(F32)
    up = bfa.ParamsUKF(1, 0, 0)
    f_usr = nl.user_dynamics(L63_SRC, 3, theta=th)
    h_usr = nl.user_emission(QUAD_FMA_SRC, 3, 1, theta=[0.05])
    a, la, ca = bfa.unscented_gaussian_sum_filter(reg, up, ys, K, 1, initial_means=im, return_loglik=True, return_carry=True)
    # f, h or both from source: to rounding
    for usr in (reg._replace(emission_function=h_usr), reg._replace(dynamics_function=f_usr),
                reg._replace(dynamics_function=f_usr, emission_function=h_usr)):
        b_, lb = bfa.unscented_gaussian_sum_filter(usr, up, ys, K, 1, initial_means=im, return_loglik=True)
        for k in FIELDS[1:]:
            assert cm.rel_err(getattr(b_, k).cpu().numpy(), getattr(a, k).cpu().numpy()) < 1e-5, k
        assert np.max(np.abs(b_.weights.cpu().numpy() - a.weights.cpu().numpy())) < 2e-5   # (mixture weights amplify: exp of a log-likelihood difference)
        assert cm.rel_err(lb.cpu().numpy(), la.cpu().numpy()) < 1e-5
    if K <= 5:
        for bb in range(B):
            ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(1, 0, 0), ys[bb], K, initial_means=im[bb])
            for k in FIELDS[1:]:   # (the first 12 steps: the chaotic map amplifies rounding beyond 1e-5 over 30, oracle against itself too)
                assert cm.rel_err(getattr(b_, k)[bb].cpu().numpy()[:, :12], getattr(ref, k)[:, :12]) < 1e-5, k
    # per-step covariances and two chunks through the carry, on the source build
    rng = np.random.default_rng(3)
    Qt = np.stack([(0.6 + rng.random()) * Q for _ in range(T)]).astype(F32)
    c1 = bfa.unscented_gaussian_sum_filter(usr._replace(dynamics_noise_covariance=Qt), up, ys, K, 1, initial_means=im)
    c2 = bfa.unscented_gaussian_sum_filter(reg._replace(dynamics_noise_covariance=Qt), up, ys, K, 1, initial_means=im)
    assert cm.rel_err(c1.covariances.cpu().numpy(), c2.covariances.cpu().numpy()) < 1e-5 and not torch.equal(c2.covariances, a.covariances)
    one = bfa.unscented_gaussian_sum_filter(usr, up, ys, K, 1, initial_means=im)
    h1, cc = bfa.unscented_gaussian_sum_filter(usr, up, ys[:, :11], K, 1, initial_means=im, return_carry=True)
    h2 = bfa.unscented_gaussian_sum_filter(usr, up, ys[:, 11:], K, 1, carry=cc)
    assert torch.equal(torch.cat([h1.means, h2.means], dim=2), one.means) and torch.equal(torch.cat([h1.weights, h2.weights], dim=2), one.weights)


@pytest.mark.parametrize("nodes,nc", [("ekf", (4, 3, 2)), ("ukf", (4, 3, 2)), ("ekf", (20, 3, 3)), ("ukf", (20, 3, 3))])
def test_augmented_filter_with_functions_from_source(nodes, nc):
    """The augmented Gaussian-sum filters (gaussfiltax/inference.py:458-1300) take arbitrary f, h like every other filter: the
    manoeuvring-target dynamics and the bearing-range emission of BOT_Experiment_script.py:31-44 as source strings, compiled at
    run time into the augmented kernel -- extended-Kalman nodes with the Jacobians by dual numbers (jacfwd w.r.t. state and
    noise), unscented nodes with the sigma points through the caller's functions -- against their registry twins (analytic
    Jacobians) and against the oracle; a tree wider than a wave; per-step covariances; two chunks through the carry."""
    import torch
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, B = 20, 3
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    r0 = np.array([0.01, -0.02], F32)
    inputs = np.array([1] * 7 + [0] * 7 + [2] * 6, F32)
    po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), r0, R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(30 + b), T, inputs.reshape(T, 1))[1] for b in range(B)])
    reg = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), r0, R)
    f_usr = nl.user_dynamics(BOT_DYN_SRC, 4, noise_dim=2, theta=nl.maneuver_bot().theta)
    g_usr = nl.user_emission(BOT_EMI_SRC, 4, 2)
    usr = reg._replace(dynamics_function=f_usr, emission_function=g_usr)
    up = bfa.ParamsUKF(1, 0, 0)
    im = (mu0 + 0.05 * np.random.default_rng(1).normal(size=(B, nc[0], 4))).astype(F32)

    def run(p, y=ys, u=inputs, **kw):
        if nodes == "ekf":
            return bfa.speedy_augmented_gaussian_sum_filter(p, y, nc, None, 1, (0.1, 0.1), u, initial_means=im, return_leaf_indices=True, **kw)
        return bfa.speedy_unscented_agsf(p, up, y, nc, None, 1, (0.1, 0.1), u, initial_means=im, return_leaf_indices=True, **kw)

    a, aa = run(reg)
    b_, ab = run(usr)
    # the drawn leaves are the same; the moments agree to rounding (analytic Jacobians against dual numbers / the same sigma points)
    assert np.array_equal(aa["leaf_indices"].cpu().numpy(), ab["leaf_indices"].cpu().numpy())
    for k in ("weights", "means", "covariances"):
        assert cm.rel_err(getattr(b_, k).cpu().numpy(), getattr(a, k).cpu().numpy()) < 1e-5, k
    if nodes == "ukf":   # one function from source, the other from the registry
        c_, _ = run(reg._replace(emission_function=g_usr))
        assert cm.rel_err(c_.means.cpu().numpy(), a.means.cpu().numpy()) < 1e-5
    else:
        with pytest.raises(Exception, match="BOTH functions"):
            run(reg._replace(emission_function=g_usr))
    # the oracle, one trajectory
    if nc[0] * nc[1] * nc[2] <= 64:
        if nodes == "ekf":
            ref, _ = go.speedy_augmented_gaussian_sum_filter(po, ys[0], nc, None, 1, (0.1, 0.1), inputs.reshape(T, 1), initial_means=im[0])
        else:
            ref, _ = go.speedy_unscented_agsf(po, go.ParamsUKF(1, 0, 0), ys[0], nc, None, 1, (0.1, 0.1), inputs.reshape(T, 1), initial_means=im[0])
        for k in ("means", "covariances"):
            assert cm.rel_err(getattr(b_, k)[0].cpu().numpy(), getattr(ref, k)) < 2e-5, k
    # per-step covariances
    rng = np.random.default_rng(5)
    Rt = np.stack([(0.6 + rng.random()) * R for _ in range(T)]).astype(F32)
    c1, _ = run(usr._replace(emission_noise_covariance=Rt))
    c2, _ = run(reg._replace(emission_noise_covariance=Rt))
    # (1e-4: the carried covariances shrink to 1e-4 of their first value over these steps and the two builds round differently)
    assert cm.rel_err(c1.covariances.cpu().numpy(), c2.covariances.cpu().numpy()) < 1e-4 and not torch.equal(c2.covariances, a.covariances)
    # two chunks through the carry == one launch
    h1, x1 = run(usr, y=ys[:, :9], u=inputs[:9], return_carry=True)
    im_saved = im
    if nodes == "ekf":
        h2, _ = bfa.speedy_augmented_gaussian_sum_filter(usr, ys[:, 9:], nc, None, 1, (0.1, 0.1), inputs[9:], carry=x1["carry"])
    else:
        h2, _ = bfa.speedy_unscented_agsf(usr, up, ys[:, 9:], nc, None, 1, (0.1, 0.1), inputs[9:], carry=x1["carry"])
    assert torch.equal(torch.cat([h1.means, h2.means], dim=2), b_.means)


def test_sampling_with_functions_from_source():
    """NonlinearSSM.sample (gaussfiltax/models.py:240-289) takes the model's f(x, q, u) and h(x, r, u) like the filters do: the
    manoeuvring-target / bearing-range model of BOT_Experiment_script.py:31-44 as source -- f, h or both -- draws the trajectories
    its registry twin draws (same Threefry stream, same canonical arithmetic: bit for bit) and the oracle's to rounding; the
    pendulum with state-dependent noise gains (outside the registry) against the oracle."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T = 24
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    r0 = np.array([0.01, -0.02], F32)
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32)
    reg = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), r0, R)
    f_usr = nl.user_dynamics(BOT_DYN_SRC, 4, noise_dim=2, theta=nl.maneuver_bot().theta)
    g_usr = nl.user_emission(BOT_EMI_SRC, 4, 2)
    keys = np.stack([otf.PRNGKey(21), otf.PRNGKey(22), otf.split(otf.PRNGKey(3), 3)[1]])
    model = bfa.NonlinearSSM(4, 2, 2, 2)
    xa, ya = model.sample(reg, keys, T, inputs)
    for usr in (reg._replace(dynamics_function=f_usr), reg._replace(emission_function=g_usr),
                reg._replace(dynamics_function=f_usr, emission_function=g_usr)):
        xb, yb = model.sample(usr, keys, T, inputs)
        assert _bits_eq(xa, xb) and _bits_eq(ya, yb)
    po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), r0, R)
    for b in range(3):
        xs, ys = go.sample_ssm(po, keys[b], T, inputs.reshape(T, 1))
        assert cm.rel_err(xb[b].cpu().numpy(), xs) < 1e-5 and cm.rel_err(yb[b].cpu().numpy(), ys) < 1e-5
    # a model outside the registry
    Qp = np.array([[1e-3, 2e-4], [2e-4, 2e-2]], F32)
    Rp = 5e-2 * np.eye(1, dtype=F32)
    q0, r0p = np.array([0.0, 0.05], F32), np.array([0.1], F32)
    m0, P0 = np.array([1.0, 0.0], F32), 0.1 * np.eye(2, dtype=F32)
    fo, ho = _PendulumDyn(0.05, 9.81), _PendulumEmi(0.2)
    pp = bfa.ParamsNLSSM(m0, P0, nl.user_dynamics(PENDULUM_DYN, 2, theta=[0.05, 9.81]), q0, Qp, nl.user_emission(PENDULUM_EMI, 2, 1, theta=[0.2]), r0p, Rp)
    xs, ys = bfa.NonlinearSSM(2, 2, 1, 1).sample(pp, keys, 40)
    for b in range(3):
        xr, yr = go.sample_ssm(go.ParamsNLSSM(m0, P0, fo, q0, Qp, ho, r0p, Rp), keys[b], 40)
        assert cm.rel_err(xs[b].cpu().numpy(), xr) < 2e-5 and cm.rel_err(ys[b].cpu().numpy(), yr) < 2e-5


def test_registry_models_at_dimensions_without_a_compiled_instance():
    """The particle / unscented / augmented kernels hold compiled instances for the dimensions of the reference's experiments; any
    other (n, dq, m, dr) used to end in BF_EUNSUPPORTED ("not compiled in").  The same kernels are now compiled at run time for
    such a registry model (the from-source machinery on an internal handle): Lorenz-96 with n = 10 / 5 in the particle and
    unscented filters, a 5-state linear model in the unscented and both augmented filters -- against the oracle."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    # ---- particle filter: Lorenz-96 n = 10, m = 5 (no (10, 10, 5) instance)
    n, m, N, T = 10, 5, 256, 10
    R = 0.5 * np.eye(m, dtype=F32)
    g = nl.pick_even(n)
    pp = bfa.ParamsBPF(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), 1e-2 * np.eye(n, dtype=F32), g,
                       np.zeros(m, F32), R, nl.gaussian_log_prob(g, R))
    po = go.ParamsBPF(*pp[:2], om.Lorenz96(n), pp[3], pp[4], om.PickEven(n), pp[6], pp[7], go.GaussianEmissionLogProb(om.PickEven(n), R))
    ys = np.stack([go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(60 + b), T)[1] for b in range(2)])
    key = np.array([0, 3], np.uint32)
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, output="both", return_ancestors=True)
    for b in range(2):
        ref, dbg = go.bootstrap_particle_filter(po, ys[b], N, key=key, debug=True, arith="canonical")
        assert np.array_equal(out["ancestors"][b].cpu().numpy().T, dbg["ancestors"]) and dbg["resampled"].any()
        assert np.array_equal(np.ascontiguousarray(out["weights"][b].cpu().numpy(), F32).view(np.uint32), ref["weights"].view(np.uint32))
        assert np.array_equal(np.ascontiguousarray(out["particles"][b].cpu().numpy(), F32).view(np.uint32), ref["particles"].view(np.uint32))
    # ---- unscented Gaussian-sum filter: Lorenz-96 n = 6, m = 3 (no (6, 6, 3, 3) instance)
    n, m, K, T = 6, 3, 5, 15
    pn = bfa.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), 1e-2 * np.eye(n, dtype=F32), nl.pick_even(n),
                         np.zeros(m, F32), 1e-1 * np.eye(m, dtype=F32))
    on = go.ParamsNLSSM(*pn[:2], om.Lorenz96(n), pn[3], pn[4], om.PickEven(n), pn[6], pn[7])
    ys = np.stack([go.sample_ssm(on, otf.PRNGKey(70 + b), T)[1] for b in range(2)])
    im = (8 + np.random.default_rng(2).normal(size=(2, K, n))).astype(F32)
    post = bfa.unscented_gaussian_sum_filter(pn, bfa.ParamsUKF(1, 0, 0), ys, K, 1, initial_means=im)
    for b in range(2):
        ref = go.unscented_gaussian_sum_filter(on, go.ParamsUKF(1, 0, 0), ys[b], K, initial_means=im[b])
        for k in ("means", "covariances", "predicted_means"):
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < 2e-5, k
    # ---- a 5-state linear model in the unscented and the augmented filters
    rng = np.random.default_rng(5)
    n, m, T, nc = 5, 2, 12, (3, 2, 2)
    A = (0.9 * np.eye(n) + 0.03 * rng.normal(size=(n, n))).astype(F32)
    H = rng.normal(size=(m, n)).astype(F32)
    pl = bfa.ParamsNLSSM(np.zeros(n, F32), np.eye(n, dtype=F32), nl.linear_dynamics(A), np.zeros(n, F32), 0.05 * np.eye(n, dtype=F32),
                         nl.linear_emission(H), np.zeros(m, F32), 0.2 * np.eye(m, dtype=F32))
    ol = go.ParamsNLSSM(*pl[:2], om.Linear(A), pl[3], pl[4], om.Linear(H), pl[6], pl[7])
    ys = go.sample_ssm(ol, otf.PRNGKey(80), T)[1]
    im = rng.normal(size=(nc[0], n)).astype(F32)
    for fn_e, fn_o, extra in ((bfa.speedy_augmented_gaussian_sum_filter, go.speedy_augmented_gaussian_sum_filter, ()),
                              (bfa.speedy_unscented_agsf, go.speedy_unscented_agsf, (1,))):
        a_e = (pl,) + ((bfa.ParamsUKF(1, 0, 0),) if extra else ()) + (ys, nc)
        a_o = (ol,) + ((go.ParamsUKF(1, 0, 0),) if extra else ()) + (ys, nc)
        post, _ = fn_e(*a_e, initial_means=im)
        ref, _ = fn_o(*a_o, initial_means=im)
        for k in ("means", "covariances"):
            assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 3e-5, (fn_e.__name__, k)


def test_filters_take_python_functions_of_numpy_operations():
    """The reference's call sites pass lambdas (docs/experiments/BOT_Experiment_script.py:31-44, :100, :151).  Written with numpy
    instead of jax.numpy they are accepted as they are: recorded once on symbolic arguments (bayesianfiltering_amd/trace.py),
    turned into source, compiled at run time -- dynamics, emission and the particle filter's log-density -- and every filter agrees
    with the registry twin; a function that cannot be recorded says why."""
    import bayesianfiltering_amd as bfa
    from tests.test_trace import f_bot, h_bot, DT, ACC
    nl = bfa.nonlinearities
    T, B = 20, 3
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    inputs = np.array([1] * 7 + [0] * 7 + [2] * 6, F32)
    reg = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(DT, ACC), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R)
    lam = bfa.ParamsNLSSM(mu0, S0, f_bot, np.zeros(2, F32), Q, h_bot, np.zeros(2, F32), R)          # plain Python functions
    po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(DT, ACC), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R)
    ys = np.stack([go.sample_ssm(po, otf.PRNGKey(40 + b), T, inputs.reshape(T, 1))[1] for b in range(B)])
    K = 4
    im = (mu0 + 0.05 * np.random.default_rng(3).normal(size=(B, K, 4))).astype(F32)
    a = bfa.gaussian_sum_filter(reg, ys, K, 1, inputs, initial_means=im)
    b_ = bfa.gaussian_sum_filter(lam, ys, K, 1, inputs, initial_means=im)
    for k in FIELDS:
        assert cm.rel_err(getattr(b_, k).cpu().numpy(), getattr(a, k).cpu().numpy()) < 2e-5, k
    up = bfa.ParamsUKF(1, 0, 0)
    ua = bfa.unscented_gaussian_sum_filter(reg, up, ys, K, 1, inputs, initial_means=im)
    ub = bfa.unscented_gaussian_sum_filter(lam, up, ys, K, 1, inputs, initial_means=im)
    assert cm.rel_err(ub.means.cpu().numpy(), ua.means.cpu().numpy()) < 2e-5
    ga, _ = bfa.speedy_augmented_gaussian_sum_filter(reg, ys, (K, 2, 2), None, 1, (0.1, 0.1), inputs, initial_means=im)
    gb, _ = bfa.speedy_augmented_gaussian_sum_filter(lam, ys, (K, 2, 2), None, 1, (0.1, 0.1), inputs, initial_means=im)
    assert cm.rel_err(gb.means.cpu().numpy(), ga.means.cpu().numpy()) < 5e-5
    # the particle filter with the reference's Python log-density
    from bayesianfiltering_amd.distributions import MVN      # (the reference: tfp's MultivariateNormalFullCovariance as MVN)
    glp = lambda x, y, u: MVN(loc=h_bot(x, np.zeros(2), u), covariance_matrix=R).log_prob(y)     # BOT_Experiment_script.py:45, as written
    g_reg = nl.bearing_range()
    pa = bfa.ParamsBPF(*reg, nl.gaussian_log_prob(g_reg, R))
    pb = bfa.ParamsBPF(*lam, glp)
    key = np.array([0, 9], np.uint32)
    xa = bfa.bootstrap_particle_filter(pa, ys, 500, key, inputs, output="summary")
    xb = bfa.bootstrap_particle_filter(pb, ys, 500, key, inputs, output="summary")
    d = np.abs(xa["mean"].cpu().numpy() - xb["mean"].cpu().numpy()).max(axis=2) / np.abs(xa["mean"].cpu().numpy()).max()
    assert (d[:, :3] < 1e-4).all() and float(xb["resampled"].mean()) > 0     # (the same filter until an ancestor differs by rounding)
    # sampling
    keys = np.stack([otf.PRNGKey(1), otf.PRNGKey(2)])
    sa = bfa.NonlinearSSM(4, 2, 2, 2).sample(reg, keys, T, inputs)
    sb = bfa.NonlinearSSM(4, 2, 2, 2).sample(lam, keys, T, inputs)
    assert cm.rel_err(sb[0].cpu().numpy(), sa[0].cpu().numpy()) < 1e-5 and cm.rel_err(sb[1].cpu().numpy(), sa[1].cpu().numpy()) < 1e-5
    # a function that cannot be recorded
    with pytest.raises(TypeError, match="truth value"):
        bfa.gaussian_sum_filter(lam._replace(dynamics_function=lambda x, q, u: x if x[0] > 0 else -x), ys, K, 1, inputs, initial_means=im)


def test_readme_call_site():
    """The ported call site of README.md, as written there: lambdas of numpy operations, one trajectory and a batch."""
    import bayesianfiltering_amd as gf
    from bayesianfiltering_amd import ParamsNLSSM
    m0, P0 = np.array([1.0, 0.0], F32), 0.1 * np.eye(2, dtype=F32)
    Q, R = np.diag([1e-3, 2e-2]).astype(F32), 5e-2 * np.eye(1, dtype=F32)
    f = lambda x, q, u: np.array([x[0] + 0.05 * x[1], x[1] - 0.05 * 9.81 * np.sin(x[0])]) + q
    h = lambda x, r, u: np.array([np.sin(x[0])]) + r
    params = ParamsNLSSM(m0, P0, f, np.zeros(2), Q, h, np.zeros(1), R)
    fo = _PendulumPlain()
    po = go.ParamsNLSSM(m0, P0, fo, np.zeros(2, F32), Q, _SinEmi(), np.zeros(1, F32), R)
    emissions = go.sample_ssm(po, otf.PRNGKey(5), 40)[1]
    post = gf.gaussian_sum_filter(params, emissions, 5, 1)
    assert tuple(post.means.shape) == (5, 40, 2)
    im = gf.sample_initial_component_means(params, 5)
    ref = go.gaussian_sum_filter(po, emissions, 5, initial_means=np.asarray(im))
    assert cm.rel_err(post.means.cpu().numpy(), ref.means) < 2e-5 and cm.rel_err(post.covariances.cpu().numpy(), ref.covariances) < 2e-5
    batch = gf.gaussian_sum_filter(params, np.stack([emissions, emissions]), 5, 1)
    assert tuple(batch.means.shape) == (2, 5, 40, 2) and np.array_equal(batch.means[1].cpu().numpy(), post.means.cpu().numpy())


class _PendulumPlain(om.Fn):
    out_dim, noise_dim = 2, 2

    def value(self, x, w, u):
        return np.array([x[0] + F32(0.05) * x[1], x[1] - F32(0.05 * 9.81) * np.sin(x[0])], F32) + w

    def jac_x(self, x, w, u):
        return np.array([[1, 0.05], [-0.05 * 9.81 * np.cos(x[0]), 1]], F32)

    def jac_noise(self, x, w, u):
        return np.eye(2, dtype=F32)


class _SinEmi(om.Fn):
    out_dim, noise_dim = 1, 1

    def value(self, x, w, u):
        return np.array([np.sin(x[0])], F32) + w

    def jac_x(self, x, w, u):
        return np.array([[np.cos(x[0]), 0]], F32)

    def jac_noise(self, x, w, u):
        return np.eye(1, dtype=F32)
