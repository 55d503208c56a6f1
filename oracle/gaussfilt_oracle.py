"""Oracle (test infrastructure): NumPy fp32 restatement of the reference's JAX filters.

Follows, line by line (paths relative to /root/reference):
  _predict                  gaussfiltax/inference.py:51-70
  _condition_on             gaussfiltax/inference.py:72-105
  _kalman_step (unused)     gaussfiltax/inference.py:107-120
  _MVN_log_prob             gaussfiltax/inference.py:24  (+ tfp MVNFullCovariance.log_prob)
  _get_params/_process_input/swap_axes_on_values   gaussfiltax/inference.py:21-26
  gaussian_sum_filter       gaussfiltax/inference.py:303-377
  bootstrap_particle_filter gaussfiltax/inference.py:1302-1380
  psd_solve, _resample, collapse   gaussfiltax/utils.py:256-259, 207-214, 10-18
  ParamsNLSSM / ParamsBPF   gaussfiltax/models.py:26-84
  NonlinearSSM.sample       gaussfiltax/models.py:240-289
  ParamsUKF, _ukf_predict_nonadditive, _ukf_condition_on_nonadditive, unscented_gaussian_sum_filter
                            gaussfiltax/inference.py:41-49, 146-174, 198-224, 379-456
  _get_sigma_points         gaussfiltax/utils.py:247-254
  speedy_augmented_gaussian_sum_filter   gaussfiltax/inference.py:621-812
  augmented_gaussian_sum_filter          gaussfiltax/inference.py:458-620 + containers.py:63-140
  speedy_unscented_agsf / unscented_agsf gaussfiltax/inference.py:966-1156 / 813-965
  augmented_gaussian_sum_filter_optimal  gaussfiltax/inference.py:1157-1300; optimal_resampling utils.py:216-244

Quirks reproduced on purpose: update->reweight->predict order; psd_solve adds
1e-6 to EVERY entry of S and uses LU (sgesv); posterior covariance P - K S K^T with the
un-jittered S and no symmetrisation; log-likelihood from a Cholesky of the un-jittered S;
linear-domain weights (0/0 -> NaN allowed); num_iter ignored; inputs=None -> zeros((T,1));
time-varying parameters only through an extra leading axis on the (d,d) covariances.

All arithmetic float32.  Pinned by the reference's recorded notebook outputs for the sampler, GSF, UGSF and BPF;
PARITY UNPINNED for the augmented filters (see oracle/__init__.py).
"""
from typing import NamedTuple, Optional, Callable
import numpy as np

from . import threefry as tf

F32 = np.float32
_LOG2PI = F32(np.log(2.0 * np.pi))


# --------------------------------------------------------------------------- containers
class ParamsNLSSM(NamedTuple):
    """gaussfiltax/models.py:26-51 (field names and order identical)."""
    initial_mean: np.ndarray
    initial_covariance: np.ndarray
    dynamics_function: Callable
    dynamics_noise_bias: np.ndarray
    dynamics_noise_covariance: np.ndarray
    emission_function: Callable
    emission_noise_bias: np.ndarray
    emission_noise_covariance: np.ndarray


class ParamsBPF(NamedTuple):
    """gaussfiltax/models.py:55-84."""
    initial_mean: np.ndarray
    initial_covariance: np.ndarray
    dynamics_function: Callable
    dynamics_noise_bias: np.ndarray
    dynamics_noise_covariance: np.ndarray
    emission_function: Callable
    emission_noise_bias: np.ndarray
    emission_noise_covariance: np.ndarray
    emission_distribution_log_prob: Callable


class PosteriorGaussianSumFiltered(NamedTuple):
    """gaussfiltax/inference.py:29-39."""
    weights: Optional[np.ndarray] = None
    means: Optional[np.ndarray] = None
    covariances: Optional[np.ndarray] = None
    predicted_means: Optional[np.ndarray] = None
    predicted_covariances: Optional[np.ndarray] = None


# --------------------------------------------------------------------------- helpers
def _get_params(x, dim, t):
    """inference.py:21."""
    x = np.asarray(x)
    return x[t] if x.ndim == dim + 1 else x


def _process_input(inputs, T):
    """inference.py:23."""
    return np.zeros((T, 1), dtype=F32) if inputs is None else np.asarray(inputs, dtype=F32)


def _mm(a, b):
    return np.matmul(a, b, dtype=F32)


def psd_solve(A, b):
    """utils.py:256-259: solve(A + 1e-6 (every entry), b) by LU with partial pivoting."""
    A = (np.asarray(A, dtype=F32) + F32(1e-6)).astype(F32)
    b = np.asarray(b, dtype=F32)
    if not (np.all(np.isfinite(A)) and np.all(np.isfinite(b))):
        return np.full(b.shape, np.nan, dtype=F32)      # jax propagates non-finite values; LAPACK may raise
    try:
        return np.linalg.solve(A, b).astype(F32)
    except np.linalg.LinAlgError:                        # exactly singular: jax returns inf / nan, never raises
        return np.full(b.shape, np.nan, dtype=F32)


def lu_solve_explicit(A, b):
    """The same solve written out (sgetrf/sgetrs order); cross-check of psd_solve's LAPACK call."""
    A = np.array(A, dtype=F32)
    b = np.array(b, dtype=F32)
    n = A.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            b[[k, p]] = b[[p, k]]
        for i in range(k + 1, n):
            l = F32(A[i, k] / A[k, k])
            A[i, k] = l
            A[i, k + 1:] = (A[i, k + 1:] - l * A[k, k + 1:]).astype(F32)
            b[i] = (b[i] - l * b[k]).astype(F32)
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            b[i] = (b[i] - A[i, j] * b[j]).astype(F32)
        b[i] = (b[i] / A[i, i]).astype(F32)
    return b


def mvn_log_prob(mean, cov, y):
    """inference.py:24: MVN(mean, cov).log_prob(atleast_1d(y)) via Cholesky (tfp MVNTriL)."""
    mean = np.atleast_1d(np.asarray(mean, dtype=F32))
    y = np.atleast_1d(np.asarray(y, dtype=F32))
    cov = np.asarray(cov, dtype=F32).reshape(mean.size, mean.size)
    try:
        L = np.linalg.cholesky(cov).astype(F32)
    except np.linalg.LinAlgError:
        return F32(np.nan)
    d = (y - mean).astype(F32)
    z = np.empty_like(d)
    for i in range(d.size):                      # forward substitution, fp32
        acc = d[i]
        for j in range(i):
            acc = F32(acc - L[i, j] * z[j])
        z[i] = F32(acc / L[i, i])
    quad = F32(0.0)
    logdet = F32(0.0)
    for i in range(d.size):
        quad = F32(quad + z[i] * z[i])
        logdet = F32(logdet + np.log(L[i, i]))
    return F32(F32(-0.5) * quad - F32(0.5) * F32(d.size) * _LOG2PI - logdet)


# --------------------------------------------------------------------------- step math
def _predict(m, P, fn, Q, q0, u):
    """inference.py:51-70.  fn supplies f and its analytic Jacobians (jacfwd in the reference)."""
    F_x = fn.jac_x(m, q0, u)
    F_q = fn.jac_noise(m, q0, u)
    mu_pred = fn.value(m, q0, u)
    Sigma_pred = (_mm(_mm(F_x, P), F_x.T) + _mm(_mm(F_q, Q), F_q.T)).astype(F32)
    return mu_pred, Sigma_pred, F_x


def _condition_on(m, P, hn, R, r0, u, y):
    """inference.py:72-105."""
    H_x = hn.jac_x(m, r0, u)
    H_r = hn.jac_noise(m, r0, u)
    S = (_mm(_mm(H_r, R), H_r.T) + _mm(_mm(H_x, P), H_x.T)).astype(F32)
    K = psd_solve(S, _mm(H_x, P)).T
    posterior_cov = (P - _mm(_mm(K, S), K.T)).astype(F32)
    hm = hn.value(m, r0, u)
    posterior_mean = (m + _mm(K, (y - hm).astype(F32))).astype(F32)
    ll = mvn_log_prob(hm, S, y)
    return ll, posterior_mean, posterior_cov, H_x, K


def _kalman_step(m, P, fn, Q, q0, u, hn, R, r0, y):
    """inference.py:107-120 (dead code in the reference: predict THEN update)."""
    mu_pred, Sigma_pred, _ = _predict(m, P, fn, Q, q0, u)
    ll, pm, pc, _, _ = _condition_on(mu_pred, Sigma_pred, hn, R, r0, u, y)
    return ll, pm, pc


def reweight(lls, weights):
    """inference.py:347-350 (and :1350-1353): linear-domain weight update."""
    lls = np.asarray(lls, dtype=F32)
    with np.errstate(invalid="ignore", divide="ignore"):
        lls = (lls - np.max(lls)).astype(F32)
        w = (np.exp(lls).astype(F32) * weights).astype(F32)
        return (w / sum_f32(w)).astype(F32)


def sum_f32(x):
    """jnp.sum in fp32.  Adjacent-pair tree over a zero-padded power-of-two array (the
    up-sweep of threefry.cumsum_assoc, so sum == cumsum[-1]): the canonical order shared with
    the HIP kernels (XLA's reduction order is unspecified)."""
    x = np.asarray(x, dtype=F32).ravel()
    n = 1
    while n < x.size:
        n *= 2
    buf = np.zeros(n, dtype=F32)
    buf[:x.size] = x
    while n > 1:
        buf = (buf[0::2] + buf[1::2]).astype(F32)
        n //= 2
    return F32(buf[0])


# --------------------------------------------------------------------------- GSF / (E)KF
def initial_component_means(params, num_components, key=None):
    """inference.py:367: MVN(m0, P0).sample(K, PRNGKey(0)) = m0 + chol(P0) @ z_k, with
    z = normal(PRNGKey(0), (K, n)) (row-major).  Best-effort restatement of tfp's sampler."""
    key = tf.PRNGKey(0) if key is None else key
    m0 = np.asarray(params.initial_mean, dtype=F32)
    n = m0.size
    L = np.linalg.cholesky(np.asarray(params.initial_covariance, dtype=F32)).astype(F32)
    z = tf.normal(key, num_components * n).reshape(num_components, n)
    return (m0[None, :] + _mm(z, L.T)).astype(F32)


def gaussian_sum_filter(params, emissions, num_components=1, num_iter=1, inputs=None,
                        initial_means=None, return_ll=False):
    """inference.py:303-377.  ``initial_means`` (K,n) overrides the PRNGKey(0) draw of :367."""
    emissions = np.asarray(emissions, dtype=F32)
    T = len(emissions)
    K = num_components
    fn, hn = params.dynamics_function, params.emission_function
    inputs = _process_input(inputs, T)
    n = np.asarray(params.initial_mean).size

    if initial_means is None:
        initial_means = initial_component_means(params, K)
    pred_means = np.array(initial_means, dtype=F32).reshape(K, n)
    pred_covs = np.stack([np.asarray(params.initial_covariance, dtype=F32)] * K)
    weights = (np.ones(K, dtype=F32) / F32(K)).astype(F32)

    out_w = np.empty((T, K), F32)
    out_m = np.empty((T, K, n), F32)
    out_P = np.empty((T, K, n, n), F32)
    out_pm = np.empty((T, K, n), F32)
    out_pP = np.empty((T, K, n, n), F32)
    out_ll = np.empty((T, K), F32)

    for t in range(T):
        Q = np.asarray(_get_params(params.dynamics_noise_covariance, 2, t), dtype=F32)
        q0 = np.asarray(_get_params(params.dynamics_noise_bias, 2, t), dtype=F32)
        R = np.asarray(_get_params(params.emission_noise_covariance, 2, t), dtype=F32)
        r0 = np.asarray(_get_params(params.emission_noise_bias, 2, t), dtype=F32)
        u = inputs[t]
        y = emissions[t]
        lls = np.empty(K, F32)
        fm = np.empty((K, n), F32)
        fP = np.empty((K, n, n), F32)
        for k in range(K):
            lls[k], fm[k], fP[k], _, _ = _condition_on(pred_means[k], pred_covs[k], hn, R, r0, u, y)
        out_ll[t] = lls
        weights = reweight(lls, weights)
        for k in range(K):
            pred_means[k], pred_covs[k], _ = _predict(fm[k], fP[k], fn, Q, q0, u)
        out_w[t], out_m[t], out_P[t], out_pm[t], out_pP[t] = weights, fm, fP, pred_means, pred_covs

    post = PosteriorGaussianSumFiltered(
        weights=out_w.swapaxes(0, 1).copy(),
        means=out_m.swapaxes(0, 1).copy(),
        covariances=out_P.swapaxes(0, 1).copy(),
        predicted_means=out_pm.swapaxes(0, 1).copy(),
        predicted_covariances=out_pP.swapaxes(0, 1).copy(),
    )
    if return_ll:
        return post, out_ll.swapaxes(0, 1).copy()
    return post


# --------------------------------------------------------------------------- unscented variants
class ParamsUKF(NamedTuple):
    """gaussfiltax/inference.py:41-49."""
    alpha: float = 1e-3
    beta: float = 2
    kappa: float = 0


def sym_sqrtm(P):
    """``jnp.real(scipy.linalg.sqrtm(P))`` of utils.py:250 for a symmetric matrix: the principal square
    root V diag(sqrt(lambda)) V^T; eigenvalues below zero (rounding) have a purely imaginary root whose
    real part is 0.  Evaluated in float64 and rounded to float32 (jax runs a float32 Schur iteration;
    the two agree to float32 resolution for the well-conditioned covariances of the parity tests)."""
    P = np.asarray(P, dtype=np.float64)
    if not np.all(np.isfinite(P)):      # NaN in, NaN out (jax propagates; LAPACK would raise)
        return np.full(P.shape, np.nan, dtype=F32)
    lam, V = np.linalg.eigh(P)
    return ((V * np.sqrt(np.maximum(lam, 0.0))) @ V.T).astype(F32)


def _get_sigma_points(m, P, ulambda):
    """utils.py:247-254: the 2 dx rows m +- sqrt(dx + lambda) * sqrtm(P)^T."""
    dx = m.shape[0]
    L = sym_sqrtm(P)
    c = np.sqrt(F32(dx) + F32(ulambda)).astype(F32)
    plus = (np.stack([m] * dx, axis=0) + c * L.T).astype(F32)
    minus = (np.stack([m] * dx, axis=0) - c * L.T).astype(F32)
    return np.concatenate([plus, minus], axis=0)


def _ulambda(uparams, L):
    return F32(F32(uparams.alpha) ** 2 * F32(L + uparams.kappa) - F32(L))


def _ukf_predict_nonadditive(m, P, f, u, Q, uparams, q0):
    """inference.py:146-174."""
    n, d = m.shape[0], Q.shape[0]
    lam = _ulambda(uparams, n + d)
    mA = np.concatenate((m, q0)).astype(F32)
    PA = np.zeros((n + d, n + d), F32)
    PA[:n, :n], PA[n:, n:] = P, Q
    sp = _get_sigma_points(mA, PA, lam)
    new = np.stack([f(x[:n], x[n:], u) for x in sp]).astype(F32)
    f0 = f(m, q0, u).astype(F32)
    den = F32(2) * (lam + F32(n + d))
    mu = (np.sum(new, axis=0, dtype=F32) / den + f0 * (lam / (lam + F32(n + d)))).astype(F32)
    dev = (new - mu).astype(F32)
    wc = F32(lam / (lam + F32(n + d)) + F32(1) - F32(uparams.alpha) ** 2 + F32(uparams.beta))
    Sigma = (_mm(dev.T, dev) / den + wc * np.outer(f0 - mu, f0 - mu).astype(F32)).astype(F32)
    return mu, Sigma


def _ukf_condition_on_nonadditive(m, P, h, R, u, y, uparams, r0):
    """inference.py:198-224."""
    n, d = m.shape[0], r0.shape[0]
    lam = _ulambda(uparams, n + d)
    mA = np.concatenate((m, r0)).astype(F32)
    PA = np.zeros((n + d, n + d), F32)
    PA[:n, :n], PA[n:, n:] = P, R
    sp = _get_sigma_points(mA, PA, lam)
    new = np.stack([h(x[:n], x[n:], u) for x in sp]).astype(F32)
    h0 = h(m, r0, u).astype(F32)
    den = F32(2) * (lam + F32(n + d))
    mu = (np.sum(new, axis=0, dtype=F32) / den + h0 * (lam / (lam + F32(n + d)))).astype(F32)
    dev = (new - mu).astype(F32)
    wc = F32(lam / (lam + F32(n + d)) + F32(1) - F32(uparams.alpha) ** 2 + F32(uparams.beta))
    S = (_mm(dev.T, dev) / den + wc * np.outer(h0 - mu, h0 - mu).astype(F32)).astype(F32)
    C = (_mm(dev.T, (sp[:, :n] - m).astype(F32)) / den).astype(F32)
    K = psd_solve(S, C).T
    posterior_cov = (P - _mm(_mm(K, S), K.T)).astype(F32)
    posterior_mean = (m + _mm(K, (y - mu).astype(F32))).astype(F32)
    ll = mvn_log_prob(mu, S, y)
    return ll, posterior_mean, posterior_cov


def unscented_gaussian_sum_filter(params, uparams, emissions, num_components=1, num_iter=1, inputs=None,
                                  initial_means=None, return_ll=False):
    """inference.py:379-456.  ``initial_means`` (K,n) overrides the PRNGKey(0) draw of :445."""
    emissions = np.asarray(emissions, dtype=F32)
    T = len(emissions)
    K = num_components
    fn, hn = params.dynamics_function, params.emission_function
    inputs = _process_input(inputs, T)
    n = np.asarray(params.initial_mean).size
    if initial_means is None:
        initial_means = initial_component_means(params, K)
    pred_means = np.array(initial_means, dtype=F32).reshape(K, n)
    pred_covs = np.stack([np.asarray(params.initial_covariance, dtype=F32)] * K)
    weights = (np.ones(K, dtype=F32) / F32(K)).astype(F32)
    out_w = np.empty((T, K), F32)
    out_m = np.empty((T, K, n), F32)
    out_P = np.empty((T, K, n, n), F32)
    out_pm = np.empty((T, K, n), F32)
    out_pP = np.empty((T, K, n, n), F32)
    out_ll = np.empty((T, K), F32)
    for t in range(T):
        Q = np.asarray(_get_params(params.dynamics_noise_covariance, 2, t), dtype=F32)
        q0 = np.asarray(_get_params(params.dynamics_noise_bias, 2, t), dtype=F32)
        R = np.asarray(_get_params(params.emission_noise_covariance, 2, t), dtype=F32)
        r0 = np.asarray(_get_params(params.emission_noise_bias, 2, t), dtype=F32)
        u, y = inputs[t], emissions[t]
        lls = np.empty(K, F32)
        fm = np.empty((K, n), F32)
        fP = np.empty((K, n, n), F32)
        for k in range(K):
            lls[k], fm[k], fP[k] = _ukf_condition_on_nonadditive(pred_means[k], pred_covs[k], hn, R, u, y, uparams, r0)
        out_ll[t] = lls
        weights = reweight(lls, weights)
        for k in range(K):
            pred_means[k], pred_covs[k] = _ukf_predict_nonadditive(fm[k], fP[k], fn, u, Q, uparams, q0)
        out_w[t], out_m[t], out_P[t], out_pm[t], out_pP[t] = weights, fm, fP, pred_means, pred_covs
    post = PosteriorGaussianSumFiltered(
        weights=out_w.swapaxes(0, 1).copy(), means=out_m.swapaxes(0, 1).copy(), covariances=out_P.swapaxes(0, 1).copy(),
        predicted_means=out_pm.swapaxes(0, 1).copy(), predicted_covariances=out_pP.swapaxes(0, 1).copy())
    if return_ll:
        return post, out_ll.swapaxes(0, 1).copy()
    return post


def optimal_resampling(weights, N, key):
    """utils.py:216-244 (Fearnhead & Clifford 2003 as written there).  Conventions where XLA's are unspecified:
    stable sort / argsort; the running sums ``lower_diag[M-N:M-1] @ sorted_weights`` accumulate in index order;
    ``.sum()`` in the adjacent-pair tree order (sum_f32); jr.choice's cumsum in associative_scan order."""
    weights = np.asarray(weights, dtype=F32)
    M = weights.shape[0]
    sorted_idx = np.argsort(weights, kind="stable").astype(np.int32)
    sw = weights[sorted_idx]
    cum = np.empty(M, F32)
    acc = F32(0.0)
    for c in range(M):
        acc = F32(acc + sw[c])
        cum[c] = acc
    L = 0
    ps = np.zeros(max(N - 1, 0), F32)
    for ind in range(1, N):
        ps[ind - 1] = F32(cum[M - 1 - ind] / F32(N - ind))
        if sw[M - ind - 1] < ps[ind - 1] < sw[M - ind]:
            L += ind
    p = F32(1.0) / F32(N) if L == 0 else ps[min(max(L, 1), N - 1) - 1]
    below = sw < p
    res_w = np.where(below, sw, F32(0.0)).astype(F32)
    with np.errstate(invalid="ignore", divide="ignore"):
        res_w = (res_w / sum_f32(res_w)).astype(F32)
    cdf = tf.cumsum_assoc(res_w)
    u = tf.uniform(key, M)
    r = (cdf[-1] * (F32(1.0) - u)).astype(F32)
    res_idx = np.array([int(np.searchsorted(cdf, rv, side="left")) if rv == rv else M - 1 for rv in r])
    res_idx = np.minimum(res_idx, M - 1)
    unsort = sorted_idx[res_idx]
    final_idx = np.where(below, unsort, sorted_idx).astype(np.int32)
    final_w = np.where(below, p, sw).astype(F32)
    top = final_w[M - N:]
    return final_idx[M - N:], (top / sum_f32(top)).astype(F32)


def chol_jax(A):
    """jnp.linalg.cholesky as the reference's CPU runs evaluate it: the input is symmetrised ((A + A^T) / 2), LAPACK potrf
    factors it, and a failed factorisation (a pivot <= 0 or NaN) returns an all-NaN matrix instead of raising."""
    A = np.asarray(A, dtype=F32)
    A = (F32(0.5) * (A + A.T)).astype(F32)
    if not np.all(np.isfinite(A)):
        return np.full(A.shape, np.nan, dtype=F32)
    try:
        return np.linalg.cholesky(A).astype(F32)
    except np.linalg.LinAlgError:
        return np.full(A.shape, np.nan, dtype=F32)


# --------------------------------------------------------------------------- augmented GSF (speedy variant)
def speedy_augmented_gaussian_sum_filter(params, emissions, num_components, rng_key=None, num_iter=1,
                                         opt_args=(0.1, 0.1), inputs=None, initial_means=None, debug=False, variant=0,
                                         uparams=None):
    """inference.py:621-812.  Every step branches each of the N0 carried components into N1 z-samples
    (drawn from N(m, P - Delta), Delta = opt_args[0] P, :675-688), predicts each with covariance Delta
    (:695-698), branches every prediction into N2 s-samples (N(m-, P- - Lambda), Lambda = opt_args[1] P-,
    :711-726), updates each with covariance Lambda (:735-744) and resamples N0 of the N0 N1 N2 leaves with
    jr.choice under the FIXED key PRNGKey(0) (:760).  Quirks kept: ``rng_key`` is never advanced, so the
    same two normal arrays are used at every step (:672, :716); the s-key is split(key)[0] of the z-key;
    the leaf weights are the carried weights / N1 / N2 times exp(ll - max) (:699, :738-743); the emitted
    weights are 1 / N0 (:765).  Returns (PosteriorGaussianSumFiltered(weights, means, covariances), aux)
    with aux = {'pre_weights', 'updated_means'} (two of the reference's aux outputs) when ``debug``."""
    emissions = np.asarray(emissions, dtype=F32)
    T = len(emissions)
    N0, N1, N2 = (int(v) for v in num_components)
    fn, hn = params.dynamics_function, params.emission_function
    inputs = _process_input(inputs, T)
    n = np.asarray(params.initial_mean).size
    rng_key = tf.PRNGKey(0) if rng_key is None else np.asarray(rng_key, dtype=np.uint32)
    if initial_means is None:
        initial_means = initial_component_means(params, N0)
    fmeans = np.array(initial_means, dtype=F32).reshape(N0, n)
    fcovs = np.stack([np.asarray(params.initial_covariance, dtype=F32)] * N0)
    weights = (np.ones(N0, dtype=F32) / F32(N0)).astype(F32)
    a0, a1 = F32(opt_args[0]), F32(opt_args[1])
    key = tf.split(rng_key, 2)[0]                       # :672  key, subkey = jr.split(rng_key)
    if variant == 0:
        eps_z = tf.normal(key, N0 * n * N1).reshape(N0, n, N1)
        key2 = tf.split(key, 2)[0]                      # :716  key, _ = jr.split(key)
        eps_s = tf.normal(key2, N0 * N1 * n * N2).reshape(N0 * N1, n, N2)
    else:
        # augmented_gaussian_sum_filter (:458-620) via containers._branches_from_tree1/2 (containers.py:63-140):
        # keys = split(subkey, #nodes); node j: jr.multivariate_normal(keys[j], mean, cov - split_cov, (num,))
        #   = mean + chol @ normal(keys[j], (num, n))[i]      (method='cholesky'); NaN samples -> mean
        sub1 = tf.split(rng_key, 2)[1]                  # :519
        sub2 = tf.split(key, 2)[1]                      # :545  key, subkey = jr.split(key)
        k1, k2 = tf.split(sub1, N0), tf.split(sub2, N0 * N1)
        eps_z = np.stack([tf.normal(k1[i], N1 * n).reshape(N1, n).T for i in range(N0)])          # (N0, n, N1)
        eps_s = np.stack([tf.normal(k2[j], N2 * n).reshape(N2, n).T for j in range(N0 * N1)])     # (N0 N1, n, N2)
    out_w = np.empty((T, N0), F32)
    out_m = np.empty((T, N0, n), F32)
    out_P = np.empty((T, N0, n, n), F32)
    aux_pre, aux_um, aux_idx = [], [], []
    M = N0 * N1 * N2
    for t in range(T):
        Q = np.asarray(_get_params(params.dynamics_noise_covariance, 2, t), dtype=F32)
        q0 = np.asarray(_get_params(params.dynamics_noise_bias, 2, t), dtype=F32)
        R = np.asarray(_get_params(params.emission_noise_covariance, 2, t), dtype=F32)
        r0 = np.asarray(_get_params(params.emission_noise_bias, 2, t), dtype=F32)
        u, y = inputs[t], emissions[t]
        pm = np.empty((N0 * N1, n), F32)
        pP = np.empty((N0 * N1, n, n), F32)
        for i0 in range(N0):
            Delta = (a0 * fcovs[i0]).astype(F32)
            Lz = chol_jax((fcovs[i0] - Delta).astype(F32))
            zc = _mm(Lz, eps_z[i0])                      # (n, N1)
            for i1 in range(N1):
                z = (fmeans[i0] + zc[:, i1]).astype(F32)
                if variant:
                    z = np.where(np.isnan(z), fmeans[i0], z).astype(F32)                     # containers.py:84
                if uparams is None:
                    pm[i0 * N1 + i1], pP[i0 * N1 + i1], _ = _predict(z, Delta, fn, Q, q0, u)
                else:    # speedy_unscented_agsf :1034-1035
                    pm[i0 * N1 + i1], pP[i0 * N1 + i1] = _ukf_predict_nonadditive(z, Delta, fn, u, Q, uparams, q0)
        lls = np.empty(M, F32)
        um = np.empty((M, n), F32)
        uP = np.empty((M, n, n), F32)
        for j in range(N0 * N1):
            Lam = (a1 * pP[j]).astype(F32)
            Ls = chol_jax((pP[j] - Lam).astype(F32))
            sc = _mm(Ls, eps_s[j])                       # (n, N2)
            for i2 in range(N2):
                sv = (pm[j] + sc[:, i2]).astype(F32)
                if variant:
                    sv = np.where(np.isnan(sv), pm[j], sv).astype(F32)                        # containers.py:121
                if uparams is None:
                    lls[j * N2 + i2], um[j * N2 + i2], uP[j * N2 + i2], _, _ = _condition_on(sv, Lam, hn, R, r0, u, y)
                else:    # :1075-1076
                    lls[j * N2 + i2], um[j * N2 + i2], uP[j * N2 + i2] = _ukf_condition_on_nonadditive(sv, Lam, hn, R, u, y, uparams, r0)
        pw = (np.repeat(weights, N1) / F32(N1)).astype(F32)          # :699
        uw = (np.repeat(pw, N2) / F32(N2)).astype(F32)               # :738
        w = reweight(lls, uw)                                         # :740-743
        if variant == 2:     # augmented_gaussian_sum_filter_optimal :1256
            idx, weights = optimal_resampling(w, N0, tf.split(key, 2)[0])
        else:
            idx = tf.choice_indices(tf.cumsum_assoc(w), tf.uniform(tf.PRNGKey(0), N0))   # :760
            idx = np.minimum(idx, M - 1)
            weights = (np.ones(N0, dtype=F32) / F32(N0)).astype(F32)
        fmeans, fcovs = um[idx].copy(), uP[idx].copy()
        out_w[t], out_m[t], out_P[t] = weights, fmeans, fcovs
        if debug:
            aux_pre.append(w.copy())
            aux_idx.append(np.asarray(idx, dtype=np.int32).copy())
            aux_um.append(um.copy())
    post = PosteriorGaussianSumFiltered(weights=out_w.swapaxes(0, 1).copy(), means=out_m.swapaxes(0, 1).copy(),
                                        covariances=out_P.swapaxes(0, 1).copy())
    aux = {"pre_weights": np.stack(aux_pre), "updated_means": np.stack(aux_um), "leaf_indices": np.stack(aux_idx)} if debug else {}
    return post, aux


def augmented_gaussian_sum_filter(params, emissions, num_components, rng_key=None, num_iter=1, opt_args=(0.1, 0.1),
                                  inputs=None, initial_means=None, debug=False):
    """inference.py:458-620: the speedy filter's tree with container-based branches (see ``variant`` above)."""
    return speedy_augmented_gaussian_sum_filter(params, emissions, num_components, rng_key, num_iter, opt_args, inputs,
                                                initial_means, debug, variant=1)


def augmented_gaussian_sum_filter_optimal(params, emissions, num_components, rng_key=None, num_iter=1, opt_args=(0.1, 0.1),
                                          inputs=None, initial_means=None, debug=False):
    """inference.py:1157-1300: container-based branches + optimal_resampling; unequal carried weights."""
    return speedy_augmented_gaussian_sum_filter(params, emissions, num_components, rng_key, num_iter, opt_args, inputs,
                                                initial_means, debug, variant=2)


def speedy_unscented_agsf(params, uparams, emissions, num_components, rng_key=None, num_iter=1, opt_args=(0.1, 0.1),
                          inputs=None, initial_means=None, debug=False):
    """inference.py:966-1156: the speedy augmented filter with unscented nodes."""
    return speedy_augmented_gaussian_sum_filter(params, emissions, num_components, rng_key, num_iter, opt_args, inputs,
                                                initial_means, debug, variant=0, uparams=uparams)


def unscented_agsf(params, uparams, emissions, num_components, rng_key=None, num_iter=1, opt_args=(0.1, 0.1),
                   inputs=None, initial_means=None, debug=False):
    """inference.py:813-965: unscented nodes, container-based branches."""
    return speedy_augmented_gaussian_sum_filter(params, emissions, num_components, rng_key, num_iter, opt_args, inputs,
                                                initial_means, debug, variant=1, uparams=uparams)


def collapse(mean_mat, covariance_tens, weight_vec):
    """utils.py:10-18 (moment-matched single Gaussian; the reference runs this in NumPy fp64
    on whatever dtype it is handed; here the inputs' dtype is kept)."""
    M, dx = np.shape(mean_mat)
    mean_out = np.matmul(weight_vec, mean_mat)
    cov_out = np.zeros([dx, dx], dtype=mean_out.dtype)
    for m in range(M):
        diff = mean_mat[m] - mean_out
        cov_out = cov_out + weight_vec[m] * (covariance_tens[m] + np.tensordot(diff, diff, axes=0))
    return mean_out, cov_out


def point_estimate(post):
    """sum_k w_k m_k per t (docs/experiments/BOT_Experiment_script.py:101)."""
    return np.sum(post.means * post.weights[..., None], axis=0).astype(F32)


# --------------------------------------------------------------------------- BPF
class GaussianEmissionLogProb:
    """MVN(loc=h(x, r_eval, u), covariance_matrix=R).log_prob(y): the form of every
    ``*lp`` function the reference's scripts define (e.g. nonlinearities.py:51-52 g96lp,
    BOT_Experiment_script.py:45 gBOTlp)."""

    def __init__(self, hn, R, r_eval=None):
        self.hn = hn
        self.R = np.asarray(R, dtype=F32)
        self.r_eval = np.zeros(hn.noise_dim, dtype=F32) if r_eval is None else np.asarray(r_eval, dtype=F32)

    def __call__(self, x, y, u):
        return mvn_log_prob(self.hn.value(np.asarray(x, dtype=F32), self.r_eval, u), self.R, y)

    def logprob_c(self, X, y, u):
        """Canonical fp32 arithmetic (oracle/fp32.py), batched over particles X (N, n): the sequence the HIP engine runs
        (ssm_device.hpp: emission_loglik) -- Cholesky factor by plain loops, forward substitution by fma, multiplication by
        the reciprocal diagonal, quadratic form by fma, -0.5 quad + const as one fma."""
        return _gaussian_logprob_c(self.hn.value_c(np.asarray(X, dtype=F32), np.tile(self.r_eval, (len(X), 1)), u), self.R, y, None)


def _gaussian_logprob_c(HX, R, y, scale):
    from . import fp32
    R = np.asarray(R, dtype=F32)
    m = R.shape[0]
    L = fp32.cholesky_lower(R)
    rd = (F32(1.0) / np.diag(L)).astype(F32)
    logdet = F32(0.0)
    for i in range(m):
        logdet = F32(logdet + fp32.canon_log(L[i, i])[0])
    lp_const = F32(F32(F32(-0.5) * F32(m)) * F32(1.8378770664093453) - logdet)
    y = np.asarray(y, dtype=F32).reshape(m)
    N = HX.shape[0]
    zz = np.zeros((N, m), dtype=F32)
    quad = np.zeros(N, dtype=F32)
    lsc = np.zeros(N, dtype=F32)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        for a in range(m):
            sres = (y[a] - HX[:, a]).astype(F32)
            if scale is not None:
                sres = (sres / scale[:, a]).astype(F32)
                lsc = (lsc + fp32.canon_log(scale[:, a])).astype(F32)
            for c in range(a):
                sres = fp32.fma(-L[a, c], zz[:, c], sres)
            zz[:, a] = (sres * rd[a]).astype(F32)
            quad = fp32.fma(zz[:, a], zz[:, a], quad)
        return (fp32.fma(F32(-0.5), quad, lp_const) - lsc).astype(F32)


class StochVolEmissionLogProb:
    """lmsvlp (docs/experiments/adaptive_experiment.py:55-57): MVN(loc=glmsv(x, r0, u), covariance_matrix=M R M^T)
    .log_prob(y) with M = u beta diag(exp(x / sigma)) + (1 - u) I, the noise Jacobian of the stochastic-volatility
    emission (models.StochVol.jac_noise)."""

    def __init__(self, hn, R, r_eval=None):
        self.hn = hn
        self.R = np.asarray(R, dtype=F32)
        self.r_eval = np.zeros(hn.noise_dim, dtype=F32) if r_eval is None else np.asarray(r_eval, dtype=F32)

    def __call__(self, x, y, u):
        x = np.asarray(x, dtype=F32)
        Mx = self.hn.jac_noise(x, self.r_eval, u)
        cov = _mm(_mm(Mx, self.R), Mx.T)
        return mvn_log_prob(self.hn.value(x, self.r_eval, u), cov, y)

    def logprob_c(self, X, y, u):
        """Canonical arithmetic: chol(M R M^T) = M chol(R) for the positive diagonal M -- the residual is divided by
        diag(M) before the constant factor's forward substitution and sum log M_ii joins the log-determinant."""
        X = np.asarray(X, dtype=F32)
        return _gaussian_logprob_c(self.hn.value_c(X, np.tile(self.r_eval, (len(X), 1)), u), self.R, y, self.hn.scale_c(X, u))


def sample_dynamics_distribution(params, key, x, u, cholQ=None):
    """models.py:82-84: q = MVN(q0, Q).sample(seed=key); return f(x, q, u)."""
    if cholQ is None:
        cholQ = np.linalg.cholesky(np.asarray(params.dynamics_noise_covariance, dtype=F32)).astype(F32)
    q = tf.mvn_sample(key, np.asarray(params.dynamics_noise_bias, dtype=F32), cholQ)
    return params.dynamics_function.value(np.asarray(x, dtype=F32), q, u)


def _resample(weights, particles, key):
    """utils.py:207-214 (multinomial inverse-CDF; returns also the ancestor indices)."""
    keys = tf.split(key, 2)
    N = weights.shape[0]
    idx = tf.choice(keys[0], weights)
    idx = np.minimum(idx, N - 1)
    return (np.ones(N, dtype=F32) / F32(N)).astype(F32), particles[idx], keys[1], idx


def systematic_indices(weights, u0):
    """Systematic resampling (north_star's alternative; NOT the reference's semantics):
    positions (i + u0)/N against the canonical cumsum."""
    N = weights.shape[0]
    cdf = tf.cumsum_assoc(weights)
    pos = ((np.arange(N, dtype=F32) + F32(u0)) / F32(N)).astype(F32) * cdf[-1]
    return np.minimum(np.searchsorted(cdf, pos.astype(F32), side="left"), N - 1).astype(np.int32)


def _bpf_canonical(params, emissions, num_particles, key, inputs, ess_threshold, resampler, debug):
    """inference.py:1302-1380 on the canonical fp32 arithmetic of oracle/fp32.py, vectorised over the particles: the
    operation sequence of the HIP engine's particle kernels, so that weights -- and with them every ancestor index of
    every step -- agree bit for bit.  Same structure as the loop below; only the rounding of each quantity is pinned."""
    from . import fp32
    emissions = np.asarray(emissions, dtype=F32)
    T, N = len(emissions), num_particles
    inputs = _process_input(inputs, T)
    fn, lp = params.dynamics_function, params.emission_distribution_log_prob
    if not hasattr(fn, "value_c") or not hasattr(lp, "logprob_c") or not hasattr(lp.hn, "value_c"):
        raise NotImplementedError("canonical arithmetic is defined for models built from IEEE operations, exp and log")
    m0 = np.asarray(params.initial_mean, dtype=F32)
    n = m0.size
    q0 = np.asarray(params.dynamics_noise_bias, dtype=F32)
    L0 = fp32.cholesky_lower(params.initial_covariance)
    LQ = fp32.cholesky_lower(_get_params(params.dynamics_noise_covariance, 2, 0))
    dq = LQ.shape[0]
    keys = tf.split(key, N + 1)                                  # :1369
    next_key = keys[0]
    weights = (np.ones(N, dtype=F32) / F32(N)).astype(F32)
    z = fp32.bits_to_normal(tf.random_bits_keys(keys[1:], n))    # :1372-1373: MVN(m0, P0).sample(seed = keys[1 + i])
    particles = (m0 + fp32.lower_matvec_fma(L0, z)).astype(F32)
    out_w = np.empty((T, N), F32)
    out_x = np.empty((T, N, n), F32)
    dbg = {"resampled": np.zeros(T, bool), "ancestors": np.tile(np.arange(N, dtype=np.int32), (T, 1)),
           "ess": np.zeros(T, F32), "pre_weights": np.empty((T, N), F32)}
    for t in range(T):
        u, y = inputs[t], emissions[t]
        keys = tf.split(next_key, N + 1)                          # :1342
        next_key = keys[0]
        z = fp32.bits_to_normal(tf.random_bits_keys(keys[1:], dq))
        q = (q0 + fp32.lower_matvec_fma(LQ, z)).astype(F32)       # models.py:82-83
        new_particles = fn.value_c(particles, q, u)               # :1344-1345
        lls = lp.logprob_c(new_particles, y, u)                   # :1348-1349
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            e = (fp32.canon_exp((lls - np.max(lls)).astype(F32)) * weights).astype(F32)   # :1350-1352
            new_weights = (e / sum_f32(e)).astype(F32)            # :1353
            ess = F32(1.0) / sum_f32((new_weights * new_weights).astype(F32))
        dbg["ess"][t] = ess
        dbg["pre_weights"][t] = new_weights
        if ess < F32(ess_threshold) * F32(N):                    # :1356
            if resampler == "multinomial":
                weights, new_particles, next_key, idx = _resample(new_weights, new_particles, next_key)
            else:
                ks = tf.split(next_key, 2)
                idx = systematic_indices(new_weights, tf.uniform(ks[0], 1)[0])
                new_particles = new_particles[idx]
                weights = (np.ones(N, dtype=F32) / F32(N)).astype(F32)
                next_key = ks[1]
            dbg["resampled"][t] = True
            dbg["ancestors"][t] = idx
        else:
            weights = new_weights
        particles = new_particles
        out_w[t] = weights
        out_x[t] = particles
    out = {"weights": out_w.swapaxes(0, 1).copy(), "particles": out_x.swapaxes(0, 1).copy()}
    return (out, dbg) if debug else out


def bootstrap_particle_filter(params, emissions, num_particles, key=None, inputs=None,
                              ess_threshold=0.5, resampler="multinomial", debug=False, arith="libm"):
    """inference.py:1302-1380.  ``arith``: "libm" = NumPy's float32 functions and BLAS products (the reference leaves the
    rounding of exp / log1p / matrix products to XLA: any faithful choice is as good); "canonical" = the ONE definition of
    oracle/fp32.py that the HIP engine also implements -- bit-exact ancestry (:func:`_bpf_canonical`)."""
    key = tf.PRNGKey(0) if key is None else np.asarray(key, dtype=np.uint32)
    if arith == "canonical":
        return _bpf_canonical(params, emissions, num_particles, key, inputs, ess_threshold, resampler, debug)
    emissions = np.asarray(emissions, dtype=F32)
    T = len(emissions)
    N = num_particles
    inputs = _process_input(inputs, T)
    m0 = np.asarray(params.initial_mean, dtype=F32)
    n = m0.size
    L0 = np.linalg.cholesky(np.asarray(params.initial_covariance, dtype=F32)).astype(F32)

    keys = tf.split(key, N + 1)                                  # :1369
    next_key = keys[0]
    weights = (np.ones(N, dtype=F32) / F32(N)).astype(F32)
    particles = np.stack([tf.mvn_sample(keys[1 + i], m0, L0) for i in range(N)])   # :1372-1373

    out_w = np.empty((T, N), F32)
    out_x = np.empty((T, N, n), F32)
    dbg = {"resampled": np.zeros(T, bool), "ancestors": np.tile(np.arange(N, dtype=np.int32), (T, 1)),
           "ess": np.zeros(T, F32), "pre_weights": np.empty((T, N), F32)}

    for t in range(T):
        Qt = np.asarray(_get_params(params.dynamics_noise_covariance, 2, t), dtype=F32)
        cholQ = np.linalg.cholesky(Qt).astype(F32)
        u = inputs[t]
        y = emissions[t]
        keys = tf.split(next_key, N + 1)                          # :1342
        next_key = keys[0]
        new_particles = np.stack([sample_dynamics_distribution(params, keys[1 + i], particles[i], u, cholQ)
                                  for i in range(N)])            # :1344-1345
        lls = np.array([params.emission_distribution_log_prob(new_particles[i], y, u) for i in range(N)],
                       dtype=F32)                                 # :1348-1349
        new_weights = reweight(lls, weights)                     # :1350-1353
        with np.errstate(invalid="ignore", divide="ignore"):
            ess = F32(1.0) / sum_f32((new_weights * new_weights).astype(F32))
        dbg["ess"][t] = ess
        dbg["pre_weights"][t] = new_weights
        if ess < F32(ess_threshold) * F32(N):                    # :1356
            if resampler == "multinomial":
                weights, new_particles, next_key, idx = _resample(new_weights, new_particles, next_key)
            else:
                ks = tf.split(next_key, 2)
                idx = systematic_indices(new_weights, tf.uniform(ks[0], 1)[0])
                new_particles = new_particles[idx]
                weights = (np.ones(N, dtype=F32) / F32(N)).astype(F32)
                next_key = ks[1]
            dbg["resampled"][t] = True
            dbg["ancestors"][t] = idx
        else:
            weights = new_weights
        particles = new_particles
        out_w[t] = weights
        out_x[t] = particles

    out = {"weights": out_w.swapaxes(0, 1).copy(), "particles": out_x.swapaxes(0, 1).copy()}
    return (out, dbg) if debug else out


# --------------------------------------------------------------------------- data generator
def sample_ssm(params, key, num_timesteps, inputs=None):
    """NonlinearSSM.sample, models.py:240-289 (tfp sampler restated as loc + chol @ normal)."""
    inputs = _process_input(inputs, num_timesteps)
    fn, hn = params.dynamics_function, params.emission_function
    q0 = np.asarray(params.dynamics_noise_bias, dtype=F32)
    r0 = np.asarray(params.emission_noise_bias, dtype=F32)
    LQ = np.linalg.cholesky(np.asarray(params.dynamics_noise_covariance, dtype=F32)).astype(F32)
    LR = np.linalg.cholesky(np.asarray(params.emission_noise_covariance, dtype=F32)).astype(F32)
    L0 = np.linalg.cholesky(np.asarray(params.initial_covariance, dtype=F32)).astype(F32)
    k = tf.split(np.asarray(key, dtype=np.uint32), 3)            # key1, key2, key  (:273)
    state = tf.mvn_sample(k[0], np.asarray(params.initial_mean, dtype=F32), L0)
    r = tf.mvn_sample(k[1], r0, LR)
    states = [state]
    emis = [hn.value(state, r, inputs[0])]
    next_keys = tf.split(k[2], num_timesteps - 1)                # :280
    for t in range(1, num_timesteps):
        k1, k2 = tf.split(next_keys[t - 1], 2)                   # :262
        q = tf.mvn_sample(k1, q0, LQ)
        r = tf.mvn_sample(k2, r0, LR)
        state = fn.value(state, q, inputs[t])
        states.append(state)
        emis.append(hn.value(state, r, inputs[t]))
    return np.stack(states).astype(F32), np.stack(emis).astype(F32)


# --------------------------------------------------------------------------- fp64 textbook KF
def textbook_kalman_f64(A, GQGt, H, R, m0, P0, ys):
    """Independent fp64 textbook Kalman filter (no jitter, Joseph-free), same step order as the
    reference (update at t with the carried prior, then predict).  Used to PIN the oracle."""
    A, GQGt, H, R = (np.asarray(v, dtype=np.float64) for v in (A, GQGt, H, R))
    m = np.asarray(m0, dtype=np.float64).copy()
    P = np.asarray(P0, dtype=np.float64).copy()
    T = len(ys)
    n = m.size
    fm, fP, pm, pP, ll = (np.empty((T, n)), np.empty((T, n, n)), np.empty((T, n)), np.empty((T, n, n)), np.empty(T))
    for t in range(T):
        S = H @ P @ H.T + R
        K = np.linalg.solve(S, H @ P).T
        v = ys[t] - H @ m
        ll[t] = -0.5 * (v @ np.linalg.solve(S, v) + np.log(np.linalg.det(S)) + len(v) * np.log(2 * np.pi))
        m = m + K @ v
        P = P - K @ S @ K.T
        fm[t], fP[t] = m, P
        m = A @ m
        P = A @ P @ A.T + GQGt
        pm[t], pP[t] = m, P
    return fm, fP, pm, pP, ll
