"""Oracle (test infrastructure): restatement of the reference's LEGACY NumPy classes.

  GaussFilt.run + EKF.moment_approx (order=1)   gaussfiltax/gaussfilt.py:88-130, 217-252
  GaussSumFilt.run                              gaussfiltax/gausssumfilt.py:30-78
  gaussian_logpdf                               gaussfiltax/utils.py:75-79

Semantics that differ from the JAX path and are reproduced here: predict -> update order with the
initial state stored at index -1 (gaussfilt.py:103-110); gain = Cxy @ inv(Sy), no jitter (:118);
noise biases c, d are NOT used by the filters (moment_approx uses func(m) only, :240); the
Gaussian-sum predict adds the covariance itself instead of Q (gausssumfilt.py:59); point_est[t] is
the sum over ALL entries of means*weights broadcast to every coordinate (:76).  The reference runs
these in fp64; this restatement is fp32 like the engine it checks (jacobians from oracle/models.py
instead of jax.jacfwd).  The random draws of the legacy classes come from NumPy's global generator
and cannot be reproduced; they are inputs here.  PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np

from .gaussfilt_oracle import mvn_log_prob

F32 = np.float32
_Z1 = np.zeros(1, dtype=F32)


def _mm(a, b):
    return np.matmul(a, b, dtype=F32)


def ekf_run(fn, hn, Q, R, ys, m0, P0):
    """gaussfilt.py:88-130 with EKF.moment_approx(order=1).  fn/hn: oracle model objects (value, jac_x)
    with additive identity noise.  Returns (ll (T,), filtered_means (T,n), filtered_covs (T,n,n))."""
    ys = np.asarray(ys, dtype=F32)
    T, n = len(ys), np.asarray(m0).size
    m = np.asarray(m0, dtype=F32).copy()
    P = np.asarray(P0, dtype=F32).copy()
    zq = np.zeros(fn.noise_dim, F32)
    zr = np.zeros(hn.noise_dim, F32)
    ll, fm, fP = np.empty(T, F32), np.empty((T, n), F32), np.empty((T, n, n), F32)
    for t in range(T):
        J = fn.jac_x(m, zq, _Z1)
        m, P = fn.value(m, zq, _Z1), (np.asarray(Q, F32) + _mm(_mm(J, P), J.T)).astype(F32)          # :231-240 'pred'
        Jg = hn.jac_x(m, zr, _Z1)
        mu_y = hn.value(m, zr, _Z1)
        Sy = (np.asarray(R, F32) + _mm(_mm(Jg, P), Jg.T)).astype(F32)
        Cxy = _mm(P, Jg.T)
        K = _mm(Cxy, np.linalg.inv(Sy).astype(F32))                                                  # :118
        m = (m + _mm(K, (ys[t] - mu_y).astype(F32))).astype(F32)
        P = (P - _mm(_mm(K, Sy), K.T)).astype(F32)
        ll[t] = mvn_log_prob(mu_y, Sy, ys[t])                                                        # :121, utils.py:75-79
        fm[t], fP[t] = m, P
    return ll, fm, fP


def gsf_run(fn, hn, R, ys, init_means, P0):
    """gausssumfilt.py:30-78.  init_means (M, n) = m0 + N(0, I) draws (input).  Returns means (T+1,n,M),
    covs (T+1,n,n,M), weights (T+1,M), point_est (T,n) with the initial state at index T."""
    ys = np.asarray(ys, dtype=F32)
    T, (M, n) = len(ys), np.asarray(init_means).shape
    zq = np.zeros(fn.noise_dim, F32)
    zr = np.zeros(hn.noise_dim, F32)
    means, covs = np.zeros((T + 1, n, M), F32), np.zeros((T + 1, n, n, M), F32)
    weights, pe = np.zeros((T + 1, M), F32), np.zeros((T, n), F32)
    weights[T] = F32(1.0) / F32(M)
    means[T] = np.asarray(init_means, F32).T
    covs[T] = np.asarray(P0, F32)[:, :, None]
    for t in range(T):
        ll = np.empty(M, F32)
        for k in range(M):
            m, P = means[t - 1, :, k], covs[t - 1, :, :, k]
            J = fn.jac_x(m, zq, _Z1)
            mp, Pp = fn.value(m, zq, _Z1), (P + _mm(_mm(J, P), J.T)).astype(F32)                     # :58-59 (Q never added)
            Jg = hn.jac_x(mp, zr, _Z1)
            mu_y = hn.value(mp, zr, _Z1)
            Sy = (np.asarray(R, F32) + _mm(_mm(Jg, Pp), Jg.T)).astype(F32)
            K = _mm(_mm(Pp, Jg.T), np.linalg.inv(Sy).astype(F32))
            means[t, :, k] = (mp + _mm(K, (ys[t] - mu_y).astype(F32))).astype(F32)
            covs[t, :, :, k] = (Pp - _mm(_mm(K, Sy), K.T)).astype(F32)
            ll[k] = mvn_log_prob(mu_y, Sy, ys[t])
        ll = (ll - ll.max()).astype(F32)
        w = (np.exp(ll).astype(F32) * weights[t - 1]).astype(F32)
        weights[t] = w / w.sum(dtype=F32)
        pe[t] = np.sum(means[t] * weights[t])                                                        # :76 (scalar, broadcast)
    return means, covs, weights, pe
