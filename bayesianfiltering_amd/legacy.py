"""The reference's legacy NumPy classes as thin wrappers over the HIP engine.

Mirrors ``gaussfiltax/gaussfilt.py`` (``SSM`` :10-52, ``GaussFilt.run`` :88-130 with
``EKF.moment_approx`` :217-252), ``gaussfiltax/gausssumfilt.py`` (``GaussSumFilt.run`` :30-78) and
``gaussfiltax/particlefilt.py`` (``BootstrapPF.run`` :26-52): same class names, constructor
arguments, ``run(ys, m0, P0)`` signatures and return layouts (initial state stored at index -1 of
the time axis).  The legacy semantics that differ from the JAX path are selected with
``bf_model.flags`` inside the Gaussian-sum kernel: predict -> update order, gain from ``S`` without
the 1e-6 jitter, ``P + J P J^T`` (no Q) in ``GaussSumFilt``'s predict, and the scalar ``point_est``
quirk of gausssumfilt.py:76.  Differences: arithmetic is fp32 (the legacy classes run NumPy fp64);
``f`` / ``g`` are registry functions of ``nonlinearities`` called as ``f(x)``; random draws come from
the engine's Threefry streams (``key=``) instead of NumPy's global generator; ``UKF`` / ``MCF`` /
``MCLAF`` / ``AugGaussSumFilt`` are out of scope.
"""
import ctypes as C

import numpy as np

from . import _lib
from .models import ParamsNLSSM, ParamsBPF, NonlinearSSM
from .nonlinearities import DeviceFunction, GaussianLogProb, linear_emission, require_device_function
from . import inference as _inf

F32 = np.float32


class _LegacyFn:
    """Registry function called the legacy way, ``f(x)``."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, x):
        return self.fn(x, np.zeros(self.fn.noise_dim, F32), None)


class SSM:
    """gaussfiltax/gaussfilt.py:10-52: x' = f(x) + N(c, Q), y = g(x') + N(d, R)."""

    def __init__(self, dx, dy, c, Q, d, R, f=None, g=None):
        self.dx, self.dy = dx, dy
        self._f = require_device_function(f, "dynamics", "f")
        self._g = require_device_function(g, "emission", "g")
        if self._f.noise_dim != dx or self._g.noise_dim != dy:
            raise ValueError("legacy SSM: noise enters additively with identity matrices (f(x) + q, g(x) + r)")
        self.f, self.g = _LegacyFn(self._f), _LegacyFn(self._g)
        self.Q, self.R = np.asarray(Q, F32), np.asarray(R, F32)
        self.c, self.d = np.asarray(c, F32), np.asarray(d, F32)

    def _params(self, m0, P0, with_bias):
        z = lambda v: v if with_bias else np.zeros_like(v)
        return ParamsNLSSM(np.asarray(m0, F32), np.asarray(P0, F32), self._f, z(self.c), self.Q, self._g, z(self.d), self.R)

    def simulate(self, T, x0, key=None):
        """gaussfilt.py:22-43: T steps from x0 (the first emitted state is already propagated)."""
        key = _inf.PRNGKey(0) if key is None else key
        # NonlinearSSM.sample emits the initial state first: start it AT x0 with a vanishing initial
        # covariance and drop that first sample
        p = self._params(x0, 1e-30 * np.eye(self.dx, dtype=F32), with_bias=True)
        xs, ys = NonlinearSSM(self.dx, self.dx, self.dy, self.dy).sample(p, key, T + 1)
        return xs[1:].cpu().numpy(), ys[1:].cpu().numpy()


def _run_gsf(ssm, ys, init_means, P0, flags, K):
    """bf_gsf_ekf_f32 with legacy flags; returns torch tensors (K,T,...) + loglik."""
    import torch
    params = ssm._params(init_means[0], P0, with_bias=False)
    lib = _lib.require_gpu()
    mdl = _inf._Model(params)
    mdl.c.flags = flags
    n, m = mdl.n, mdl.m
    y = _inf._dev_f32(ys, "cuda").reshape(1, -1, m)
    T = y.shape[1]
    m_in = _inf._dev_f32(init_means, "cuda").reshape(1, K, n).contiguous()
    P_in = _inf._dev_f32(P0, "cuda").reshape(1, 1, n, n).expand(1, K, n, n).contiguous()
    bufs, ll, od = _inf._alloc_outputs(1, K, T, n, _inf.FILTERED, "reference", None, True, y.device)
    yd = _lib.bf_cstream()
    yd.ptr, yd.sB, yd.sK, yd.sT, yd.sE = y.data_ptr(), y.stride(0), 0, y.stride(1), y.stride(2)
    ud = _lib.bf_cstream()
    cr = _lib.bf_carry()
    cr.m_in, cr.P_in = m_in.data_ptr(), P_in.data_ptr()
    stream = torch.cuda.current_stream(y.device).cuda_stream
    _lib.check(lib.bf_gsf_ekf_f32(C.byref(mdl.c), C.byref(yd), C.byref(ud), 1, T, K, C.byref(cr), C.byref(od), C.c_void_p(stream)))
    return bufs, ll


class EKF:
    """gaussfiltax/gaussfilt.py:201-252 (first-order moments) with GaussFilt.run (:88-130)."""

    def __init__(self, ssm, order=1):
        if order != 1:
            raise NotImplementedError("second-order terms are commented out in the reference (gaussfilt.py:241-247)")
        self.ssm = ssm
        self.dx, self.dy = ssm.dx, ssm.dy

    def __str__(self):
        return "EKF"

    def run(self, ys, m0, P0, verbose=False):
        """-> (ll (T,), filtered_means (T, dx), filtered_covs (T, dx, dx)) as NumPy arrays."""
        flags = _lib.BF_MODEL_PREDICT_FIRST | _lib.BF_MODEL_NO_JITTER
        bufs, ll = _run_gsf(self.ssm, ys, np.asarray(m0, F32).reshape(1, -1), P0, flags, 1)
        return ll[0, 0].cpu().numpy(), bufs["means"][0, 0].cpu().numpy(), bufs["covariances"][0, 0].cpu().numpy()


class GaussSumFilt:
    """gaussfiltax/gausssumfilt.py:11-78."""

    def __init__(self, ssm, M):
        self.ssm, self.M = ssm, int(M)
        self.dx, self.dy = ssm.dx, ssm.dy

    def __str__(self):
        return "GSF"

    def run(self, ys, m0, P0, verbose=False, key=None, initial_means=None):
        """-> (means (T+1, dx, M), covs (T+1, dx, dx, M), weights (T+1, M), point_est (T, dx)); the
        initial components sit at index T (= -1).  Initial means are m0 + N(0, I) (gausssumfilt.py:46-48),
        drawn from the engine's Threefry stream of ``key`` or passed as ``initial_means`` (M, dx)."""
        M, n = self.M, self.dx
        if initial_means is None:
            z = _inf._random_normal(_inf.PRNGKey(0) if key is None else key, M * n).reshape(M, n)
            initial_means = np.asarray(m0, F32)[None, :] + z
        initial_means = np.asarray(initial_means, F32).reshape(M, n)
        flags = _lib.BF_MODEL_PREDICT_FIRST | _lib.BF_MODEL_NO_JITTER | _lib.BF_MODEL_LEGACY_GSF_COV
        bufs, _ = _run_gsf(self.ssm, ys, initial_means, P0, flags, M)
        T = len(ys)
        means = np.zeros((T + 1, n, M), F32)
        covs = np.zeros((T + 1, n, n, M), F32)
        weights = np.zeros((T + 1, M), F32)
        means[:T] = bufs["means"][0].permute(1, 2, 0).cpu().numpy()
        covs[:T] = bufs["covariances"][0].permute(1, 2, 3, 0).cpu().numpy()
        weights[:T] = bufs["weights"][0].permute(1, 0).cpu().numpy()
        means[T], covs[T], weights[T] = initial_means.T, np.asarray(P0, F32)[:, :, None], F32(1.0) / F32(M)
        # gausssumfilt.py:76: np.sum without an axis -> one scalar per step, broadcast over the coordinates
        pe = np.repeat(np.sum(means[:T] * weights[:T, None, :], axis=(1, 2))[:, None], n, axis=1).astype(F32)
        return means, covs, weights, pe


class BootstrapPF:
    """gaussfiltax/particlefilt.py:11-52: weights are not carried and every step resamples."""

    def __init__(self, ssm, N):
        self.ssm, self.N = ssm, int(N)
        self.dx, self.dy = ssm.dx, ssm.dy

    def __str__(self):
        return "BPF"

    def run(self, ys, m0, P0, verbose=False, key=None):
        """-> particles (T+1, N, dx); index T holds the initial draw from N(m0, P0) (:30)."""
        import torch
        s = self.ssm
        p = ParamsBPF(np.asarray(m0, F32), np.asarray(P0, F32), s._f, np.zeros(s.dx, F32), s.Q, s._g, np.zeros(s.dy, F32), s.R,
                      GaussianLogProb(s._g, s.R))
        key = _inf.PRNGKey(0) if key is None else key
        # ess_threshold = 2 makes `ess < threshold * N` always true: resample at every step
        out = _inf.bootstrap_particle_filter(p, ys, self.N, key, None, 2.0, return_ancestors=True)
        x, anc = out["particles"], out["ancestors"].long()           # (N, T, dx), (N, T)
        # multinomial counts lay the survivors out in ascending ancestor order (:45-50)
        order = torch.argsort(anc, dim=0, stable=True)
        x = torch.gather(x, 0, order[:, :, None].expand(-1, -1, self.dx))
        T = x.shape[1]
        particles = np.zeros((T + 1, self.N, self.dx), F32)
        particles[:T] = x.permute(1, 0, 2).cpu().numpy()
        # the initial particles of the engine's run (keys[1+i] of split(key, N+1), inference.py:1369-1373)
        z = np.stack([_inf._random_normal(k, self.dx) for k in _split(key, self.N + 1)[1:]])
        particles[T] = np.asarray(m0, F32)[None, :] + z @ np.linalg.cholesky(np.asarray(P0, np.float64)).astype(F32).T
        return particles


def _split(key, num):
    lib = _lib.load()
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2))
    out = np.empty((num, 2), dtype=np.uint32)
    _lib.check(lib.bf_random_split(key.ctypes.data_as(C.POINTER(C.c_uint32)), num, out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out
