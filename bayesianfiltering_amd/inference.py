"""Filter drivers with the reference's call surface, running on the HIP engine.

Mirrors ``gaussfiltax/inference.py``:

* :class:`PosteriorGaussianSumFiltered`   -- inference.py:29-39 (same fields, same order)
* :func:`gaussian_sum_filter`             -- inference.py:303-377
* :func:`bootstrap_particle_filter`       -- inference.py:1302-1380

What is new is the batch axis: ``emissions`` may be ``(T, m)`` like the reference (one
trajectory; outputs shaped ``(K, T, ...)`` exactly as the reference returns them after
``swap_axes_on_values``, inference.py:372) or ``(B, T, m)`` for B independent trajectories
(outputs ``(B, K, T, ...)``).  Outputs are ``torch`` tensors on the GPU.  PyTorch is plumbing
only (device memory + streams); all arithmetic happens in the HIP kernels behind the C-ABI of
``include/bayesfilt.h``.  There is no CPU path: without the built library or without an
MI355X the calls raise.
"""
import ctypes as C
from typing import NamedTuple, Optional, Any, Sequence

import numpy as np

from . import _lib
from .nonlinearities import DeviceFunction, DYN_LINEAR, EMI_LINEAR, require_device_function

F32 = np.float32
FULL5 = ("weights", "means", "covariances", "predicted_means", "predicted_covariances")
FILTERED = ("weights", "means", "covariances")


class PosteriorGaussianSumFiltered(NamedTuple):
    """Marginals of the Gaussian-sum filtering posterior (gaussfiltax/inference.py:29-39)."""
    weights: Optional[Any] = None
    means: Optional[Any] = None
    covariances: Optional[Any] = None
    predicted_means: Optional[Any] = None
    predicted_covariances: Optional[Any] = None


def _torch():
    import torch
    return torch


def _dev_f32(x, device):
    """float32 tensor on ``device`` (no copy if it already is one)."""
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x, dtype=F32), device=device)


def _host_f32(x):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=F32))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _alloc_stream(shape_bkt, event_shape, layout, device):
    """Allocate one output stream with logical shape (B, K, T, *event) in the chosen physical layout."""
    torch = _torch()
    B, K, T = shape_bkt
    ev = tuple(event_shape)
    if layout == "reference":
        return torch.empty((B, K, T) + ev, dtype=torch.float32, device=device)
    if layout == "batch_inner":
        phys = torch.empty((K, T) + ev + (B,), dtype=torch.float32, device=device)
        nd = phys.dim()
        return phys.permute((nd - 1,) + tuple(range(nd - 1)))
    raise ValueError(f"unknown layout {layout!r}")


def _stream_desc(t, n_event_dims):
    """bf_stream for a (B, K, T, *event) tensor whose event dims flatten with one stride."""
    s = _lib.bf_stream()
    if t is None:
        return s
    st = t.stride()
    if n_event_dims == 0:
        sE = 1
    elif n_event_dims == 1:
        sE = st[3]
    else:
        if st[3] != st[4] * t.shape[4]:
            raise ValueError("matrix output must be row-major-flattenable")
        sE = st[4]
    s.ptr, s.sB, s.sK, s.sT, s.sE = t.data_ptr(), st[0], st[1], st[2], sE
    return s


def _time_varying(x, d):
    x = _host_f32(x)
    if x.ndim == 3:
        return x, x.shape[0]
    if x.shape != (d, d):
        raise ValueError(f"covariance must be ({d},{d}) or (T,{d},{d}); got {x.shape}")
    return x, 1


class _Lgssm:
    """Host-side build of bf_lgssm from a ParamsNLSSM with linear registry functions."""

    def __init__(self, params):
        f = require_device_function(params.dynamics_function, "dynamics", "params.dynamics_function")
        h = require_device_function(params.emission_function, "emission", "params.emission_function")
        if f.fn_id != DYN_LINEAR or h.fn_id != EMI_LINEAR:
            raise ValueError("kalman_filter needs linear_dynamics / linear_emission functions")
        self.n, self.dq, self.m, self.dr = f.out_dim, f.noise_dim, h.out_dim, h.noise_dim
        if f.in_dim != self.n or h.in_dim != self.n:
            raise ValueError("dynamics / emission matrices do not match the state dimension")
        self.A, self.G = np.ascontiguousarray(f.M), np.ascontiguousarray(f.N)
        self.H, self.D = np.ascontiguousarray(h.M), np.ascontiguousarray(h.N)
        self.q0 = _host_f32(params.dynamics_noise_bias).reshape(self.dq)
        self.r0 = _host_f32(params.emission_noise_bias).reshape(self.dr)
        self.Q, self.Q_steps = _time_varying(params.dynamics_noise_covariance, self.dq)
        self.R, self.R_steps = _time_varying(params.emission_noise_covariance, self.dr)
        c = _lib.bf_lgssm()
        c.n, c.dq, c.m, c.dr = self.n, self.dq, self.m, self.dr
        c.A, c.G, c.H, c.D = _fp(self.A), _fp(self.G), _fp(self.H), _fp(self.D)
        c.q0, c.r0, c.Q, c.R = _fp(self.q0), _fp(self.r0), _fp(self.Q), _fp(self.R)
        c.Q_steps, c.R_steps = self.Q_steps, self.R_steps
        self.c = c


class FilterCarry(NamedTuple):
    """The scan carry (weights, pred_means, pred_covs) of inference.py:334,356 at the end of a
    chunk; feed it back through ``carry=`` to continue the same trajectories."""
    weights: Any
    means: Any
    covariances: Any


def kalman_filter(params, emissions, *, initial_means=None, initial_covariances=None, carry=None,
                  fields: Sequence[str] = FULL5, layout: str = "reference", out=None,
                  return_loglik: bool = False, return_carry: bool = False, device="cuda"):
    """Batched Kalman filter == ``gaussian_sum_filter(params, y, num_components=1)`` of the
    reference for linear ``f(x,q,u) = A x + G q``, ``h(x,r,u) = H x + D r`` with the initial
    component mean given explicitly (the reference samples it from N(m0, P0) with PRNGKey(0),
    inference.py:367; pass ``initial_means=`` to choose it, default ``params.initial_mean``).

    emissions: (T, m) or (B, T, m).  Returns PosteriorGaussianSumFiltered with arrays shaped
    (1, T, ...) or (B, 1, T, ...); ``layout='batch_inner'`` returns the same logical shapes as
    strided views of a [K][T][E][B] buffer (the fastest store pattern).  ``out`` may hold a
    previously returned posterior whose buffers are reused.
    """
    torch = _torch()
    lib = _lib.require_gpu()
    mdl = _Lgssm(params)
    n, m = mdl.n, mdl.m
    y = _dev_f32(emissions, device)
    squeeze = y.dim() == 2
    if squeeze:
        y = y.unsqueeze(0)
    if y.dim() != 3 or y.shape[2] != m:
        raise ValueError(f"emissions must be (T,{m}) or (B,T,{m}); got {tuple(y.shape)}")
    B, T = int(y.shape[0]), int(y.shape[1])
    if T == 0 or B == 0:
        raise ValueError("empty emissions")

    # carry in
    if carry is not None:
        w_in, m_in, P_in = (_dev_f32(v, device).contiguous() for v in carry)
        w_in = w_in.reshape(B, 1)
    else:
        w_in = None
        if initial_means is None:
            m_in = _dev_f32(_host_f32(params.initial_mean), device).reshape(1, 1, n).expand(B, 1, n).contiguous()
        else:
            m_in = _dev_f32(initial_means, device).reshape(-1, 1, n)
            m_in = m_in.expand(B, 1, n).contiguous() if m_in.shape[0] == 1 else m_in.contiguous()
        P0 = params.initial_covariance if initial_covariances is None else initial_covariances
        P_in = _dev_f32(P0, device).reshape(-1, 1, n, n)
        P_in = P_in.expand(B, 1, n, n).contiguous() if P_in.shape[0] == 1 else P_in.contiguous()
    if m_in.shape != (B, 1, n) or P_in.shape != (B, 1, n, n):
        raise ValueError("initial means / covariances do not match (B, 1, n) / (B, 1, n, n)")

    ev = {"weights": (), "means": (n,), "covariances": (n, n), "predicted_means": (n,), "predicted_covariances": (n, n)}
    bufs = {}
    for name in FULL5:
        if name in fields:
            reuse = getattr(out, name, None) if out is not None else None
            if reuse is not None:
                if tuple(reuse.shape) != (B, 1, T) + ev[name]:
                    raise ValueError(f"out.{name} has shape {tuple(reuse.shape)}")
                bufs[name] = reuse
            else:
                bufs[name] = _alloc_stream((B, 1, T), ev[name], layout, y.device)
        else:
            bufs[name] = None
    ll = _alloc_stream((B, 1, T), (), layout, y.device) if return_loglik else None

    od = _lib.bf_out_desc()
    od.weights = _stream_desc(bufs["weights"], 0)
    od.means = _stream_desc(bufs["means"], 1)
    od.covs = _stream_desc(bufs["covariances"], 2)
    od.pred_means = _stream_desc(bufs["predicted_means"], 1)
    od.pred_covs = _stream_desc(bufs["predicted_covariances"], 2)
    od.loglik = _stream_desc(ll, 0)

    yd = _lib.bf_cstream()
    yd.ptr, yd.sB, yd.sK, yd.sT, yd.sE = y.data_ptr(), y.stride(0), 0, y.stride(1), y.stride(2)

    cr = _lib.bf_carry()
    cr.w_in = w_in.data_ptr() if w_in is not None else None
    cr.m_in, cr.P_in = m_in.data_ptr(), P_in.data_ptr()
    c_out = None
    if return_carry:
        c_out = FilterCarry(torch.empty((B, 1), dtype=torch.float32, device=y.device),
                            torch.empty((B, 1, n), dtype=torch.float32, device=y.device),
                            torch.empty((B, 1, n, n), dtype=torch.float32, device=y.device))
        cr.w_out, cr.m_out, cr.P_out = (t.data_ptr() for t in c_out)

    stream = torch.cuda.current_stream(y.device).cuda_stream
    _lib.check(lib.bf_kalman_filter_f32(C.byref(mdl.c), C.byref(yd), B, T, C.byref(cr), C.byref(od),
                                        C.c_void_p(stream)))

    post = PosteriorGaussianSumFiltered(**{k: (v[0] if (squeeze and v is not None) else v) for k, v in bufs.items()})
    extras = []
    if return_loglik:
        extras.append(ll[0] if squeeze else ll)
    if return_carry:
        extras.append(c_out)
    return (post, *extras) if extras else post


def gaussian_sum_filter(params, emissions, num_components: int = 1, num_iter: int = 1, inputs=None, *,
                        initial_means=None, **kw):
    """Gaussian-sum filter (bank of K extended Kalman filters + weight update),
    gaussfiltax/inference.py:303-377.  ``num_iter`` is accepted and ignored exactly as in the
    reference (:307, never read).  ``initial_means`` (K, n) / (B, K, n) overrides the
    reference's fixed ``MVN(m0, P0).sample(K, PRNGKey(0))`` draw (:367).
    """
    f = require_device_function(params.dynamics_function, "dynamics", "params.dynamics_function")
    h = require_device_function(params.emission_function, "emission", "params.emission_function")
    if num_components == 1 and f.fn_id == DYN_LINEAR and h.fn_id == EMI_LINEAR and inputs is None \
            and initial_means is not None:
        return kalman_filter(params, emissions, initial_means=initial_means, **kw)
    raise _lib.BayesFiltError(_lib.BF_EUNSUPPORTED,
                              "gaussian_sum_filter: the nonlinear / multi-component HIP kernel is not built yet")
