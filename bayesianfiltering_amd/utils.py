"""Small helpers of gaussfiltax/utils.py that sit on the filtering path's edges."""
import numpy as np


def mse(x_est, x_base):
    """gaussfiltax/utils.py:179-182."""
    x_est, x_base = np.asarray(x_est), np.asarray(x_base)
    return np.sum((x_est - x_base) ** 2) / x_est.shape[0]


def rmse(x_est, x_base):
    """gaussfiltax/utils.py:184-187."""
    x_est, x_base = np.asarray(x_est), np.asarray(x_base)
    return np.sqrt(np.sum((x_est - x_base) ** 2) / x_est.shape[0])


def _post_desc(post):
    w, m, P = post.weights, post.means, post.covariances
    if w.dim() == 2:                       # (K, T, ...) single trajectory -> add the batch axis
        w, m, P = w.unsqueeze(0), m.unsqueeze(0), (P.unsqueeze(0) if P is not None else None)
    return w, m, P


def collapse_posterior(post, with_covariance: bool = True):
    """Moment-matched single Gaussian of the mixture posterior at every step, on the device:
    ``mu_t = sum_k w_k m_k`` (the point estimate of BOT_Experiment_script.py:101) and
    ``Sigma_t = sum_k w_k (P_k + (m_k - mu)(m_k - mu)^T)`` (gaussfiltax/utils.py:10-18).
    ``post``: PosteriorGaussianSumFiltered from this package ((K,T,..) or (B,K,T,..) tensors).
    Returns (means (.., T, n), covariances (.., T, n, n) or None)."""
    import ctypes as C
    import torch
    from . import _lib
    from .inference import _stream_desc
    lib = _lib.require_gpu()
    squeeze = post.weights.dim() == 2
    w, m, P = _post_desc(post)
    B, K, T = w.shape
    n = m.shape[-1]
    mean = torch.empty((B, T, n), dtype=torch.float32, device=w.device)
    cov = torch.empty((B, T, n, n), dtype=torch.float32, device=w.device) if (with_covariance and P is not None) else None
    wd, md = _stream_desc(w, 0), _stream_desc(m, 1)
    Pd = _stream_desc(P, 2) if cov is not None else _lib.bf_stream()
    stream = torch.cuda.current_stream(w.device).cuda_stream
    _lib.check(lib.bf_collapse_f32(C.byref(wd), C.byref(md), C.byref(Pd), B, T, K, n, mean.data_ptr(),
                                   cov.data_ptr() if cov is not None else None, C.c_void_p(stream)))
    if squeeze:
        return mean[0], (cov[0] if cov is not None else None)
    return mean, cov


def collapse(mean_mat, covariance_tens, weight_vec):
    """gaussfiltax/utils.py:10-18 with the reference's signature: one mixture (M, dx), (M, dx, dx),
    (M,) -> (mean (dx,), covariance (dx, dx)); evaluated by the device kernel."""
    import torch
    from .inference import PosteriorGaussianSumFiltered, _dev_f32
    m = _dev_f32(mean_mat, "cuda")
    P = _dev_f32(covariance_tens, "cuda")
    w = _dev_f32(weight_vec, "cuda")
    post = PosteriorGaussianSumFiltered(weights=w.reshape(-1, 1).contiguous(), means=m.unsqueeze(1).contiguous(),
                                        covariances=P.unsqueeze(1).contiguous())
    mean, cov = collapse_posterior(post)
    return mean[0], cov[0]
