"""Parameter containers with the reference's field names and order.

``ParamsNLSSM`` mirrors gaussfiltax/models.py:26-51 and ``ParamsBPF`` gaussfiltax/models.py:55-84.
The only difference: the function fields hold :class:`~.nonlinearities.DeviceFunction`
objects (host-callable with the same ``(x, noise, u)`` signature) instead of arbitrary
lambdas, and ``emission_distribution_log_prob`` holds a
:class:`~.nonlinearities.GaussianLogProb`.
"""
from typing import NamedTuple, Any


class ParamsNLSSM(NamedTuple):
    initial_mean: Any
    initial_covariance: Any
    dynamics_function: Any
    dynamics_noise_bias: Any
    dynamics_noise_covariance: Any
    emission_function: Any
    emission_noise_bias: Any
    emission_noise_covariance: Any


class ParamsBPF(NamedTuple):
    initial_mean: Any
    initial_covariance: Any
    dynamics_function: Any
    dynamics_noise_bias: Any
    dynamics_noise_covariance: Any
    emission_function: Any
    emission_noise_bias: Any
    emission_noise_covariance: Any
    emission_distribution_log_prob: Any


class NonlinearSSM:
    """Sampling half of the reference's model class (gaussfiltax/models.py:160-289): synthetic
    states and emissions of a registry state-space model, generated on the device.  The fitting
    half (EM / SGD, gaussfiltax/ssm.py) is out of scope."""

    def __init__(self, state_dim: int, state_noise_dim: int, emission_dim: int, emission_noise_dim: int, input_dim: int = 0):
        self.state_dim, self.state_noise_dim = state_dim, state_noise_dim
        self.emission_dim, self.emission_noise_dim = emission_dim, emission_noise_dim
        self.input_dim = input_dim

    def sample(self, params, key, num_timesteps: int, inputs=None):
        """``model.sample(params, key, T, inputs)`` of models.py:240-289 -> (states (T, n), emissions (T, m)).
        ``key`` (2,) uint32 for one trajectory, or (B, 2) for B independent ones (-> (B, T, ...))."""
        import ctypes as C
        import numpy as np
        import torch
        from . import _lib
        from .inference import _Model, _host_f32, _fp, _dev_f32
        lib = _lib.require_gpu()
        mdl = _Model(params)
        if (mdl.n, mdl.dq, mdl.m, mdl.dr) != (self.state_dim, self.state_noise_dim, self.emission_dim, self.emission_noise_dim):
            raise ValueError("params do not match the dimensions this NonlinearSSM was built with")
        keys = np.ascontiguousarray(np.asarray(key, dtype=np.uint32))
        squeeze = keys.ndim == 1
        keys = keys.reshape(-1, 2)
        B, T, n, m = keys.shape[0], int(num_timesteps), mdl.n, mdl.m
        bm = _lib.bf_bpf_model()
        bm.ssm = mdl.c
        m0 = _host_f32(params.initial_mean).reshape(n)
        P0 = _host_f32(params.initial_covariance).reshape(n, n)
        bm.m0, bm.P0 = _fp(m0), _fp(P0)
        dkeys = torch.as_tensor(keys.view(np.int32), device="cuda")
        states = torch.empty((B, T, n), dtype=torch.float32, device="cuda")
        emis = torch.empty((B, T, m), dtype=torch.float32, device="cuda")
        ud = _lib.bf_cstream()
        u_keep = None
        if inputs is not None:
            u_keep = _dev_f32(inputs, "cuda")
            u_keep = u_keep.reshape(1, T, -1) if u_keep.dim() <= 2 else u_keep
            ud.ptr, ud.sB, ud.sT, ud.sE = u_keep.data_ptr(), (u_keep.stride(0) if u_keep.shape[0] == B else 0), u_keep.stride(1), 1
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.bf_sample_ssm_f32(C.byref(bm), dkeys.data_ptr(), C.byref(ud), B, T, states.data_ptr(), emis.data_ptr(),
                                         C.c_void_p(stream)))
        return (states[0], emis[0]) if squeeze else (states, emis)
