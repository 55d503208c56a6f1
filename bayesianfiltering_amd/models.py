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
