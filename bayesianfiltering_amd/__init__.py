"""bayesianfiltering_amd: MI355X-native batched Bayesian filtering (Kalman / Gaussian-sum /
bootstrap particle filters) behind the call surface of the reference package ``gaussfiltax``
(kostastsa/BayesianFiltering).  See DESIGN.md for the hot path and its boundary."""
from .models import ParamsNLSSM, ParamsBPF, NonlinearSSM
from .containers import GaussianComponent, GaussianSum
from .inference import (PosteriorGaussianSumFiltered, gaussian_sum_filter, unscented_gaussian_sum_filter, ParamsUKF,
                        speedy_augmented_gaussian_sum_filter, augmented_gaussian_sum_filter,
                        speedy_unscented_agsf, unscented_agsf, augmented_gaussian_sum_filter_optimal, optimal_resampling,
                        kalman_filter, FilterCarry,
                        FULL5, FILTERED, PRNGKey, sample_initial_component_means,
                        bootstrap_particle_filter, ParticleCarry, resample_indices)
from ._lib import BayesFiltError
from . import nonlinearities, utils

__all__ = ["ParamsNLSSM", "ParamsBPF", "NonlinearSSM", "GaussianComponent", "GaussianSum", "PosteriorGaussianSumFiltered",
           "gaussian_sum_filter", "unscented_gaussian_sum_filter", "ParamsUKF", "speedy_augmented_gaussian_sum_filter", "augmented_gaussian_sum_filter", "speedy_unscented_agsf", "unscented_agsf", "augmented_gaussian_sum_filter_optimal", "optimal_resampling", "kalman_filter", "FilterCarry", "FULL5", "FILTERED", "PRNGKey",
           "sample_initial_component_means", "bootstrap_particle_filter", "ParticleCarry", "resample_indices",
           "nonlinearities", "utils", "BayesFiltError"]
