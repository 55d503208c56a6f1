"""Mixture containers (gaussfiltax/containers.py:17-61): field names / order and the AoS <-> SoA converters."""
import numpy as np

from bayesianfiltering_amd import containers as ct
from bayesianfiltering_amd import GaussianComponent, GaussianSum


def _mixture(K=3, n=2, seed=0):
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(K, n)).astype(np.float32)
    L = rng.normal(size=(K, n, n)).astype(np.float32)
    covs = L @ L.transpose(0, 2, 1) + np.eye(n, dtype=np.float32)
    w = rng.random(K).astype(np.float32)
    return means, covs, (w / w.sum()).astype(np.float32)


def test_field_names_match_the_reference():
    assert GaussianComponent._fields == ("mean", "covariance", "weight")          # containers.py:20-22
    assert GaussianSum._fields == ("means", "covariances", "weights")             # containers.py:36-38


def test_round_trip_sum_components_sum():
    means, covs, w = _mixture()
    gs = GaussianSum(means, covs, w)
    comps = ct._gaussian_sum_to_components(gs)                                    # containers.py:43-44
    assert len(comps) == 3 and all(isinstance(c, GaussianComponent) for c in comps)
    for k, c in enumerate(comps):
        assert np.array_equal(c.mean, means[k]) and np.array_equal(c.covariance, covs[k]) and c.weight == w[k]
    back = ct._components_to_gaussian_sum(comps)                                  # containers.py:46-61: lists
    assert isinstance(back.means, list) and isinstance(back.covariances, list) and isinstance(back.weights, list)
    assert np.array_equal(np.stack(back.means), means)
    assert np.array_equal(np.stack(back.covariances), covs)
    assert np.array_equal(np.array(back.weights), w)
    # and once more through the list form
    again = ct._components_to_gaussian_sum(ct._gaussian_sum_to_components(back))
    assert np.array_equal(np.stack(again.means), means) and np.array_equal(np.array(again.weights), w)


def test_normalisation_check():
    means, covs, w = _mixture(K=5)
    assert GaussianSum(means, covs, w)._check_normalization()
    assert GaussianSum(list(means), list(covs), list(w))._check_normalization()
    assert not GaussianSum(means, covs, 2 * w)._check_normalization()
    assert abs(float(GaussianSum(means, covs, w)._sum_weights()) - 1.0) < 1e-6
