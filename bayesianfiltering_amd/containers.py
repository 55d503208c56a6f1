"""Mixture containers (gaussfiltax/containers.py:17-61): AoS <-> SoA views of a Gaussian sum."""
from typing import NamedTuple, Any, List


class GaussianComponent(NamedTuple):
    """gaussfiltax/containers.py:17-23."""
    mean: Any
    covariance: Any
    weight: Any


class GaussianSum(NamedTuple):
    """gaussfiltax/containers.py:32-41."""
    means: Any
    covariances: Any
    weights: Any

    def _sum_weights(self):
        return sum(w for w in self.weights) if isinstance(self.weights, (list, tuple)) else self.weights.sum()

    def _check_normalization(self):
        return abs(float(self._sum_weights()) - 1.0) <= 1e-5 + 1e-8


def _gaussian_sum_to_components(gaussian_sum: GaussianSum) -> List[GaussianComponent]:
    """gaussfiltax/containers.py:43-44."""
    return [GaussianComponent(m, P, w)
            for m, P, w in zip(gaussian_sum.means, gaussian_sum.covariances, gaussian_sum.weights)]


def _components_to_gaussian_sum(components) -> GaussianSum:
    """gaussfiltax/containers.py:46-61 (lists, as the reference returns)."""
    return GaussianSum([c.mean for c in components], [c.covariance for c in components],
                       [c.weight for c in components])
