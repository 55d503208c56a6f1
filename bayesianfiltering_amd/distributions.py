"""``MVN(loc=..., covariance_matrix=...).log_prob(y)`` for recorded log-densities.

The reference writes its particle-filter densities as
``lambda x, y, u: MVN(loc=g(x, r0, u), covariance_matrix=R).log_prob(y)`` with tensorflow_probability's
``MultivariateNormalFullCovariance`` (docs/experiments/BOT_Experiment_script.py:45, adaptive_experiment.py:55-57).  This class has
that constructor and ``log_prob`` written with operations :mod:`bayesianfiltering_amd.trace` can record -- an unrolled Cholesky
factorization, forward substitution and the log-determinant -- so the mean AND the covariance may depend on the state; with plain
numbers it is an ordinary NumPy evaluation of the same formula.
"""
import numpy as np


class MVN:
    def __init__(self, loc=None, covariance_matrix=None, scale_tril=None):
        if (covariance_matrix is None) == (scale_tril is None):
            raise ValueError("give covariance_matrix or scale_tril")
        self.loc = np.atleast_1d(np.asarray(loc, dtype=object if _symbolic(loc) else np.float64))
        self._L = None if scale_tril is None else np.atleast_2d(np.asarray(scale_tril, dtype=object if _symbolic(scale_tril) else np.float64))
        self._C = None if covariance_matrix is None else np.atleast_2d(
            np.asarray(covariance_matrix, dtype=object if _symbolic(covariance_matrix) else np.float64))

    def _chol(self):
        if self._L is not None:
            return self._L
        C = self._C
        m = C.shape[0]
        if C.dtype != object:
            return np.linalg.cholesky(C)
        L = np.zeros((m, m), dtype=object)
        for j in range(m):
            d = C[j, j]
            for k in range(j):
                d = d - L[j, k] * L[j, k]
            L[j, j] = np.sqrt(d)
            for i in range(j + 1, m):
                s = C[i, j]
                for k in range(j):
                    s = s - L[i, k] * L[j, k]
                L[i, j] = s / L[j, j]
        return L

    def log_prob(self, y):
        L = self._chol()
        m = L.shape[0]
        d = np.atleast_1d(np.asarray(y, dtype=object if _symbolic(y) else np.float64)) - self.loc
        z = [None] * m
        quad, logdet = 0.0, 0.0
        for i in range(m):                      # forward substitution L z = y - loc
            s = d[i]
            for k in range(i):
                s = s - L[i, k] * z[k]
            z[i] = s / L[i, i]
            quad = quad + z[i] * z[i]
            logdet = logdet + np.log(L[i, i])
        return -0.5 * quad - logdet - 0.5 * m * float(np.log(2.0 * np.pi))


MultivariateNormalFullCovariance = MVN


def _symbolic(v):
    a = np.asarray(v, dtype=object).ravel() if not isinstance(v, np.ndarray) or v.dtype == object else ()
    from .trace import Sym
    return any(isinstance(e, Sym) for e in a)
