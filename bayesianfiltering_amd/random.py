"""``import bayesianfiltering_amd.random as jr`` -- the handful of ``jax.random`` calls the reference's scripts make around the
filters (docs/experiments/*.py: ``jr.PRNGKey``, ``jr.split``, ``jr.normal``, ``jr.multivariate_normal``), on the engine's own
Threefry-2x32 (the library's host functions ``bf_random_split`` / ``bf_random_normal_f32``: JAX's counter layout and bits -> normal
mapping, pinned against keys and draws the reference recorded -- tests/test_oracle_rng.py, tests/test_sample_gpu.py).  Keys are
``(2,) uint32`` arrays as in JAX's raw form.
"""
import ctypes as C

import numpy as np

from . import _lib
from .inference import PRNGKey, _random_normal   # noqa: F401  (PRNGKey re-exported)

F32 = np.float32


def split(key, num: int = 2):
    """``jax.random.split``: (num, 2) uint32; ``k1, k2 = split(key)`` unpacks as with JAX."""
    lib = _lib.load()
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32).reshape(2))
    out = np.empty((int(num), 2), dtype=np.uint32)
    _lib.check(lib.bf_random_split(key.ctypes.data_as(C.POINTER(C.c_uint32)), int(num), out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def normal(key, shape=()):
    """``jax.random.normal(key, shape)`` in float32."""
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
    count = int(np.prod(shape)) if shape else 1
    z = _random_normal(key, count)
    return z.reshape(shape) if shape else F32(z[0])


def multivariate_normal(key, mean, cov, shape=()):
    """``jax.random.multivariate_normal(key, mean, cov, shape)``: ``mean + chol(cov) @ normal(key, shape + (d,))``."""
    mean = np.asarray(mean, dtype=F32)
    d = mean.shape[-1]
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
    L = np.linalg.cholesky(np.asarray(cov, dtype=np.float64)).astype(F32)
    z = normal(key, shape + (d,))
    return (mean + z @ L.T).astype(F32)
