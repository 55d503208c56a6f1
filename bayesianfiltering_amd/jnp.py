"""``import bayesianfiltering_amd.jnp as jnp`` -- NumPy under the name the reference's model functions use.

The reference writes its f / h / log-density lambdas with ``jax.numpy`` (docs/experiments/*.py).  For recording them
(:mod:`bayesianfiltering_amd.trace`) NumPy is the drop-in, except for selects: ``jnp.where(cond, a, b)`` on the state has to
become a recorded select, which :func:`bayesianfiltering_amd.trace.where` is.  This module is NumPy with that one replacement
(and the array comparisons that go with it), so a script changes its import line and nothing else.
"""
from numpy import *            # noqa: F401,F403
import numpy as _np
from .trace import where, greater, less, greater_equal, less_equal   # noqa: F401  (override numpy's)

linalg = _np.linalg
float32, float64, int32 = _np.float32, _np.float64, _np.int32
