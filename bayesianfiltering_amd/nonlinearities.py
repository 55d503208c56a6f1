"""Model-function registry: the device-side replacement for the Python callables of the reference.

In the reference, ``ParamsNLSSM.dynamics_function`` / ``emission_function`` are arbitrary Python
lambdas ``f(x, q, u)``, ``h(x, r, u)`` (gaussfiltax/models.py:46-49) differentiated by ``jacfwd``
(gaussfiltax/inference.py:328-329).  Python callables cannot cross the C-ABI, so the functions
the reference actually uses are enumerated here; each factory returns a :class:`DeviceFunction`
that (a) is callable on the host with the reference's signature (NumPy, for simulation and
inspection only -- never used by the filters) and (b) carries the ``fn_id`` / ``theta`` the HIP
kernels dispatch on, where value and analytic Jacobians are evaluated per lane.

Catalogue (reference sources):
  linear_dynamics / linear_emission   docs/experiments/adaptive_experiment.py:59-64,
                                      docs/experiments/BOT_Experiment_script.py:31-41
  lorenz96 / pick_even (f96 / g96)    gaussfiltax/nonlinearities.py:37-50
  lorenz63                            docs/experiments/exp_lorentz63.py:37-41
  maneuver_bot / bearing_range        docs/experiments/BOT_Experiment_script.py:31-44
  sine / quadratic / growth           docs/notebooks/Experiment_TSP_2023.ipynb cell 2 (f1, g1, f3)
  stoch_vol                           docs/experiments/adaptive_experiment.py:51-54 (glmsv)
  gaussian_log_prob                   the ``*lp`` pattern, e.g. gaussfiltax/nonlinearities.py:51-52
"""
import numpy as np

F32 = np.float32

# fn_id values shared with csrc/models.hpp
DYN_LINEAR, DYN_LORENZ96, DYN_LORENZ63, DYN_MANEUVER_BOT, DYN_SINE, DYN_GROWTH = 0, 1, 2, 3, 4, 5
EMI_LINEAR, EMI_BEARING_RANGE, EMI_QUADRATIC, EMI_STOCH_VOL, EMI_BEARING = 0, 1, 2, 3, 4


class DeviceFunction:
    """A registry function: host-callable, and dispatchable on the device by (kind, fn_id, theta)."""

    def __init__(self, kind, fn_id, in_dim, out_dim, noise_dim, theta, host_fn, name):
        self.kind, self.fn_id = kind, int(fn_id)
        self.in_dim, self.out_dim, self.noise_dim = int(in_dim), int(out_dim), int(noise_dim)
        self.theta = np.ascontiguousarray(theta, dtype=F32).ravel()
        self._host_fn, self.name = host_fn, name

    def __call__(self, x, w, u=None):
        x = np.asarray(x, dtype=F32)
        w = np.broadcast_to(np.asarray(w, dtype=F32), (self.noise_dim,)) if np.ndim(w) == 0 else np.asarray(w, dtype=F32)
        u0 = F32(0.0) if u is None else F32(np.asarray(u, dtype=F32).reshape(-1)[0])
        return np.asarray(self._host_fn(x, w, u0), dtype=F32)

    def __repr__(self):
        return f"DeviceFunction({self.name}, kind={self.kind}, in={self.in_dim}, out={self.out_dim}, noise={self.noise_dim})"


def _linear(kind, fid, M, N, name):
    M = np.atleast_2d(np.asarray(M, dtype=F32))
    N = np.eye(M.shape[0], dtype=F32) if N is None else np.atleast_2d(np.asarray(N, dtype=F32))
    if N.shape[0] != M.shape[0]:
        raise ValueError("noise matrix must have as many rows as the function has outputs")
    fn = DeviceFunction(kind, fid, M.shape[1], M.shape[0], N.shape[1], np.concatenate([M.ravel(), N.ravel()]),
                        lambda x, w, u: M @ x + N @ w, name)
    fn.M, fn.N = M, N
    return fn


def linear_dynamics(A, G=None):
    """f(x, q, u) = A x + G q   (G defaults to the identity)."""
    return _linear("dynamics", DYN_LINEAR, A, G, "linear_dynamics")


def linear_emission(H, D=None):
    """h(x, r, u) = H x + D r   (D defaults to the identity)."""
    return _linear("emission", EMI_LINEAR, H, D, "linear_emission")


def lorenz96(state_dim, alpha=1.0, beta=1.0, gamma=8.0, dt=0.01, mode="matrix_power"):
    """f96 of gaussfiltax/nonlinearities.py:49.  mode='matrix_power' is the intended Lorenz-96
    ((Bx)_i = x_{i+1} - x_{i-2}); mode='as_written' reproduces the element-wise jnp.power of
    :48, which makes B == 0."""
    if mode not in ("matrix_power", "as_written"):
        raise ValueError(mode)
    a, b, g, h = F32(alpha), F32(beta), F32(gamma), F32(dt)
    mp = mode == "matrix_power"

    def host(x, q, u):
        ax = np.roll(x, 1)
        bx = (np.roll(x, -1) - np.roll(x, 2)) if mp else np.zeros_like(x)
        return x + h * (a * ax * bx - b * x + g) + q
    return DeviceFunction("dynamics", DYN_LORENZ96, state_dim, state_dim, state_dim,
                          [alpha, beta, gamma, dt, 1.0 if mp else 0.0], host, f"lorenz96[{mode}]")


def pick_even(state_dim):
    """g96 of gaussfiltax/nonlinearities.py:42-45,50: observe the even-indexed states, m = n/2."""
    m = state_dim // 2
    H = np.zeros((m, state_dim), dtype=F32)
    H[np.arange(m), 2 * np.arange(m)] = 1.0
    return linear_emission(H)


def lorenz63(sigma=10.0, rho=28.0, beta=2.667, dt=0.01):
    s, r, b, h = F32(sigma), F32(rho), F32(beta), F32(dt)

    def host(x, q, u):
        return np.array([x[0] + h * s * (x[1] - x[0]),
                         x[1] + h * (x[0] * r - x[1] - x[0] * x[2]),
                         x[2] + h * (x[0] * x[1] - b * x[2])], dtype=F32) + q
    return DeviceFunction("dynamics", DYN_LORENZ63, 3, 3, 3, [sigma, rho, beta, dt], host, "lorenz63")


def maneuver_bot(dt=0.5, acc=0.5):
    """fManBOT of docs/experiments/BOT_Experiment_script.py:31-42; u in {0, 1, 2} selects
    constant-velocity / left turn / right turn."""
    h, a = F32(dt), F32(acc)
    G = np.array([[0.5, 0], [1, 0], [0, 0.5], [0, 1]], dtype=F32)
    FCV = np.array([[1, h, 0, 0], [0, 1, 0, 0], [0, 0, 1, h], [0, 0, 0, 1]], dtype=F32)

    def fct(x, aa):
        om = F32(0.1) * aa / np.sqrt(x[1] ** 2 + x[3] ** 2)
        sn, cs = np.sin(h * om), np.cos(h * om)
        return np.array([[1, sn / om, 0, -(1 - cs) / om], [0, cs, 0, -sn],
                         [0, (1 - cs) / om, 1, sn / om], [0, sn, 0, cs]], dtype=F32)

    def host(x, q, u):
        M = 0.5 * (u - 1) * (u - 2) * FCV - u * (u - 2) * fct(x, a) + 0.5 * u * (u - 1) * fct(x, -a)
        return M.astype(F32) @ x + G @ q
    return DeviceFunction("dynamics", DYN_MANEUVER_BOT, 4, 4, 2, [dt, acc], host, "maneuver_bot")


def bearing_range():
    """gBOT2 of docs/experiments/BOT_Experiment_script.py:44."""
    return DeviceFunction("emission", EMI_BEARING_RANGE, 4, 2, 2, [],
                          lambda x, r, u: np.array([np.arctan2(x[2], x[0]), np.sqrt(x[0] ** 2 + x[2] ** 2)], dtype=F32) + r,
                          "bearing_range")


def bearing():
    """gBOT of docs/experiments/BOT_Experiment_script.py:43 and docs/tests/test_inference.py:46: the bearing
    arctan2(x[2], x[0]) + r alone (emission_dim = 1)."""
    return DeviceFunction("emission", EMI_BEARING, 4, 1, 1, [],
                          lambda x, r, u: np.reshape(np.arctan2(x[2], x[0]) + r, (1,)).astype(F32), "bearing")


def sine(state_dim, w0=10.0):
    w0_ = F32(w0)
    return DeviceFunction("dynamics", DYN_SINE, state_dim, state_dim, state_dim, [w0],
                          lambda x, q, u: np.sin(w0_ * x) + q, "sine")


def quadratic(state_dim, c=1.0):
    c_ = F32(c)
    return DeviceFunction("emission", EMI_QUADRATIC, state_dim, 1, 1, [c],
                          lambda x, r, u: np.reshape(c_ * np.dot(x, x) + r, (1,)), "quadratic")


def growth():
    return DeviceFunction("dynamics", DYN_GROWTH, 1, 1, 1, [],
                          lambda x, q, u: x / F32(2) + F32(25) * x / (1 + x * x) + u + q, "growth")


def stoch_vol(state_dim, sigma=5.0, beta=0.5, c=0.1):
    s, b, c_ = F32(sigma), F32(beta), F32(c)
    return DeviceFunction("emission", EMI_STOCH_VOL, state_dim, state_dim, state_dim, [sigma, beta, c],
                          lambda x, r, u: u * b * np.exp(x / s) * r + (1 - u) * (c_ * x + r), "stoch_vol")


class GaussianLogProb:
    """``lambda x, y, u: MVN(loc=h(x, r_eval, u), covariance_matrix=R).log_prob(y)``: the
    emission log-density every ``*lp`` function of the reference's scripts has (e.g. g96lp,
    gaussfiltax/nonlinearities.py:51-52).  The bootstrap particle filter evaluates it on the
    device; this object only carries (h, R, r_eval)."""

    def __init__(self, emission_function, covariance, r_eval=None):
        if not isinstance(emission_function, DeviceFunction) or emission_function.kind != "emission":
            raise TypeError("emission_function must be an emission DeviceFunction from this module")
        self.emission_function = emission_function
        self.covariance = np.asarray(covariance, dtype=F32)
        self.r_eval = (np.zeros(emission_function.noise_dim, dtype=F32) if r_eval is None
                       else np.asarray(r_eval, dtype=F32))

    def __call__(self, x, y, u=None):
        mu = self.emission_function(x, self.r_eval, u)
        L = np.linalg.cholesky(self.covariance.astype(np.float64))
        z = np.linalg.solve(L, np.asarray(y, dtype=np.float64) - mu)
        return F32(-0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * len(z) * np.log(2 * np.pi))


def gaussian_log_prob(emission_function, covariance, r_eval=None):
    if getattr(emission_function, "fn_id", None) == EMI_STOCH_VOL:
        raise ValueError("the stochastic-volatility emission has a state-dependent noise scale: use stoch_vol_log_prob "
                         "(the reference's lmsvlp), not a constant-covariance Gaussian")
    return GaussianLogProb(emission_function, covariance, r_eval)


class StochVolLogProb(GaussianLogProb):
    """``lmsvlp`` of docs/experiments/adaptive_experiment.py:55-57: ``MVN(loc=glmsv(x, r0, u),
    covariance_matrix=M R M^T).log_prob(y)`` with ``M = u beta diag(exp(x / sigma)) + (1 - u) I`` -- the emission
    log-density of the stochastic-volatility model, whose noise scale depends on the state."""

    def __call__(self, x, y, u):
        x = np.asarray(x, dtype=np.float64)
        sigma, beta, _ = (float(v) for v in self.emission_function.theta)
        uu = float(np.asarray(u).reshape(-1)[0])
        d = uu * beta * np.exp(x / sigma) + (1.0 - uu)
        mu = self.emission_function(x, self.r_eval, uu)
        L = np.linalg.cholesky((d[:, None] * self.covariance.astype(np.float64)) * d[None, :])
        z = np.linalg.solve(L, np.asarray(y, dtype=np.float64) - mu)
        return F32(-0.5 * z @ z - np.log(np.diag(L)).sum() - 0.5 * len(z) * np.log(2 * np.pi))


def stoch_vol_log_prob(emission_function, covariance, r_eval=None):
    if getattr(emission_function, "fn_id", None) != EMI_STOCH_VOL:
        raise ValueError("stoch_vol_log_prob needs the stoch_vol emission function")
    return StochVolLogProb(emission_function, covariance, r_eval)


FN_USER = 100   # BF_FN_USER of include/bayesfilt.h: a function compiled at run time from source text


class UserFunction(DeviceFunction):
    """A dynamics / emission function OUTSIDE the registry, given as HIP C++ source text and compiled at run time
    (hiprtc) into the scan kernel -- the engine's counterpart of the arbitrary Python callables the reference accepts
    (gaussfiltax/models.py:46-49).  Its Jacobians come from forward-mode dual numbers, which is what the reference's
    ``jacfwd`` computes (gaussfiltax/inference.py:328-329).  ``host_fn`` (optional) is a NumPy twin for use on the host;
    the filters never call it."""

    def __init__(self, kind, source, in_dim, out_dim, noise_dim, theta=(), host_fn=None, name="user"):
        def _no_host(x, w, u):
            raise NotImplementedError("this function exists as device source only (pass host_fn= for a NumPy twin)")
        super().__init__(kind, FN_USER, in_dim, out_dim, noise_dim, theta, host_fn or _no_host, name)
        self.source = str(source)


def user_dynamics(source, state_dim, noise_dim=None, theta=(), host_fn=None, name="user_dynamics"):
    """f(x, q, u) from source.  ``source`` defines::

        template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* theta, T* out)

    writing ``out[0 .. BF_N)``; ``BF_N``, ``BF_DQ``, ``BF_M``, ``BF_DR`` are compile-time constants, ``theta`` is this
    function's parameter vector, and sin cos tan exp log sqrt tanh atan atan2 pow abs work for ``T``.  Example (the
    Lorenz-63 map of docs/experiments/exp_lorentz63.py:37-41)::

        template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
          out[0] = th[3] * th[0] * (x[1] - x[0]) + x[0] + q[0];
          out[1] = th[3] * (x[0] * th[1] - x[1] - x[0] * x[2]) + x[1] + q[1];
          out[2] = th[3] * (x[0] * x[1] - th[2] * x[2]) + x[2] + q[2];
        }
    """
    return UserFunction("dynamics", source, state_dim, state_dim, state_dim if noise_dim is None else noise_dim, theta, host_fn, name)


def user_emission(source, state_dim, emission_dim, noise_dim=None, theta=(), host_fn=None, name="user_emission"):
    """h(x, r, u) from source: ``template <class T> __device__ void emission(const T* x, const T* r, T u, const float*
    theta, T* out)`` writing ``out[0 .. BF_M)`` (see :func:`user_dynamics`)."""
    return UserFunction("emission", source, state_dim, emission_dim, emission_dim if noise_dim is None else noise_dim, theta,
                        host_fn, name)


class UserLogProb:
    """``emission_distribution_log_prob(x, y, u)`` (gaussfiltax/models.py:73-84, inference.py:1348-1349) OUTSIDE the Gaussian
    family, given as HIP C++ source text and compiled at run time into the particle-filter kernel::

        template <class T> __device__ T log_prob(const T* x, const float* y, T u, const float* theta)

    (``BF_N`` / ``BF_M`` are compile-time constants; sin cos sincos atan atan2 exp log sqrt abs fma are the canonical fp32
    arithmetic of the weight path).  ``host_fn(x, y, u)`` (optional) is a NumPy twin for the host."""

    def __init__(self, source, theta=(), host_fn=None):
        self.source = str(source)
        self.theta = np.asarray(theta, dtype=F32).reshape(-1)
        self._host_fn = host_fn

    def __call__(self, x, y, u=None):
        if self._host_fn is None:
            raise NotImplementedError("this log-density exists as device source only (pass host_fn= for a NumPy twin)")
        return self._host_fn(x, y, u)


def user_log_prob(source, theta=(), host_fn=None):
    return UserLogProb(source, theta, host_fn)


# ---------------------------------------------------------------------------------------------
# Python callables written with NumPy operations: recorded once on symbolic arguments and turned into the source the
# user_* constructors take (trace.py).  The reference's call sites pass lambdas (docs/experiments/*.py); with jax.numpy
# replaced by numpy they drop in unchanged.
_TRACED = {}


def _traced(key, build):
    fn = _TRACED.get(key)
    if fn is None:
        if len(_TRACED) > 256:
            _TRACED.clear()
        fn = _TRACED[key] = build()
    return fn


def trace_dynamics(fn, state_dim, noise_dim=None, name="traced_dynamics"):
    """``f(x, q, u)`` as a Python function of NumPy operations -> a device function (see :mod:`bayesianfiltering_amd.trace`)."""
    from . import trace
    dq = state_dim if noise_dim is None else noise_dim
    def build():
        src, out_dim = trace.dynamics_source(fn, state_dim, dq)
        if out_dim != state_dim:
            raise trace.TraceError(f"dynamics_function returned {out_dim} values for a state of dimension {state_dim}")
        return user_dynamics(src, state_dim, dq, host_fn=lambda x, w, u: fn(x, w, np.asarray([u], dtype=F32)), name=name)
    return _traced((fn, "dynamics", state_dim, dq), build)


def trace_emission(fn, state_dim, noise_dim, name="traced_emission"):
    """``h(x, r, u)`` as a Python function of NumPy operations -> a device function; the emission dimension is what it returns."""
    from . import trace
    def build():
        src, out_dim = trace.emission_source(fn, state_dim, noise_dim)
        return user_emission(src, state_dim, out_dim, noise_dim, host_fn=lambda x, w, u: fn(x, w, np.asarray([u], dtype=F32)), name=name)
    return _traced((fn, "emission", state_dim, noise_dim), build)


def trace_log_prob(fn, state_dim, emission_dim):
    """``emission_distribution_log_prob(x, y, u)`` as a Python function of NumPy operations -> :class:`UserLogProb`."""
    from . import trace
    return _traced((fn, "log_prob", state_dim, emission_dim),
                   lambda: user_log_prob(trace.log_prob_source(fn, state_dim, emission_dim), host_fn=fn))


def require_device_function(fn, kind, what, state_dim=None, noise_dim=None):
    """A DeviceFunction as it is; a plain Python callable is recorded (``state_dim`` / ``noise_dim`` from the parameters)."""
    if isinstance(fn, DeviceFunction) and fn.kind == kind:
        return fn
    if callable(fn) and not isinstance(fn, DeviceFunction) and state_dim is not None:
        from . import trace
        try:
            return trace_dynamics(fn, state_dim, noise_dim) if kind == "dynamics" else trace_emission(fn, state_dim, noise_dim)
        except trace.TraceError as e:
            raise TypeError(f"{what}: {e}") from e
    raise TypeError(
        f"{what} must be a {kind} DeviceFunction from bayesianfiltering_amd.nonlinearities -- a registry function, "
        f"nonlinearities.user_{kind}(source, ...), or a Python function of NumPy operations (recorded by "
        f"bayesianfiltering_amd.trace) -- (got {type(fn).__name__}): an arbitrary Python callable cannot run inside the HIP "
        "kernels, and there is no CPU fallback.")


# ---------------------------------------------------------------------------------------------
# Scalar test nonlinearities of gaussfiltax/nonlinearities.py:4-34 (f1..f5 with gradient J and
# Hessian H).  Host-side NumPy helpers: the reference only uses them to exercise second-order
# moment approximations on the CPU, they are not part of the device registry.  The reference's H1
# and H2 read an undefined global `dx`; here the dimension comes from the argument.
class ScalarTestFunction:
    """value / gradient / hessian triple of one scalar-valued test function."""

    def __init__(self, name, value, gradient, hessian):
        self.name, self.value, self.gradient, self.hessian = name, value, gradient, hessian

    def __call__(self, x, *args):
        return self.value(np.asarray(x, dtype=np.float64), *args)

    def __repr__(self):
        return f"<scalar test function {self.name}>"


def _f1():  # (1 + x.x)^(p/2)                                            nonlinearities.py:5-7
    def val(x, p):
        return (1.0 + x @ x) ** (p / 2)

    def grad(x, p):
        x = np.asarray(x, dtype=np.float64)
        return p * (1.0 + x @ x) ** (p / 2 - 1) * x

    def hess(x, p):
        x = np.asarray(x, dtype=np.float64)
        s = 1.0 + x @ x
        return 2 * p * (p / 2 - 1) * s ** (p / 2 - 2) * np.outer(x, x) + p * s ** (p / 2 - 1) * np.eye(x.size)

    return ScalarTestFunction("f1", val, grad, hess)


def _f2():  # sinc of the squared norm: sin(x.x) / x.x                     nonlinearities.py:10-17
    def val(x):
        r = x @ x
        return np.sin(r) / r

    def grad(x):
        x = np.asarray(x, dtype=np.float64)
        r = x @ x
        return 2.0 * (r * np.cos(r) - np.sin(r)) / r ** 2 * x

    def hess(x):
        x = np.asarray(x, dtype=np.float64)
        r = x @ x
        g = (r * np.cos(r) - np.sin(r)) / r ** 2          # d/dr [sin r / r]
        gp = -np.sin(r) / r - 2.0 * (r * np.cos(r) - np.sin(r)) / r ** 3
        return 4.0 * gp * np.outer(x, x) + 2.0 * g * np.eye(x.size)

    return ScalarTestFunction("f2", val, grad, hess)


def _f3():  # x0 sin x1                                                    nonlinearities.py:20-22
    return ScalarTestFunction(
        "f3", lambda x: x[0] * np.sin(x[1]),
        lambda x: np.array([np.sin(x[1]), x[0] * np.cos(x[1])]),
        lambda x: np.array([[0.0, np.cos(x[1])], [np.cos(x[1]), -x[0] * np.sin(x[1])]]))


def _f4():  # x0 + sin x1                                                  nonlinearities.py:25-27
    return ScalarTestFunction(
        "f4", lambda x: x[0] + np.sin(x[1]),
        lambda x: np.array([1.0, np.cos(x[1])]),
        lambda x: np.array([[0.0, 0.0], [0.0, -np.sin(x[1])]]))


def _f5(a=1.0, b=1.0):  # x^T diag(a, b) x / 2                             nonlinearities.py:30-34
    A = np.diag([float(a), float(b)])
    return ScalarTestFunction("f5", lambda x: 0.5 * (x @ A @ x), lambda x: A @ np.asarray(x, dtype=np.float64), lambda x: A)


f1, f2, f3, f4, f5 = _f1(), _f2(), _f3(), _f4(), _f5()
J1, J2, J3, J4, J5 = f1.gradient, f2.gradient, f3.gradient, f4.gradient, f5.gradient
H1, H2, H3, H4, H5 = f1.hessian, f2.hessian, f3.hessian, f4.hessian, f5.hessian
