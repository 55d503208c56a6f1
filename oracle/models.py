"""Oracle (test infrastructure): NumPy model zoo with analytic Jacobians.

The reference passes Python callables ``f(x,q,u)``, ``h(x,r,u)`` in ``ParamsNLSSM`` /
``ParamsBPF`` (gaussfiltax/models.py:26-84) and differentiates them with ``jacfwd``
(gaussfiltax/inference.py:328-329).  Here each function is an object exposing the value and
the analytic Jacobians w.r.t. the state and the noise; tests check every Jacobian against
central finite differences in fp64.  All arithmetic is float32 like the reference's JAX path
(no ``jax_enable_x64`` anywhere).

Sources of the functions (citations relative to /root/reference):
* linear  ``A x + G q`` / ``H x + D r``     docs/experiments/adaptive_experiment.py:59-64,
                                            docs/experiments/BOT_Experiment_script.py:31-41
* Lorenz-96 ``f96 / g96``                   gaussfiltax/nonlinearities.py:37-52
* Lorenz-63 ``f63``                         docs/experiments/exp_lorentz63.py:37-41,
                                            docs/notebooks/Experiment_TSP_2023.ipynb cell 2
* BOT ``fManBOT / gBOT2``                   docs/experiments/BOT_Experiment_script.py:31-45
* ``sin(10x)+q``, ``c*x.x + r``             docs/notebooks/Experiment_TSP_2023.ipynb cell 2 (f1, g1)
* growth ``x/2 + 25x/(1+x^2) + u + q``      same cell (f3, g3 = 0.8 x + r)
* stochastic volatility ``glmsv``           docs/experiments/adaptive_experiment.py:47-57
"""
import numpy as np

F32 = np.float32


def _a(x):
    return np.asarray(x, dtype=F32)


class Fn:
    """value(x, noise, u), jac_x(x, noise, u), jac_noise(x, noise, u); all float32."""
    out_dim = None
    noise_dim = None

    def __call__(self, x, w, u):
        return self.value(_a(x), _a(w), _a(u))


class Linear(Fn):
    """x -> M x + N w  (dynamics: M=A, N=G; emission: M=H, N=D)."""

    def __init__(self, M, N=None):
        self.M = _a(M)
        self.N = np.eye(self.M.shape[0], dtype=F32) if N is None else _a(N)
        self.out_dim, self.noise_dim = self.M.shape[0], self.N.shape[1]

    def value(self, x, w, u):
        return (self.M @ x + self.N @ w).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic (oracle/fp32.py), batched: X (N, n), W (N, d) -> (N, out).  M x as an fma chain over
        k ascending whose first term is a plain product (kf_math.hpp: mv), N w as an fma chain from 0, then one add."""
        from . import fp32
        return (fp32.dot_fma(self.M, X) + fp32.lower_matvec_fma(self.N, W)).astype(F32)

    def jac_x(self, x, w, u):
        return self.M

    def jac_noise(self, x, w, u):
        return self.N


class Lorenz96(Fn):
    """gaussfiltax/nonlinearities.py:37-49.  mode 'matrix_power': B = A^(n-1) - A^2 (intended,
    (Bx)_i = x_{i+1} - x_{i-2}); mode 'as_written': jnp.power is element-wise => B == 0."""

    def __init__(self, n, alpha=1.0, beta=1.0, gamma=8.0, dt=0.01, mode="matrix_power"):
        self.n, self.alpha, self.beta, self.gamma, self.dt = n, F32(alpha), F32(beta), F32(gamma), F32(dt)
        self.mode = mode
        self.out_dim = self.noise_dim = n

    def _ab(self, x):
        ax = np.roll(x, 1)                      # (A x)_i = x_{i-1}
        if self.mode == "matrix_power":
            bx = np.roll(x, -1) - np.roll(x, 2)  # x_{i+1} - x_{i-2}
        else:
            bx = np.zeros_like(x)
        return ax.astype(F32), bx.astype(F32)

    def value(self, x, w, u):
        ax, bx = self._ab(x)
        return (x + self.dt * (self.alpha * (ax * bx) - self.beta * x + self.gamma) + w).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched (N, n): the expression of :meth:`value` with every operation rounded on its
        own, in the order written -- which is what NumPy does with float32 operands."""
        X = np.asarray(X, dtype=F32)
        ax = np.roll(X, 1, axis=-1)
        bx = (np.roll(X, -1, axis=-1) - np.roll(X, 2, axis=-1)).astype(F32) if self.mode == "matrix_power" else np.zeros_like(X)
        return (X + self.dt * (self.alpha * (ax * bx) - self.beta * X + self.gamma) + np.asarray(W, dtype=F32)).astype(F32)

    def jac_x(self, x, w, u):
        n = self.n
        ax, bx = self._ab(x)
        J = np.zeros((n, n), dtype=F32)
        for i in range(n):
            J[i, i] += F32(1.0) - self.dt * self.beta
            if self.mode == "matrix_power":
                J[i, (i - 1) % n] += self.dt * self.alpha * bx[i]
                J[i, (i + 1) % n] += self.dt * self.alpha * ax[i]
                J[i, (i - 2) % n] += -self.dt * self.alpha * ax[i]
        return J

    def jac_noise(self, x, w, u):
        return np.eye(self.n, dtype=F32)


class PickEven(Linear):
    """g96: H[row, 2*row] = 1, m = n/2 (gaussfiltax/nonlinearities.py:42-45,50)."""

    def __init__(self, n):
        m = n // 2
        H = np.zeros((m, n), dtype=F32)
        H[np.arange(m), 2 * np.arange(m)] = 1.0
        super().__init__(H, np.eye(m, dtype=F32))


class Lorenz63(Fn):
    """lorentz_63(x) + q  (docs/experiments/exp_lorentz63.py:37-41)."""

    def __init__(self, sigma=10.0, rho=28.0, beta=2.667, dt=0.01):
        self.s, self.r, self.b, self.dt = F32(sigma), F32(rho), F32(beta), F32(dt)
        self.out_dim = self.noise_dim = 3

    def value(self, x, w, u):
        dt = self.dt
        dx = dt * self.s * (x[1] - x[0])
        dy = dt * (x[0] * self.r - x[1] - x[0] * x[2])
        dz = dt * (x[0] * x[1] - self.b * x[2])
        return (np.array([dx + x[0], dy + x[1], dz + x[2]], dtype=F32) + w).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched (N, 3): :meth:`value` column-wise (every operation rounded on its own)."""
        X = np.asarray(X, dtype=F32)
        x0, x1, x2 = X[:, 0], X[:, 1], X[:, 2]
        dt = self.dt
        dx = dt * self.s * (x1 - x0)
        dy = dt * (x0 * self.r - x1 - x0 * x2)
        dz = dt * (x0 * x1 - self.b * x2)
        return (np.stack([dx + x0, dy + x1, dz + x2], axis=1).astype(F32) + np.asarray(W, dtype=F32)).astype(F32)

    def jac_x(self, x, w, u):
        dt, s, r, b = self.dt, self.s, self.r, self.b
        return np.array([[1 - dt * s, dt * s, 0],
                         [dt * (r - x[2]), 1 - dt, -dt * x[0]],
                         [dt * x[1], dt * x[0], 1 - dt * b]], dtype=F32)

    def jac_noise(self, x, w, u):
        return np.eye(3, dtype=F32)


class ManeuverBOT(Fn):
    """fManBOT: (0.5(u-1)(u-2) FCV - u(u-2) FCT(x,acc) + 0.5u(u-1) FCT(x,-acc)) @ x + G q
    (docs/experiments/BOT_Experiment_script.py:31-42); u scalar in {0,1,2}."""

    def __init__(self, dt=0.5, acc=0.5):
        self.dt, self.acc = F32(dt), F32(acc)
        self.G = np.array([[0.5, 0], [1, 0], [0, 0.5], [0, 1]], dtype=F32)
        self.out_dim, self.noise_dim = 4, 2

    def _mats(self, x, a, dtype=F32):
        """FCT(x,a) and d FCT / d x1, d x3 (x1, x3 are the velocities)."""
        dt = dtype(self.dt)
        s2 = x[1] * x[1] + x[3] * x[3]
        nrm = np.sqrt(s2)
        om = dtype(0.1) * dtype(a) / nrm
        sn, cs = np.sin(dt * om), np.cos(dt * om)
        F = np.array([[1, sn / om, 0, -(1 - cs) / om],
                      [0, cs, 0, -sn],
                      [0, (1 - cs) / om, 1, sn / om],
                      [0, sn, 0, cs]], dtype=dtype)
        # derivative of entries w.r.t. omega
        dsn_om = (dt * cs * om - sn) / (om * om)          # d/dom (sn/om)
        dcs_om = (dt * sn * om - (1 - cs)) / (om * om)    # d/dom ((1-cs)/om)
        dF = np.array([[0, dsn_om, 0, -dcs_om],
                       [0, -dt * sn, 0, -dt * cs],
                       [0, dcs_om, 0, dsn_om],
                       [0, dt * cs, 0, -dt * sn]], dtype=dtype)
        dom = np.array([0, -om * x[1] / s2, 0, -om * x[3] / s2], dtype=dtype)  # d om / d x
        return F, dF, dom

    def _coef(self, u):
        u = F32(np.asarray(u).reshape(-1)[0])
        return F32(0.5) * (u - 1) * (u - 2), -u * (u - 2), F32(0.5) * u * (u - 1)

    def _fcv(self, dtype=F32):
        dt = dtype(self.dt)
        return np.array([[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1]], dtype=dtype)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched (N, 4), (N, 2): the operation sequence of csrc/ssm_device.hpp (dyn_value_t,
        DYN_MANEUVER_BOT): the mixing matrix entry by entry (products and sums rounded one by one, sin / cos by
        fp32.sincos), M x as a product followed by an fma chain (kf_math.hpp: mv), G q as an fma chain from 0."""
        from . import fp32
        X, W = np.asarray(X, dtype=F32), np.asarray(W, dtype=F32)
        u0 = F32(np.asarray(u).reshape(-1)[0])
        dt, acc = self.dt, self.acc
        c0 = F32(F32(F32(0.5) * F32(u0 - F32(1))) * F32(u0 - F32(2)))
        c1 = F32(F32(-u0) * F32(u0 - F32(2)))
        c2 = F32(F32(F32(0.5) * u0) * F32(u0 - F32(1)))
        x1, x3 = X[:, 1], X[:, 3]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            nrm = np.sqrt((x1 * x1 + x3 * x3).astype(F32)).astype(F32)
            sn0, cs0 = fp32.sincos((dt * (F32(F32(0.1) * acc) / nrm).astype(F32)).astype(F32))
            N = X.shape[0]
            Mx = np.zeros((N, 16), dtype=F32)
            for i, v in ((0, c0), (1, F32(c0 * dt)), (5, c0), (10, c0), (11, F32(c0 * dt)), (15, c0)):
                Mx[:, i] = v
            for sgn, cc in ((0, c1), (1, c2)):
                a = acc if sgn == 0 else F32(-acc)
                om = (F32(F32(0.1) * a) / nrm).astype(F32)
                sn = sn0 if sgn == 0 else (-sn0).astype(F32)
                cs = cs0
                so = (sn / om).astype(F32)
                co = ((F32(1.0) - cs).astype(F32) / om).astype(F32)
                one, zero = np.ones(N, F32), np.zeros(N, F32)
                Fm = [one, so, zero, -co, zero, cs, zero, -sn, zero, co, one, so, zero, sn, zero, cs]
                for i in range(16):
                    Mx[:, i] = (Mx[:, i] + (cc * Fm[i]).astype(F32)).astype(F32)
            out = np.empty((N, 4), dtype=F32)
            for i in range(4):
                s = (Mx[:, 4 * i] * X[:, 0]).astype(F32)
                for k in range(1, 4):
                    s = fp32.fma(Mx[:, 4 * i + k], X[:, k], s)
                g = np.zeros(N, dtype=F32)
                for k in range(2):
                    g = fp32.fma(np.full(N, self.G[i, k], F32), W[:, k], g)
                out[:, i] = (s + g).astype(F32)
        return out

    def value(self, x, w, u):
        c0, c1, c2 = self._coef(u)
        Fp, _, _ = self._mats(x, self.acc)
        Fm, _, _ = self._mats(x, -self.acc)
        M = (c0 * self._fcv() + c1 * Fp + c2 * Fm).astype(F32)
        return (M @ x + self.G @ w).astype(F32)

    def jac_x(self, x, w, u):
        c0, c1, c2 = self._coef(u)
        Fp, dFp, domp = self._mats(x, self.acc)
        Fm, dFm, domm = self._mats(x, -self.acc)
        M = c0 * self._fcv() + c1 * Fp + c2 * Fm
        J = M + c1 * np.outer(dFp @ x, domp) + c2 * np.outer(dFm @ x, domm)
        return J.astype(F32)

    def jac_noise(self, x, w, u):
        return self.G


class BearingRange(Fn):
    """gBOT2: [atan2(x2, x0), sqrt(x0^2 + x2^2)] + r  (BOT_Experiment_script.py:44)."""

    def __init__(self):
        self.out_dim = self.noise_dim = 2

    def value(self, x, w, u):
        return (np.array([np.arctan2(x[2], x[0]), np.sqrt(x[0] * x[0] + x[2] * x[2])], dtype=F32) + w).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched: fp32.atan2, an IEEE square root of the sum of two rounded squares, then + r."""
        from . import fp32
        X, W = np.asarray(X, dtype=F32), np.asarray(W, dtype=F32)
        x0, x2 = X[:, 0], X[:, 2]
        with np.errstate(invalid="ignore", over="ignore"):
            b = fp32.atan2(x2, x0)
            r = np.sqrt(((x0 * x0).astype(F32) + (x2 * x2).astype(F32)).astype(F32)).astype(F32)
        return (np.stack([b, r], axis=1).astype(F32) + W).astype(F32)

    def jac_x(self, x, w, u):
        d2 = x[0] * x[0] + x[2] * x[2]
        d = np.sqrt(d2)
        return np.array([[-x[2] / d2, 0, x[0] / d2, 0], [x[0] / d, 0, x[2] / d, 0]], dtype=F32)

    def jac_noise(self, x, w, u):
        return np.eye(2, dtype=F32)


class Bearing(Fn):
    """gBOT: atan2(x2, x0) + r  (BOT_Experiment_script.py:43, docs/tests/test_inference.py:46)."""

    def __init__(self):
        self.out_dim = self.noise_dim = 1

    def value(self, x, w, u):
        return (np.array([np.arctan2(x[2], x[0])], dtype=F32) + w).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched: fp32.atan2 + r."""
        from . import fp32
        X, W = np.asarray(X, dtype=F32), np.asarray(W, dtype=F32)
        return (fp32.atan2(X[:, 2], X[:, 0]).reshape(-1, 1) + W).astype(F32)

    def jac_x(self, x, w, u):
        d2 = x[0] * x[0] + x[2] * x[2]
        return np.array([[-x[2] / d2, 0, x[0] / d2, 0]], dtype=F32)

    def jac_noise(self, x, w, u):
        return np.eye(1, dtype=F32)


class Sine(Fn):
    """f1: sin(w0 * x) + q element-wise (Experiment_TSP_2023.ipynb cell 2, w0 = 10)."""

    def __init__(self, n, w0=10.0):
        self.n, self.w0 = n, F32(w0)
        self.out_dim = self.noise_dim = n

    def value(self, x, w, u):
        return (np.sin(self.w0 * x) + w).astype(F32)

    def jac_x(self, x, w, u):
        return np.diag((self.w0 * np.cos(self.w0 * x)).astype(F32))

    def jac_noise(self, x, w, u):
        return np.eye(self.n, dtype=F32)


class Quadratic(Fn):
    """g1: c * dot(x, x) + r, scalar emission (Experiment_TSP_2023.ipynb cell 2)."""

    def __init__(self, n, c=1.0):
        self.n, self.c = n, F32(c)
        self.out_dim = self.noise_dim = 1

    def value(self, x, w, u):
        return (self.c * np.dot(x, x) + w).astype(F32).reshape(1)

    def jac_x(self, x, w, u):
        return (F32(2.0) * self.c * x).astype(F32).reshape(1, self.n)

    def jac_noise(self, x, w, u):
        return np.eye(1, dtype=F32)


class Growth(Fn):
    """f3: x/2 + 25 x/(1+x^2) + u + q, scalar (Experiment_TSP_2023.ipynb cell 2)."""

    def __init__(self):
        self.out_dim = self.noise_dim = 1

    def value(self, x, w, u):
        uu = F32(np.asarray(u).reshape(-1)[0])
        return (x / F32(2.0) + F32(25.0) * x / (1 + x * x) + uu + w).astype(F32)

    def jac_x(self, x, w, u):
        d = 1 + x * x
        return (F32(0.5) + F32(25.0) * (1 - x * x) / (d * d)).astype(F32).reshape(1, 1)

    def jac_noise(self, x, w, u):
        return np.eye(1, dtype=F32)


class StochVol(Fn):
    """glmsv: u*beta*exp(x/sigma)*r + (1-u)*(H0 x + r), H0 = c*eye
    (docs/experiments/adaptive_experiment.py:51-54)."""

    def __init__(self, n, sigma=5.0, beta=0.5, c=0.1):
        self.n, self.sigma, self.beta, self.c = n, F32(sigma), F32(beta), F32(c)
        self.out_dim = self.noise_dim = n

    def _u(self, u):
        return F32(np.asarray(u).reshape(-1)[0])

    def value(self, x, w, u):
        uu = self._u(u)
        return (uu * self.beta * np.exp(x / self.sigma) * w + (1 - uu) * (self.c * x + w)).astype(F32)

    def value_c(self, X, W, u):
        """Canonical fp32 arithmetic, batched: :meth:`value` with exp by fp32.canon_exp."""
        from . import fp32
        uu = self._u(u)
        X, W = np.asarray(X, dtype=F32), np.asarray(W, dtype=F32)
        e = fp32.canon_exp((X / self.sigma).astype(F32)).reshape(X.shape)
        return (uu * self.beta * e * W + (F32(1) - uu) * (self.c * X + W)).astype(F32)

    def scale_c(self, X, u):
        """Diagonal of the noise Jacobian M(x, u) = u beta exp(x / sigma) + (1 - u), canonical arithmetic."""
        from . import fp32
        uu = self._u(u)
        X = np.asarray(X, dtype=F32)
        e = fp32.canon_exp((X / self.sigma).astype(F32)).reshape(X.shape)
        return (uu * self.beta * e + (F32(1) - uu)).astype(F32)

    def jac_x(self, x, w, u):
        uu = self._u(u)
        d = uu * self.beta * np.exp(x / self.sigma) * w / self.sigma + (1 - uu) * self.c
        return np.diag(d.astype(F32))

    def jac_noise(self, x, w, u):
        uu = self._u(u)
        return np.diag((uu * self.beta * np.exp(x / self.sigma) + (1 - uu)).astype(F32))


def finite_difference_jacobians(fn, x, w, u, eps=1e-6):
    """Central differences in fp64 of a float64 re-evaluation; used by tests only."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)

    def val(xx, ww):
        # evaluate in float64 by temporarily bypassing the float32 casts
        return _eval64(fn, xx, ww, u)

    Jx = np.zeros((fn.out_dim, x.size))
    Jw = np.zeros((fn.out_dim, w.size))
    for i in range(x.size):
        d = np.zeros_like(x); d[i] = eps
        Jx[:, i] = (val(x + d, w) - val(x - d, w)) / (2 * eps)
    for i in range(w.size):
        d = np.zeros_like(w); d[i] = eps
        Jw[:, i] = (val(x, w + d) - val(x, w - d)) / (2 * eps)
    return Jx, Jw


def _eval64(fn, x, w, u):
    """float64 evaluation of the same formulas (independent of the float32 code paths)."""
    u0 = float(np.asarray(u).reshape(-1)[0]) if np.size(u) else 0.0
    if isinstance(fn, Linear):
        return fn.M.astype(np.float64) @ x + fn.N.astype(np.float64) @ w
    if isinstance(fn, Lorenz96):
        ax = np.roll(x, 1)
        bx = np.roll(x, -1) - np.roll(x, 2) if fn.mode == "matrix_power" else 0.0 * x
        return x + float(fn.dt) * (float(fn.alpha) * ax * bx - float(fn.beta) * x + float(fn.gamma)) + w
    if isinstance(fn, Lorenz63):
        dt, s, r, b = map(float, (fn.dt, fn.s, fn.r, fn.b))
        return np.array([x[0] + dt * s * (x[1] - x[0]),
                         x[1] + dt * (x[0] * r - x[1] - x[0] * x[2]),
                         x[2] + dt * (x[0] * x[1] - b * x[2])]) + w
    if isinstance(fn, ManeuverBOT):
        c0, c1, c2 = 0.5 * (u0 - 1) * (u0 - 2), -u0 * (u0 - 2), 0.5 * u0 * (u0 - 1)
        Fp = fn._mats(x, float(fn.acc), np.float64)[0]
        Fm = fn._mats(x, -float(fn.acc), np.float64)[0]
        return (c0 * fn._fcv(np.float64) + c1 * Fp + c2 * Fm) @ x + fn.G.astype(np.float64) @ w
    if isinstance(fn, BearingRange):
        return np.array([np.arctan2(x[2], x[0]), np.hypot(x[0], x[2])]) + w
    if isinstance(fn, Bearing):
        return np.array([np.arctan2(x[2], x[0])]) + w
    if isinstance(fn, Sine):
        return np.sin(float(fn.w0) * x) + w
    if isinstance(fn, Quadratic):
        return np.array([float(fn.c) * x @ x]) + w
    if isinstance(fn, Growth):
        return x / 2 + 25 * x / (1 + x * x) + u0 + w
    if isinstance(fn, StochVol):
        return u0 * float(fn.beta) * np.exp(x / float(fn.sigma)) * w + (1 - u0) * (float(fn.c) * x + w)
    raise TypeError(type(fn))
