"""Oracle (test infrastructure): ONE definition of the fp32 arithmetic of the particle filter's weight path.

``north_star`` asks for "bit-exact resampling indices for fixed RNG".  Inside a filter run the ancestor draw
``searchsorted(cumsum(w), c[-1] * (1 - u))`` (gaussfiltax/utils.py:210) sees weights that went through ``erf_inv``
(the normal draws of models.py:83), ``f`` / ``h``, the Gaussian log-density, ``exp`` and the normalisation
(inference.py:1344-1353); one ulp of difference in any of them can move a draw across a CDF step.  The reference leaves
all of this to XLA (``log1p``, ``exp``, fusion and FMA contraction decided by the compiler and the CPU it runs on), so
there is no single bit pattern to match; parity of indices needs the arithmetic DEFINED.  This module defines it as
IEEE-754 binary32 operations in a fixed order -- add, sub, mul, div, sqrt, fused multiply-add, round-to-nearest-even to
integer, and integer bit manipulation -- restated here in NumPy and, independently, in the HIP engine
(``csrc/bf_canon_math.hpp``, ``bf_rng.hpp``, ``bpf_scan.hpp``).  Same inputs, same key => same bits, every step.

* :func:`fma`        -- correctly rounded binary32 fused multiply-add (NumPy has none): exact product in binary64,
                        sum rounded to ODD in binary64 (Boldo-Melquiond), then one rounding to binary32.
* :func:`canon_log`  -- natural logarithm: frexp to [sqrt(1/2), sqrt(2)), degree-8 polynomial of Cephes' ``logf``
                        (S. Moshier; the coefficients of the widely copied ``sse_mathfun`` ``log_ps``), Horner by fma.
* :func:`canon_exp`  -- exponential: k = rint(x log2 e), two-constant Cody-Waite reduction, Cephes' ``expf`` polynomial,
                        scaling by 2^k through the exponent field; flushes to 0 below -86 (no subnormal results), inf above 88.
* :func:`erfinv`     -- XLA's float32 ``erf_inv`` polynomial (Giles 2012, as oracle/threefry.py) with
                        ``w = -log(1 - x x)`` through :func:`canon_log` and the polynomial by fma.
Accuracy (tests/test_oracle_fp32.py): canon_log / canon_exp within 2 ulp of the correctly rounded result, erfinv within
3 ulp of the libm-based form of oracle/threefry.py.
"""
import numpy as np

F32 = np.float32
F64 = np.float64
_U32 = np.uint32
_I32 = np.int32


def f32(x):
    return np.asarray(x, dtype=F32)


def fma(a, b, c):
    """Correctly rounded float32 a * b + c (elementwise, broadcasting)."""
    a64, b64, c64 = (np.asarray(v, dtype=F32).astype(F64) for v in (a, b, c))
    with np.errstate(invalid="ignore", over="ignore"):
        p = a64 * b64                       # exact: 24 + 24 significand bits fit in 53
        s = p + c64                         # rounded to nearest in binary64
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)   # TwoSum: exact error of that addition
        # round to odd: if inexact, take the truncated value and set its last bit
        si = np.atleast_1d(s).view(np.int64).copy()
        e = np.atleast_1d(err)
        sv = np.atleast_1d(s)
        inexact = (e != 0) & np.isfinite(sv) & np.isfinite(e)
        away = inexact & (np.signbit(e) == np.signbit(sv)) & (sv != 0)   # exact value lies beyond s: s is already the truncation
        toward = inexact & ~away                                         # exact value lies between pred(s) and s
        si = np.where(toward, si - 1, si)                                # sign-magnitude: -1 steps towards zero
        si = np.where(inexact, si | 1, si)
        out = si.view(F64).astype(F32)
    return out.reshape(np.shape(s)) if np.ndim(s) else F32(out[0])


def _bits(x):
    return np.asarray(x, dtype=F32).view(_U32) if np.ndim(x) else np.asarray([x], dtype=F32).view(_U32)


_LOG_P = [F32(v) for v in (7.0376836292E-2, -1.1514610310E-1, 1.1676998740E-1, -1.2420140846E-1, 1.4249322787E-1,
                           -1.6668057665E-1, 2.0000714765E-1, -2.4999993993E-1, 3.3333331174E-1)]
_LN2_HI = F32(0.693359375)
_LN2_LO = F32(-2.12194440e-4)
_SQRTHF = F32(0.70710678118654752)


def canon_log(x):
    """log(x) for positive normal float32 x (0 -> -inf, negative / NaN -> NaN, inf -> inf)."""
    x = np.atleast_1d(np.asarray(x, dtype=F32))
    ix = x.view(_U32)
    e = ((ix >> _U32(23)) & _U32(0xFF)).astype(_I32) - _I32(126)                 # frexp exponent: x = m 2^e, m in [0.5, 1)
    m = ((ix & _U32(0x007FFFFF)) | _U32(0x3F000000)).view(F32)
    small = m < _SQRTHF
    e = np.where(small, e - 1, e).astype(_I32)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        f = np.where(small, (m - F32(1.0)) + m, m - F32(1.0)).astype(F32)        # m in [sqrt(1/2), sqrt(2)) - 1 (both forms exact)
        z = (f * f).astype(F32)
        y = np.full(f.shape, _LOG_P[0], dtype=F32)
        for c in _LOG_P[1:]:
            y = fma(y, f, c)
        y = (y * f).astype(F32)
        y = (y * z).astype(F32)
        ef = e.astype(F32)
        y = fma(ef, _LN2_LO, y)
        y = fma(F32(-0.5), z, y)
        r = (f + y).astype(F32)
        r = fma(ef, _LN2_HI, r)
    r = np.where(x == 0, F32(-np.inf), r)
    r = np.where((x < 0) | np.isnan(x), F32(np.nan), r)
    r = np.where(np.isposinf(x), F32(np.inf), r)
    sub = (x > 0) & (x < F32(1.17549435e-38))
    if np.any(sub):                                                              # subnormal inputs: scale by 2^24 first
        xs = np.where(sub, x, F32(1.0)).astype(F32)
        r = np.where(sub, (canon_log((xs * F32(16777216.0)).astype(F32)) - F32(16.635532333438686)).astype(F32), r)
    return r.astype(F32)


_EXP_P = [F32(v) for v in (1.9875691500E-4, 1.3981999507E-3, 8.3334519073E-3, 4.1665795894E-2, 1.6666665459E-1,
                           5.0000001201E-1)]
_LOG2E = F32(1.44269504088896341)


def canon_exp(x):
    """exp(x): 0 for x < -86 (no subnormal results), +inf for x > 88, NaN for NaN."""
    x = np.atleast_1d(np.asarray(x, dtype=F32))
    with np.errstate(invalid="ignore", over="ignore"):
        xc = np.clip(x, F32(-86.0), F32(88.0)).astype(F32)
        k = np.rint((xc * _LOG2E).astype(F32)).astype(F32)                       # round half to even
        r = fma(k, -_LN2_HI, xc)
        r = fma(k, -_LN2_LO, r)
        z = (r * r).astype(F32)
        y = np.full(r.shape, _EXP_P[0], dtype=F32)
        for c in _EXP_P[1:]:
            y = fma(y, r, c)
        y = fma(y, z, r)
        y = (y + F32(1.0)).astype(F32)
        ki = np.where(np.isnan(k), 0, k).astype(_I32)
        # 2^k through the exponent field; k = 128 (x near 88) in two factors
        hi = np.where(ki > 127, ki - 127, 0).astype(_I32)
        s1 = (((ki - hi + 127).astype(_U32)) << _U32(23)).view(F32)
        s2 = (((hi + 127).astype(_U32)) << _U32(23)).view(F32)
        out = ((y * s1).astype(F32) * s2).astype(F32)
    out = np.where(x < F32(-86.0), F32(0.0), out)
    out = np.where(x > F32(88.0), F32(np.inf), out)
    out = np.where(np.isnan(x), F32(np.nan), out)
    return out.astype(F32)


_ERFINV_LT = [F32(v) for v in (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                               -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)]
_ERFINV_GE = [F32(v) for v in (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                               -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)]


def erfinv(x):
    """XLA's float32 erf_inv (Giles) on the canonical log: w = -log(1 - x x); |x| = 1 -> +-inf."""
    x = np.atleast_1d(np.asarray(x, dtype=F32))
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        w = (-canon_log((F32(1.0) - (x * x).astype(F32)).astype(F32))).astype(F32)   # x x and 1 - t rounded on their own, as XLA's log1p(-x x) sees them
        lt = w < F32(5.0)
        ww = np.where(lt, (w - F32(2.5)).astype(F32), (np.sqrt(w).astype(F32) - F32(3.0)).astype(F32)).astype(F32)
        p = np.where(lt, _ERFINV_LT[0], _ERFINV_GE[0]).astype(F32)
        for i in range(1, 9):
            p = fma(p, ww, np.where(lt, _ERFINV_LT[i], _ERFINV_GE[i]).astype(F32))
        out = (p * x).astype(F32)
        out = np.where(np.abs(x) == F32(1.0), (x * F32(np.inf)).astype(F32), out)
    return out.astype(F32)


_NORMAL_LO = np.nextafter(F32(-1.0), F32(0.0), dtype=F32)
_SQRT2 = F32(1.41421356237309515)


def bits_to_normal(bits):
    """jax.random.normal's bits -> N(0, 1) map on the canonical erf_inv."""
    fb = ((np.asarray(bits, dtype=_U32) >> _U32(9)) | _U32(0x3F800000)).astype(_U32)
    unit = (fb.view(F32) - F32(1.0)).astype(F32)
    u = fma(unit, F32(F32(1.0) - _NORMAL_LO), _NORMAL_LO)            # (1 - lo) = 2.0f in binary32: the product is exact
    u = np.maximum(_NORMAL_LO, u).astype(F32)
    return (_SQRT2 * erfinv(u)).astype(F32).reshape(np.shape(bits))


def dot_fma(rows, x):
    """s_i = fma chain over k ascending of rows[i, k] * x[..., k], FIRST TERM A PLAIN PRODUCT (kf_math.hpp: mv):
    rows (R, K), x (..., K) -> (..., R)."""
    rows = np.asarray(rows, dtype=F32)
    x = np.asarray(x, dtype=F32)
    s = (rows[:, 0] * x[..., 0:1]).astype(F32)
    for k in range(1, rows.shape[1]):
        s = fma(rows[:, k], x[..., k:k + 1], s)
    return s.astype(F32)


def lower_matvec_fma(L, z):
    """s_d = fma chain over c = 0..d of L[d, c] * z[..., c] STARTING FROM 0 (bf_rng / bpf_scan: loc + chol z):
    L (D, D) lower triangular, z (..., D) -> (..., D).  (The zero upper part contributes exact zeros.)"""
    L = np.asarray(L, dtype=F32)
    z = np.asarray(z, dtype=F32)
    s = np.zeros(z.shape[:-1] + (L.shape[0],), dtype=F32)
    for c in range(L.shape[1]):
        s = fma(L[:, c], z[..., c:c + 1], s)
    return s.astype(F32)


def cholesky_lower(A):
    """fp32 Cholesky factor by plain loops, every operation rounded on its own (ssm_device.hpp: cholesky_lower)."""
    A = np.asarray(A, dtype=F32)
    n = A.shape[0]
    L = np.zeros((n, n), dtype=F32)
    for j in range(n):
        d = F32(A[j, j])
        for k in range(j):
            d = F32(d - F32(L[j, k] * L[j, k]))
        if not d > 0:
            raise np.linalg.LinAlgError("not positive definite")
        d = F32(np.sqrt(d))
        L[j, j] = d
        for i in range(j + 1, n):
            s = F32(A[i, j])
            for k in range(j):
                s = F32(s - F32(L[i, k] * L[j, k]))
            L[i, j] = F32(s / d)
    return L


def sincos(x):
    """bf_canon_math.hpp: canon_sincos -- Cephes sinf / cosf with every product and sum rounded on its own (NumPy float32
    expressions do exactly that).  Returns (sin, cos); |x| > 8192, NaN, inf: NumPy's functions (outside the canonical range)."""
    x = np.atleast_1d(np.asarray(x, dtype=F32))
    ax = np.abs(x)
    ok = ax <= F32(8192.0)
    axs = np.where(ok, ax, F32(0.0)).astype(F32)
    j = (axs * F32(1.27323954473516)).astype(F32).astype(np.int32)      # truncation
    j = j + (j & 1)
    y = j.astype(F32)
    r = (axs - y * F32(0.78515625)).astype(F32)
    r = (r - y * F32(2.4187564849853515625e-4)).astype(F32)
    r = (r - y * F32(3.77489497744594108e-8)).astype(F32)
    z = (r * r).astype(F32)
    ps = (((F32(-1.9515295891e-4) * z + F32(8.3321608736e-3)) * z - F32(1.6666654611e-1)) * z * r + r).astype(F32)
    pc = (((F32(2.443315711809948e-5) * z - F32(1.388731625493765e-3)) * z + F32(4.166664568298827e-2)) * z * z
          - F32(0.5) * z + F32(1.0)).astype(F32)
    q = (j >> 1) & 3
    s_abs = np.select([q == 0, q == 1, q == 2], [ps, pc, -ps], -pc).astype(F32)
    cs = np.select([q == 0, q == 1, q == 2], [pc, -ps, -pc], ps).astype(F32)
    sn = np.where(x < 0, -s_abs, s_abs).astype(F32)
    with np.errstate(invalid="ignore"):
        sn = np.where(ok, sn, np.sin(x)).astype(F32)
        cs = np.where(ok, cs, np.cos(x)).astype(F32)
    return sn, cs


def atan(x):
    """bf_canon_math.hpp: canon_atan (Cephes atanf), operations rounded one by one."""
    x = np.atleast_1d(np.asarray(x, dtype=F32))
    ax = np.abs(x)
    big, mid = ax > F32(2.414213562373095), ax > F32(0.4142135623730950)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_big = (-(F32(1.0) / ax)).astype(F32)
        t_mid = ((ax - F32(1.0)) / (ax + F32(1.0))).astype(F32)
    t = np.where(big, t_big, np.where(mid, t_mid, ax)).astype(F32)
    y0 = np.where(big, F32(1.5707963267948966), np.where(mid, F32(0.7853981633974483), F32(0.0))).astype(F32)
    z = (t * t).astype(F32)
    p = ((((F32(8.05374449538e-2) * z - F32(1.38776856032e-1)) * z + F32(1.99777106478e-1)) * z - F32(3.33329491539e-1)) * z * t
         + t).astype(F32)
    r = (y0 + p).astype(F32)
    return np.where(x < 0, -r, r).astype(F32)


def atan2(y, x):
    """bf_canon_math.hpp: canon_atan2."""
    y, x = np.broadcast_arrays(np.atleast_1d(np.asarray(y, dtype=F32)), np.atleast_1d(np.asarray(x, dtype=F32)))
    pi, half = F32(3.14159265358979323846), F32(1.5707963267948966)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        z = atan((y / x).astype(F32))
    out = np.where(x < 0, np.where(y < 0, (z - pi).astype(F32), (z + pi).astype(F32)), z).astype(F32)
    out = np.where(y == 0, np.where(x < 0, pi, F32(0.0)), out)
    out = np.where(x == 0, np.where(y > 0, half, np.where(y < 0, -half, F32(0.0))), out)
    with np.errstate(invalid="ignore"):
        out = np.where(np.isnan(x) | np.isnan(y), (x + y).astype(F32), out)
    return out.astype(F32)
