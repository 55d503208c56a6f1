"""The canonical fp32 arithmetic (oracle/fp32.py <-> csrc/bf_canon_math.hpp, bf_rng.hpp): the oracle's exact fused
multiply-add, the accuracy of its log / exp / erf_inv, and BIT equality with the engine's host build of the same
definitions (the device build is compared in tests/test_bpf_gpu.py)."""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest

from oracle import fp32, threefry as otf, gaussfilt_oracle as go, models as om

F32 = np.float32


def _exact_round_f32(v: Fraction) -> np.float32:
    if v == 0:
        return F32(0)
    sign = -1 if v < 0 else 1
    v = abs(v)
    e = v.numerator.bit_length() - v.denominator.bit_length()
    while Fraction(2) ** e > v:
        e -= 1
    while Fraction(2) ** (e + 1) <= v:
        e += 1
    e = max(e, -126)
    scale = Fraction(2) ** (e - 23)
    q = v / scale
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return F32(sign * float(n * scale))


def test_fma_is_correctly_rounded():
    rng = np.random.default_rng(0)
    N = 6000
    a = (rng.normal(size=N) * 10.0 ** rng.integers(-3, 3, N)).astype(F32)
    b = rng.normal(size=N).astype(F32)
    c = (-a.astype(np.float64) * b * (1 + rng.normal(size=N) * 1e-4)).astype(F32)      # heavy cancellation
    c[::3] = rng.normal(size=len(c[::3])).astype(F32)
    # ties of the binary64 intermediate: a * b exactly representable, c half an ulp of the float32 sum away
    a[:50] = F32(1.0) + F32(2.0 ** -12) * np.arange(50, dtype=F32)
    b[:50] = a[:50]
    c[:50] = F32(2.0 ** -25) * (1 + np.arange(50) % 3).astype(F32)
    got = fp32.fma(a, b, c)
    for i in range(N):
        ref = _exact_round_f32(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))
        assert ref == got[i], (a[i], b[i], c[i], got[i], ref)


def _ulps(got, ref64):
    return np.abs(got.astype(np.float64) - ref64) / np.spacing(np.abs(ref64).astype(F32)).astype(np.float64)


def test_canonical_log_exp_erfinv_accuracy():
    rng = np.random.default_rng(1)
    xs = np.exp(rng.uniform(-85, 85, 200000)).astype(F32)
    assert _ulps(fp32.canon_log(xs), np.log(xs.astype(np.float64))).max() < 2.0
    near1 = (1 + rng.uniform(-0.3, 0.42, 200000)).astype(F32)
    assert np.abs(fp32.canon_log(near1) - np.log(near1.astype(np.float64))).max() < 4e-8
    xe = rng.uniform(-85.9, 88, 200000).astype(F32)
    assert _ulps(fp32.canon_exp(xe), np.exp(xe.astype(np.float64))).max() < 2.0
    assert fp32.canon_exp(F32(0.0))[0] == 1.0 and fp32.canon_exp(F32(-87.0))[0] == 0.0 and np.isinf(fp32.canon_exp(F32(89.0))[0])
    assert np.isneginf(fp32.canon_log(F32(0.0))[0]) and np.isnan(fp32.canon_log(F32(-1.0))[0])
    assert _ulps(fp32.canon_log(F32(1e-41)), np.log(np.float64(F32(1e-41))))[0] < 2.0          # subnormal input
    bits = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32)
    z_c, z_l = fp32.bits_to_normal(bits), otf.bits_to_normal(bits)
    assert (np.abs(z_c.astype(np.float64) - z_l) / np.spacing(np.abs(z_l))).max() <= 3.0      # vs the libm-based form


def _host_eval(op, x):
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    x = np.ascontiguousarray(x)
    out = np.empty(x.shape, F32)
    _lib.check(lib.bf_canon_eval_f32(op, x.ctypes.data_as(C.c_void_p), x.size, out.ctypes.data_as(C.c_void_p), 0, None))
    return out


def _same_bits(a, b):
    return np.array_equal(np.asarray(a, F32).view(np.uint32), np.asarray(b, F32).view(np.uint32))


def test_engine_host_build_equals_oracle_bit_for_bit():
    rng = np.random.default_rng(2)
    edge = np.array([0.0, 1.0, 2.0, 0.5, 0.70710677, 0.7071068, 1e-38, 1e-41, 3.4e38, np.inf, -1.0, np.nan], F32)
    xs = np.concatenate([edge, np.exp(rng.uniform(-87, 88, 300000)).astype(F32), (1 + rng.uniform(-0.3, 0.42, 100000)).astype(F32)])
    got, ref = _host_eval(0, xs), fp32.canon_log(xs)
    ok = np.isnan(got) & np.isnan(ref)
    assert _same_bits(got[~ok], ref[~ok])
    xe = np.concatenate([np.array([0.0, -0.0, -86.0, -86.00001, 88.0, 88.00001, -200.0, 200.0, 0.34657359, -0.34657359], F32),
                         rng.uniform(-90, 90, 400000).astype(F32), rng.uniform(-1, 1, 100000).astype(F32)])
    assert _same_bits(_host_eval(1, xe), fp32.canon_exp(xe))
    bits = np.concatenate([np.array([0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xFFFFFE00, 0x1FF], np.uint32),
                           rng.integers(0, 2 ** 32, 500000, dtype=np.uint64).astype(np.uint32)])
    assert _same_bits(_host_eval(2, bits.view(F32)), fp32.bits_to_normal(bits))


def test_library_normal_draw_equals_canonical_oracle():
    """bf_random_normal_f32 (host; the initial component means of inference.py:367) == threefry.normal_canonical."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import inference as inf
    for seed, count in ((0, 30), (7, 1), (123456789, 1001)):
        assert _same_bits(inf._random_normal(bfa.PRNGKey(seed), count), otf.normal_canonical(otf.PRNGKey(seed), count))


def test_canonical_particle_filter_agrees_with_the_libm_oracle():
    """The two arithmetics of oracle.bootstrap_particle_filter restate the same algorithm: identical resampling decisions,
    weights and particles to rounding, on a run short enough that no draw sits within an ulp of a CDF step."""
    n, m, N, T = 8, 4, 128, 5
    R, Q = 0.5 * np.eye(m, dtype=F32), 1e-1 * np.eye(n, dtype=F32)
    po = go.ParamsBPF(8 * np.ones(n, F32), np.eye(n, dtype=F32), om.Lorenz96(n), np.zeros(n, F32), Q, om.PickEven(n),
                      np.zeros(m, F32), R, go.GaussianEmissionLogProb(om.PickEven(n), R))
    ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(1), T)[1]
    key = np.array([0, 11], np.uint32)
    r1, d1 = go.bootstrap_particle_filter(po, ys, N, key=key, debug=True)
    r2, d2 = go.bootstrap_particle_filter(po, ys, N, key=key, debug=True, arith="canonical")
    assert np.array_equal(d1["resampled"], d2["resampled"])
    same = [np.array_equal(d1["ancestors"][t], d2["ancestors"][t]) for t in range(T)]
    t_ok = same.index(False) if False in same else T
    assert t_ok >= 2
    assert np.max(np.abs(r1["weights"][:, :t_ok] - r2["weights"][:, :t_ok])) < 1e-6
    assert np.max(np.abs(r1["particles"][:, :t_ok] - r2["particles"][:, :t_ok])) < 2e-5
    # a full R (forward substitution) and the stochastic-volatility density
    nn = 3
    Rf = (1e-1 * np.eye(nn) + 0.02).astype(F32)
    hn = om.StochVol(nn)
    lp = go.StochVolEmissionLogProb(hn, Rf)
    X = np.random.default_rng(0).normal(size=(64, nn)).astype(F32)
    y = np.array([0.1, -0.2, 0.3], F32)
    for u in (np.zeros(1, F32), np.ones(1, F32)):
        a = lp.logprob_c(X, y, u)
        b = np.array([lp(X[i], y, u) for i in range(64)], F32)
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1)) < 5e-6


def test_engine_host_trig_equals_oracle_bit_for_bit():
    """canon_sincos / canon_atan2 (csrc/bf_canon_math.hpp, host build) against oracle/fp32.py, and their accuracy against
    float64: Cephes single-precision algorithms, < 2 ulp for |x| <= 64 and 3e-7 absolute up to 8192 (sin, cos), <= 3 ulp (atan2)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([np.array([0.0, -0.0, 1e-30, 0.78539816, 0.7853982, 1.5707964, 3.1415927, -3.1415927, 6.2831855, 100.0,
                                  8191.9, 8192.0], F32),
                        (rng.normal(size=300000) * 20).astype(F32), (rng.normal(size=100000) * 1e-3).astype(F32),
                        rng.uniform(-8192, 8192, 100000).astype(F32)])
    sn, cs = fp32.sincos(x)
    assert _same_bits(_host_eval(3, x), sn) and _same_bits(_host_eval(4, x), cs)
    xd = x.astype(np.float64)
    near = np.abs(x) <= 64          # relative accuracy where the three-part reduction is exact enough; absolute beyond
    assert _ulps(sn[near], np.sin(xd[near])).max() < 2.0 and _ulps(cs[near], np.cos(xd[near])).max() < 2.0
    assert np.abs(sn - np.sin(xd)).max() < 3e-7 and np.abs(cs - np.cos(xd)).max() < 3e-7
    y = np.concatenate([np.array([0.0, 0.0, 1.0, -1.0, 0.0, 1.0, -1.0, 1.0, -1.0, 1e-30, 1e30], F32), rng.normal(size=300000).astype(F32),
                        (rng.normal(size=100000) * 1e3).astype(F32)])
    xx = np.concatenate([np.array([1.0, -1.0, 0.0, 0.0, 0.0, 1.0, 1.0, -1.0, -1.0, 1e30, 1e-30], F32), rng.normal(size=300000).astype(F32),
                         rng.normal(size=100000).astype(F32)])
    a = fp32.atan2(y, xx)
    from bayesianfiltering_amd import _lib
    yx, got = np.ascontiguousarray(np.concatenate([y, xx])), np.empty(y.size, F32)
    _lib.check(_lib.load().bf_canon_eval_f32(5, yx.ctypes.data_as(C.c_void_p), y.size, got.ctypes.data_as(C.c_void_p), 0, None))
    assert _same_bits(got, a)
    nz = (y != 0) & (xx != 0)
    assert _ulps(a[nz], np.arctan2(y[nz].astype(np.float64), xx[nz].astype(np.float64))).max() < 3.5
