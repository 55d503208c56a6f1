"""Oracle (test infrastructure): restatement of the JAX PRNG pieces the reference calls.

The reference draws every random number through ``jax.random`` (third-party, NOT under
/root/reference; unpinned: setup.py:8-10 says only 'JAX'; era ~0.4.x):

* ``jr.PRNGKey(0)``                      gaussfiltax/inference.py:367, :1306
* ``jr.split(key, N+1)``                 gaussfiltax/inference.py:1342, :1369
* ``jr.split(key, 2)``, ``jr.choice``    gaussfiltax/utils.py:208-210
* ``MVN(...).sample(seed=key)``          gaussfiltax/models.py:83 (tfp: loc + chol(cov) @ normal)

What is restated here is the published algorithm of those functions for the default
(non-partitionable, "threefry2x32") implementation of JAX 0.4.x:

* Threefry-2x32, 20 rounds (Salmon et al., SC'11; Random123 ``threefry2x32_20``).  Pinned by
  the public Random123 known-answer vectors (tests/test_oracle_rng.py).
* ``threefry_2x32(key, counts)``: counts are split in two halves (padded with one 0 if odd),
  the halves are the two words of each block, outputs are concatenated half-after-half.
* ``split(key, num)``: counts = iota(2*num) -> reshape (num, 2).  Pinned by the 20 keys the reference's own
  notebook run printed (BOTExperiment.ipynb cell 6; tests/golden/reference_notebook_keys.json).
* ``random_bits(key, 32, shape)``: counts = iota(size).
* ``uniform``: mantissa trick ``(bits >> 9) | 0x3f800000`` -> [1,2) - 1, affine to [lo, hi), max(lo, .).
* ``normal``: ``sqrt(2) * erf_inv(uniform(nextafter(-1, 0), 1))`` with XLA's f32 ``erf_inv``
  (Giles' single-precision polynomial, "Approximating the erfinv function", 2012).
* ``choice(key, N, (N,), p=w)``: ``c = cumsum(w); r = c[-1] * (1 - uniform(key, (N,)));
  idx = searchsorted(c, r)`` (side='left').
* ``cumsum`` on the CPU backend lowered to ``lax.associative_scan(add)`` (odd/even recursive
  scan); its fp32 rounding order is restated in :func:`cumsum_assoc` and is the canonical
  order both for this oracle and for the HIP resampler (bit-exact index parity needs ONE
  summation order; JAX itself does not promise one across versions).

Stream parity with one particular JAX build cannot be verified offline ("best
effort"); arithmetic parity given identical random inputs is what the tests require.
"""
import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << _U32(r)) | (x >> _U32(32 - r))).astype(_U32)


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32-20 block function on uint32 arrays (vectorised over x0/x1; the key words may be arrays that
    broadcast against them: one key per row)."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(k0, dtype=_U32)
        k1 = np.asarray(k1, dtype=_U32)
        x0 = np.asarray(x0, dtype=_U32).copy()
        x1 = np.asarray(x1, dtype=_U32).copy()
        ks = (k0, k1, (k0 ^ k1 ^ _U32(0x1BD11BDA)).astype(_U32))
        x0 = (x0 + ks[0]).astype(_U32)
        x1 = (x1 + ks[1]).astype(_U32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(_U32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(_U32)
            x0 = (x0 + ks[(i + 1) % 3]).astype(_U32)
            x1 = (x1 + ks[(i + 2) % 3] + _U32(i + 1)).astype(_U32)
    return x0, x1


def PRNGKey(seed):
    """jax.random.PRNGKey for the threefry impl: [hi32(seed), lo32(seed)]."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=_U32)


def threefry_2x32(key, counts):
    """jax._src.prng.threefry_2x32: hash a flat uint32 count array with ``key``."""
    counts = np.asarray(counts, dtype=_U32).ravel()
    odd = counts.size % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, dtype=_U32)])
    half = counts.size // 2
    o0, o1 = threefry2x32(key[0], key[1], counts[:half], counts[half:])
    out = np.concatenate([o0, o1])
    return out[:-1] if odd else out


def split(key, num=2):
    """jax.random.split (non-partitionable threefry): (num, 2) uint32 keys."""
    return threefry_2x32(key, np.arange(2 * num, dtype=_U32)).reshape(num, 2)


def random_bits(key, size):
    """jax.random.bits(key, (size,), uint32)."""
    return threefry_2x32(key, np.arange(size, dtype=_U32))


def random_bits_keys(keys, size):
    """random_bits for MANY keys at once: keys (N, 2) -> (N, size) uint32, row i = random_bits(keys[i], size)."""
    keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
    counts = np.arange(size, dtype=_U32)
    odd = size % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, dtype=_U32)])
    half = counts.size // 2
    o0, o1 = threefry2x32(keys[:, 0:1], keys[:, 1:2], counts[None, :half], counts[None, half:])
    out = np.concatenate([o0, o1], axis=1)
    return out[:, :-1] if odd else out


def bits_to_uniform(bits, lo=np.float32(0.0), hi=np.float32(1.0)):
    """The float32 mantissa trick of jax.random.uniform, applied to raw uint32 bits."""
    lo = np.float32(lo)
    hi = np.float32(hi)
    fb = ((np.asarray(bits, dtype=_U32) >> _U32(9)) | _U32(0x3F800000)).astype(_U32)
    floats = fb.view(np.float32) - np.float32(1.0)
    return np.maximum(lo, (floats * np.float32(hi - lo) + lo).astype(np.float32))


def uniform(key, size, lo=np.float32(0.0), hi=np.float32(1.0)):
    return bits_to_uniform(random_bits(key, size), lo, hi)


_ERFINV_LT = np.array([2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06,
                       0.00021858087, -0.00125372503, -0.00417768164, 0.246640727,
                       1.50140941], dtype=np.float32)
_ERFINV_GE = np.array([-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844,
                       0.00573950773, -0.0076224613, 0.00943887047, 1.00167406,
                       2.83297682], dtype=np.float32)


def erfinv_f32(x):
    """XLA's float32 erf_inv (Giles' polynomial); Horner in fp32."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (-np.log1p((-x * x).astype(np.float32))).astype(np.float32)
        lt = w < np.float32(5.0)
        wl = (w - np.float32(2.5)).astype(np.float32)
        wg = (np.sqrt(w) - np.float32(3.0)).astype(np.float32)
        ww = np.where(lt, wl, wg).astype(np.float32)
        p = np.where(lt, _ERFINV_LT[0], _ERFINV_GE[0]).astype(np.float32)
        for i in range(1, 9):
            c = np.where(lt, _ERFINV_LT[i], _ERFINV_GE[i]).astype(np.float32)
            p = (c + p * ww).astype(np.float32)
        out = (p * x).astype(np.float32)
        out = np.where(np.abs(x) == np.float32(1.0), np.float32(np.inf) * x, out)
    return out.astype(np.float32)


_NORMAL_LO = np.nextafter(np.float32(-1.0), np.float32(0.0), dtype=np.float32)
_SQRT2 = np.float32(np.sqrt(2))


def bits_to_normal(bits):
    """jax.random.normal's bits -> N(0,1) map (float32)."""
    u = bits_to_uniform(bits, _NORMAL_LO, np.float32(1.0))
    return (_SQRT2 * erfinv_f32(u)).astype(np.float32)


def normal(key, size):
    return bits_to_normal(random_bits(key, size))


def _scan_pow2(x):
    """lax.associative_scan(add) on a power-of-two length: out[2i+1] = S[i], out[0] = x[0],
    out[2i] = S[i-1] + x[2i] with S = scan(x[0::2] + x[1::2])  (== Brent-Kung up/down sweep)."""
    n = x.shape[0]
    if n < 2:
        return x.copy()
    odd = _scan_pow2((x[0::2] + x[1::2]).astype(np.float32))
    out = np.empty(n, dtype=np.float32)
    out[1::2] = odd
    out[0] = x[0]
    out[2::2] = (odd[:-1] + x[2::2]).astype(np.float32)
    return out


def cumsum_assoc(x):
    """fp32 inclusive prefix sum in the rounding order of ``lax.associative_scan(add)``.  Lengths
    that are not a power of two are zero-padded to the next one (zeros on the right change no
    rounding of the kept prefix); this is the ONE canonical order of the oracle and of the HIP
    resampler (a workgroup Brent-Kung scan reproduces it bit for bit)."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    p = 1
    while p < n:
        p *= 2
    buf = np.zeros(p, dtype=np.float32)
    buf[:n] = x
    return _scan_pow2(buf)[:n].copy()


def choice_indices(cdf, u):
    """Inverse-CDF draw of jax.random.choice(p=w): r = c[-1]*(1-u); searchsorted left."""
    cdf = np.asarray(cdf, dtype=np.float32)
    r = (cdf[-1] * (np.float32(1.0) - np.asarray(u, dtype=np.float32))).astype(np.float32)
    return np.searchsorted(cdf, r, side="left").astype(np.int32)


def choice(key, weights):
    """jax.random.choice(key, arange(N), (N,), p=weights) -> int32 indices."""
    n = weights.shape[0]
    cdf = cumsum_assoc(weights)
    return choice_indices(cdf, uniform(key, n))


def mvn_sample(key, loc, chol):
    """tfp MultivariateNormalFullCovariance(loc, cov).sample(seed=key) = loc + chol @ z."""
    z = normal(key, loc.shape[0])
    return (loc + (chol @ z).astype(np.float32)).astype(np.float32)


def normal_canonical(key, size):
    """jax.random.normal(key, (size,)) on the canonical arithmetic of oracle/fp32.py (what the HIP engine computes,
    bit for bit); within 3 ulp of :func:`normal`."""
    from . import fp32
    return fp32.bits_to_normal(random_bits(key, size))
