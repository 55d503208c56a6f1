// Counter-based PRNG pieces shared by host and device code: Threefry-2x32 (20 rounds) and the
// bits -> uniform / normal maps of jax.random for its default threefry implementation, which
// is where every random number of the reference comes from (jr.PRNGKey / split / choice:
// gaussfiltax/inference.py:367,1342,1369, gaussfiltax/utils.py:208-210; MVN.sample:
// gaussfiltax/models.py:83).  Layout of split / random_bits follows JAX 0.4.x's
// non-partitionable threefry: counts are split in two halves (padded with one zero if odd), the
// halves are the two words of each block, outputs are concatenated half after half.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "bf_canon_math.hpp"

namespace bf {

struct U32x2 {
  uint32_t x, y;
};

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

__host__ __device__ __forceinline__ U32x2 threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
#define BF_TF_ROUND(R_) x0 += x1; x1 = rotl32(x1, R_); x1 ^= x0;
#define BF_TF_A BF_TF_ROUND(13) BF_TF_ROUND(15) BF_TF_ROUND(26) BF_TF_ROUND(6)
#define BF_TF_B BF_TF_ROUND(17) BF_TF_ROUND(29) BF_TF_ROUND(16) BF_TF_ROUND(24)
  BF_TF_A x0 += ks[1]; x1 += ks[2] + 1u;
  BF_TF_B x0 += ks[2]; x1 += ks[0] + 2u;
  BF_TF_A x0 += ks[0]; x1 += ks[1] + 3u;
  BF_TF_B x0 += ks[1]; x1 += ks[2] + 4u;
  BF_TF_A x0 += ks[2]; x1 += ks[0] + 5u;
#undef BF_TF_A
#undef BF_TF_B
#undef BF_TF_ROUND
  return U32x2{x0, x1};
}

// element i of jax's threefry_2x32(key, iota(count)): the flat count array is cut in two
// halves of h = ceil(count / 2) entries, block j hashes (j, h + j) (a zero pads an odd count)
__host__ __device__ __forceinline__ uint32_t threefry_bits(uint32_t k0, uint32_t k1, uint32_t i, uint32_t count) {
  const uint32_t h = (count + 1u) >> 1;
  const uint32_t j = i < h ? i : i - h;
  const uint32_t c1 = (h + j < count) ? h + j : 0u;
  const U32x2 o = threefry2x32(k0, k1, j, c1);
  return i < h ? o.x : o.y;
}

// key number i of jax.random.split(key, num): elements 2i and 2i+1 of threefry_2x32(key, iota(2*num))
__host__ __device__ __forceinline__ U32x2 threefry_split(uint32_t k0, uint32_t k1, uint32_t i, uint32_t num) {
  return U32x2{threefry_bits(k0, k1, 2u * i, 2u * num), threefry_bits(k0, k1, 2u * i + 1u, 2u * num)};
}

__host__ __device__ __forceinline__ float bits_to_unit(uint32_t bits) {  // [0, 1)
  union { uint32_t u; float f; } c;
  c.u = (bits >> 9) | 0x3F800000u;
  return c.f - 1.0f;
}

// XLA's float32 erf_inv (Giles' single-precision polynomial) on the canonical arithmetic of bf_canon_math.hpp:
// t = x x and 1 - t rounded on their own (as XLA's log1p(-x x) sees them), w = -log(1 - t) by canon_log, the
// polynomial by fma, an IEEE square root in the tail branch.  Host and device, and the oracle's restatement
// (the test oracle: fp32.erfinv), return the same bits; against the libm-based form the result moves by <= 3 ulp.
__host__ __device__ __forceinline__ float erfinv_f32(float x) {
#pragma clang fp contract(off)
  const float t = x * x;
  const float a = 1.0f - t;
  float w = -canon_log(a);
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = __builtin_fmaf(p, w, 3.43273939e-07f);
    p = __builtin_fmaf(p, w, -3.5233877e-06f);
    p = __builtin_fmaf(p, w, -4.39150654e-06f);
    p = __builtin_fmaf(p, w, 0.00021858087f);
    p = __builtin_fmaf(p, w, -0.00125372503f);
    p = __builtin_fmaf(p, w, -0.00417768164f);
    p = __builtin_fmaf(p, w, 0.246640727f);
    p = __builtin_fmaf(p, w, 1.50140941f);
  } else {
    w = __builtin_sqrtf(w) - 3.0f;  // correctly rounded (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt)
    p = -0.000200214257f;
    p = __builtin_fmaf(p, w, 0.000100950558f);
    p = __builtin_fmaf(p, w, 0.00134934322f);
    p = __builtin_fmaf(p, w, -0.00367342844f);
    p = __builtin_fmaf(p, w, 0.00573950773f);
    p = __builtin_fmaf(p, w, -0.0076224613f);
    p = __builtin_fmaf(p, w, 0.00943887047f);
    p = __builtin_fmaf(p, w, 1.00167406f);
    p = __builtin_fmaf(p, w, 2.83297682f);
  }
  return fabsf(x) == 1.0f ? x * __builtin_inff() : p * x;
}

// jax.random.normal's map: u = max(lo, unit * (1 - lo) + lo), lo = nextafter(-1, 0); sqrt(2) * erf_inv(u)
__host__ __device__ __forceinline__ float bits_to_normal(uint32_t bits) {
#pragma clang fp contract(off)
  const float lo = -0.99999994f;
  float u = __builtin_fmaf(bits_to_unit(bits), 1.0f - lo, lo);  // (1 - lo) = 2.0f in binary32: the product is exact
  u = u > lo ? u : lo;
  return 1.41421356237309515f * erfinv_f32(u);
}

}  // namespace bf
