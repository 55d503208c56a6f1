// The ONE definition of the fp32 transcendental arithmetic on the particle filter's weight path.
//
// `north_star` asks for bit-exact resampling indices for a fixed RNG.  Inside a filter run the ancestor draw
// (gaussfiltax/utils.py:210) sees weights that went through erf_inv (the normal draws of models.py:83), exp and the
// normalisation (inference.py:1344-1353); the reference leaves those to XLA, so there is no single bit pattern to
// match unless the arithmetic is DEFINED.  Here it is: IEEE-754 binary32 add / sub / mul / div / sqrt / fma,
// round-to-nearest-even to integer and integer bit manipulation, in a fixed order, identical on host and device
// (no v_exp_f32 / v_log_f32 / v_rcp_f32, whose results no CPU can restate) -- and restated independently in NumPy
// by the test oracle (its fp32 module).  Contraction is off inside these functions: every fma is written out.
//   canon_log: frexp to [sqrt(1/2), sqrt(2)), Cephes logf polynomial (S. Moshier), Horner by fma; < 1 ulp measured.
//   canon_exp: k = rint(x log2 e), two-constant Cody-Waite reduction, Cephes expf polynomial, 2^k through the
//              exponent field; 0 below -86 (no subnormal results), +inf above 88; < 1 ulp measured.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>

namespace bf {

__host__ __device__ __forceinline__ float canon_bits_f(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
__host__ __device__ __forceinline__ uint32_t canon_f_bits(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}

__host__ __device__ inline float canon_log(float x) {
#pragma clang fp contract(off)
  if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
  if (x == __builtin_inff()) return x;
  float adj = 0.0f;
  if (x < 1.17549435e-38f) {  // subnormal: scale by 2^24 first
    x = x * 16777216.0f;
    adj = 16.635532333438686f;
  }
  const uint32_t ix = canon_f_bits(x);
  int e = (int)((ix >> 23) & 0xFFu) - 126;                                  // x = m 2^e, m in [0.5, 1)
  const float m = canon_bits_f((ix & 0x007FFFFFu) | 0x3F000000u);
  const bool small = m < 0.70710678118654752f;
  e -= small ? 1 : 0;
  const float f = small ? (m - 1.0f) + m : m - 1.0f;                         // in [sqrt(1/2), sqrt(2)) - 1, exact
  const float z = f * f;
  float y = 7.0376836292E-2f;
  y = __builtin_fmaf(y, f, -1.1514610310E-1f);
  y = __builtin_fmaf(y, f, 1.1676998740E-1f);
  y = __builtin_fmaf(y, f, -1.2420140846E-1f);
  y = __builtin_fmaf(y, f, 1.4249322787E-1f);
  y = __builtin_fmaf(y, f, -1.6668057665E-1f);
  y = __builtin_fmaf(y, f, 2.0000714765E-1f);
  y = __builtin_fmaf(y, f, -2.4999993993E-1f);
  y = __builtin_fmaf(y, f, 3.3333331174E-1f);
  y = y * f;
  y = y * z;
  const float ef = (float)e;
  y = __builtin_fmaf(ef, -2.12194440e-4f, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = f + y;
  r = __builtin_fmaf(ef, 0.693359375f, r);
  return adj != 0.0f ? r - adj : r;
}

__host__ __device__ inline float canon_exp(float x) {
#pragma clang fp contract(off)
  if (x != x) return x;
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  const float k = __builtin_rintf(x * 1.44269504088896341f);                 // round half to even (v_rndne_f32)
  float r = __builtin_fmaf(k, -0.693359375f, x);
  r = __builtin_fmaf(k, 2.12194440e-4f, r);
  const float z = r * r;
  float y = 1.9875691500E-4f;
  y = __builtin_fmaf(y, r, 1.3981999507E-3f);
  y = __builtin_fmaf(y, r, 8.3334519073E-3f);
  y = __builtin_fmaf(y, r, 4.1665795894E-2f);
  y = __builtin_fmaf(y, r, 1.6666665459E-1f);
  y = __builtin_fmaf(y, r, 5.0000001201E-1f);
  y = __builtin_fmaf(y, z, r);
  y = y + 1.0f;
  const int ki = (int)k;
  const int hi = ki > 127 ? ki - 127 : 0;                                     // 2^128 in two factors
  const float s1 = canon_bits_f((uint32_t)(ki - hi + 127) << 23);
  const float s2 = canon_bits_f((uint32_t)(hi + 127) << 23);
  return (y * s1) * s2;
}

}  // namespace bf
