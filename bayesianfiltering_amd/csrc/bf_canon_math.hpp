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
#ifndef BF_JIT
#include <cmath>
#endif

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

// canon_log for a POSITIVE, NORMAL, FINITE argument: the arithmetic of canon_log without its special cases (call sites that
// can prove the precondition -- erf_inv's 1 - x^2 with |x| < 1 -- skip a dozen compares, selects and their branches).
__host__ __device__ __forceinline__ float canon_log_core(float x) {
#pragma clang fp contract(off)
#if defined(BF_BPF_HW_ARITH) && defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_logf(x) * 0.693147180559945309f;   // bf_set_option "bpf_arith" = 1: v_log_f32 (log2), 1 ulp
#endif
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
  const float r = f + y;
  return __builtin_fmaf(ef, 0.693359375f, r);
}

__host__ __device__ inline float canon_log(float x) {
#pragma clang fp contract(off)
  if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
  if (x == __builtin_inff()) return x;
  if (x < 1.17549435e-38f) return canon_log_core(x * 16777216.0f) - 16.635532333438686f;  // subnormal: scale by 2^24 first
  return canon_log_core(x);
}

__host__ __device__ inline float canon_exp(float x) {
#pragma clang fp contract(off)
#if defined(BF_BPF_HW_ARITH) && defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);   // bf_set_option "bpf_arith" = 1: v_exp_f32 (2^x), 1 ulp
#endif
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

// sin and cos of one argument: Cephes sinf / cosf (S. Moshier) -- octant j = trunc(|x| 4 / pi) made even, three-part
// Cody-Waite reduction |x| - j pi/4, the degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4] -- with every product and
// sum rounded on its own, in the order written (no fused multiply-adds: the test oracle restates it with NumPy float32
// expressions).  < 2 ulp for |x| <= 64, 3e-7 absolute up to |x| = 8192, the range the reduction constants cover; beyond it
// (and for NaN / inf) libm.
__host__ __device__ inline void canon_sincos(float x, float* sn, float* cs) {
#pragma clang fp contract(off)
  const float ax = __builtin_fabsf(x);
  if (!(ax <= 8192.0f)) {
    *sn = __builtin_sinf(x);
    *cs = __builtin_cosf(x);
    return;
  }
  int j = (int)(ax * 1.27323954473516f);
  j += j & 1;
  const float y = (float)j;
  float r = ax - y * 0.78515625f;
  r = r - y * 2.4187564849853515625e-4f;
  r = r - y * 3.77489497744594108e-8f;
  const float z = r * r;
  const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = (j >> 1) & 3;
  const float s_abs = q == 0 ? ps : q == 1 ? pc : q == 2 ? -ps : -pc;
  *cs = q == 0 ? pc : q == 1 ? -ps : q == 2 ? -pc : ps;
  *sn = x < 0.0f ? -s_abs : s_abs;
}
__host__ __device__ inline float canon_sin(float x) {
  float s, c;
  canon_sincos(x, &s, &c);
  return s;
}

// atan2(y, x): Cephes atanf (two range reductions at tan(pi/8), tan(3 pi/8), degree-4 polynomial in t^2) of the IEEE
// quotient y / x, plus the quadrant's multiple of pi; products and sums rounded one by one.  Zeros: atan2(0, x) = 0 or pi
// by the sign of x, atan2(y, 0) = +- pi/2 (the sign of a zero argument is not looked at).
__host__ __device__ inline float canon_atan(float x) {
#pragma clang fp contract(off)
  const float ax = __builtin_fabsf(x);
  float y0, t;
  if (ax > 2.414213562373095f) {
    y0 = 1.5707963267948966f;
    t = -(1.0f / ax);
  } else if (ax > 0.4142135623730950f) {
    y0 = 0.7853981633974483f;
    t = (ax - 1.0f) / (ax + 1.0f);
  } else {
    y0 = 0.0f;
    t = ax;
  }
  const float z = t * t;
  const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
  const float r = y0 + p;
  return x < 0.0f ? -r : r;
}
__host__ __device__ inline float canon_atan2(float y, float x) {
#pragma clang fp contract(off)
  if (x != x || y != y) return x + y;
  if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : y < 0.0f ? -1.5707963267948966f : 0.0f;
  if (y == 0.0f) return x < 0.0f ? 3.14159265358979323846f : 0.0f;
  const float z = canon_atan(y / x);
  if (x < 0.0f) return y < 0.0f ? z - 3.14159265358979323846f : z + 3.14159265358979323846f;
  return z;
}

}  // namespace bf
