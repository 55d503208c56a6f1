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

// erfinv_f32 for |x| < 1 with 1 - x^2 a normal number (every argument jax.random.normal's map produces: |u| <= 0.99999994):
// the same operations, bit for bit, minus the special cases -- the log without its argument checks, the central polynomial
// evaluated for every lane and the tail (|x| > 0.9966, 3 draws in 1000) behind ONE rarely taken branch instead of a
// two-sided one, no |x| = 1 select.
__host__ __device__ __forceinline__ float erfinv_open_f32(float x) {
#pragma clang fp contract(off)
  const float t = x * x;
  const float a = 1.0f - t;
  const float w0 = -canon_log_core(a);
  float w = w0 - 2.5f;
  float p = 2.81022636e-08f;
  p = __builtin_fmaf(p, w, 3.43273939e-07f);
  p = __builtin_fmaf(p, w, -3.5233877e-06f);
  p = __builtin_fmaf(p, w, -4.39150654e-06f);
  p = __builtin_fmaf(p, w, 0.00021858087f);
  p = __builtin_fmaf(p, w, -0.00125372503f);
  p = __builtin_fmaf(p, w, -0.00417768164f);
  p = __builtin_fmaf(p, w, 0.246640727f);
  p = __builtin_fmaf(p, w, 1.50140941f);
  if (__builtin_expect(!(w0 < 5.0f), 0)) {
    w = __builtin_sqrtf(w0) - 3.0f;
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
  return p * x;
}

// jax.random.normal's map: u = max(lo, unit * (1 - lo) + lo), lo = nextafter(-1, 0); sqrt(2) * erf_inv(u)
__host__ __device__ __forceinline__ float bits_to_normal(uint32_t bits) {
#pragma clang fp contract(off)
  const float lo = -0.99999994f;
  float u = __builtin_fmaf(bits_to_unit(bits), 1.0f - lo, lo);  // (1 - lo) = 2.0f in binary32: the product is exact
  u = u > lo ? u : lo;   // |u| <= 0.99999994: 1 - u^2 >= 1.19e-7, a normal number
  return 1.41421356237309515f * erfinv_open_f32(u);
}

// The tail branch of erfinv_open_f32 alone (w0 = -log(1 - x^2) >= 5), for callers that evaluate the central branch elsewhere.
__host__ __device__ __forceinline__ float erfinv_tail_poly(float w0) {
#pragma clang fp contract(off)
  const float w = __builtin_sqrtf(w0) - 3.0f;
  float p = -0.000200214257f;
  p = __builtin_fmaf(p, w, 0.000100950558f);
  p = __builtin_fmaf(p, w, 0.00134934322f);
  p = __builtin_fmaf(p, w, -0.00367342844f);
  p = __builtin_fmaf(p, w, 0.00573950773f);
  p = __builtin_fmaf(p, w, -0.0076224613f);
  p = __builtin_fmaf(p, w, 0.00943887047f);
  p = __builtin_fmaf(p, w, 1.00167406f);
  p = __builtin_fmaf(p, w, 2.83297682f);
  return p;
}

#ifdef __HIPCC__
// threefry2x32(k0, k1, C0, C1) and bits_to_normal of both output words as ONE hand-scheduled gfx950 instruction sequence:
// the operations of the C++ functions above, one for one and in their order (the bits are the same: tests/test_bpf_gpu.py,
// the compile-time instance against the run-time one and both against the oracle), written out because the compiler, given
// sixteen of these chains per particle and four particles per thread inside a 128-register budget, hoists their ~40 literal
// constants into registers, interleaves the chains and pays with hundreds of spills.  As an opaque block it needs 12 vector
// registers, holds every constant as an instruction literal and keeps its two normals' polynomial chains interleaved (two
// independent chains per wave: a dependent gfx950 VALU instruction issues 8 cycles after its producer).
//   rounds:   x0 += x1; x1 = rotl(x1, R) [v_alignbit_b32 by 32 - R]; x1 ^= x0;   key injection: x0 += ks[a]; x1 += ks[b] + i
//   normal:   unit = bits >> 9 | 1.0f, - 1; u = max(fma(unit, 2, lo), lo); a = 1 - u u; canon_log_core(a); w = -2.5 - log;
//             central erf_inv polynomial p(w)  ->  outputs p, log(a) and u; the caller takes the rare tail branch and the
//             final products in C++ (erfinv_tail_poly).
#define BF_TFR(R_) "v_add_u32 %[x0], %[x0], %[x1]\n\tv_alignbit_b32 %[x1], %[x1], %[x1], " #R_ "\n\tv_xor_b32 %[x1], %[x1], %[x0]\n\t"
#define BF_TFA BF_TFR(19) BF_TFR(17) BF_TFR(6) BF_TFR(26)
#define BF_TFB BF_TFR(15) BF_TFR(3) BF_TFR(16) BF_TFR(8)
#define BF_TFI(KA_, KB_, I_) "v_add_u32 %[x0], %[x0], %[" #KA_ "]\n\tv_add3_u32 %[x1], %[x1], %[" #KB_ "], " #I_ "\n\t"
// bits (register X_, dies) -> u (U_), a = 1 - u^2 (R_), m - 1 or 2 m - 1 (F_), unbiased exponent as float (Z_); Y_ scratch.
// (m - 1) + (small ? m : 0): m - 1 lies in [-0.5, -2^-24], so adding +0 is the identity on its bits, as adding -0 is in the
// C++ form's `small ? (m - 1) + m : m - 1`; the select cannot take a literal next to its implicit vcc read.)
#define BF_NRM_HEAD(X_, U_, R_, F_, Y_, Z_)                                  \
  "v_lshrrev_b32 %[" #U_ "], 9, %[" #X_ "]\n\t"                              \
  "v_or_b32 %[" #U_ "], 1.0, %[" #U_ "]\n\t"                                 \
  "v_add_f32 %[" #U_ "], -1.0, %[" #U_ "]\n\t"                               \
  "v_fmaak_f32 %[" #U_ "], 2.0, %[" #U_ "], 0xbf7fffff\n\t"                  \
  "v_max_f32 %[" #U_ "], 0xbf7fffff, %[" #U_ "]\n\t"                         \
  "v_mul_f32 %[" #R_ "], %[" #U_ "], %[" #U_ "]\n\t"                         \
  "v_sub_f32 %[" #R_ "], 1.0, %[" #R_ "]\n\t"                                \
  BF_NRM_FREXP(R_, F_, Y_, Z_)
#ifdef BF_BPF_HW_ARITH
#define BF_NRM_FREXP(R_, F_, Y_, Z_)
#else
#define BF_NRM_FREXP(R_, F_, Y_, Z_)                                         \
  "v_and_b32 %[" #F_ "], 0x7fffff, %[" #R_ "]\n\t"                           \
  "v_or_b32 %[" #F_ "], 0.5, %[" #F_ "]\n\t"                                 \
  "v_lshrrev_b32 %[" #Z_ "], 23, %[" #R_ "]\n\t"                             \
  "v_cmp_gt_f32 vcc, 0x3f3504f3, %[" #F_ "]\n\t"                             \
  "v_add_f32 %[" #Y_ "], -1.0, %[" #F_ "]\n\t"                               \
  "v_cndmask_b32 %[" #F_ "], 0, %[" #F_ "], vcc\n\t"                         \
  "v_subbrev_co_u32 %[" #Z_ "], vcc, 0, %[" #Z_ "], vcc\n\t"                 \
  "v_add_f32 %[" #F_ "], %[" #Y_ "], %[" #F_ "]\n\t"                         \
  "v_add_u32 %[" #Z_ "], 0xffffff82, %[" #Z_ "]\n\t"                         \
  "v_cvt_f32_i32 %[" #Z_ "], %[" #Z_ "]\n\t"
#endif
#define BF_2(A_, B_) A_ B_
// one Horner step of both chains: y = y f + K
#define BF_HORNER2(YA_, FA_, YB_, FB_, K_)                                                      \
  "v_fmaak_f32 %[" #YA_ "], %[" #YA_ "], %[" #FA_ "], " #K_ "\n\t"                              \
  "v_fmaak_f32 %[" #YB_ "], %[" #YB_ "], %[" #FB_ "], " #K_ "\n\t"
template <int C0, int C1>
__device__ __forceinline__ void threefry_two_normals_gfx950(uint32_t k0, uint32_t k1, uint32_t k2, float& z0, float& z1) {
#pragma clang fp contract(off)
  uint32_t x0, x1;
  float ua, ra, fa, ya, za, pa, ub, rb, fb, yb, zb, pb;
  asm volatile(
      "v_add_u32 %[x0], %[c0], %[k0]\n\t"
      "v_add_u32 %[x1], %[c1], %[k1]\n\t"
      BF_TFA BF_TFI(k1, k2, 1) BF_TFB BF_TFI(k2, k0, 2) BF_TFA BF_TFI(k0, k1, 3) BF_TFB BF_TFI(k1, k2, 4) BF_TFA BF_TFI(k2, k0, 5)
      BF_NRM_HEAD(x0, ua, ra, fa, ya, za)
      BF_NRM_HEAD(x1, ub, rb, fb, yb, zb)
#ifdef BF_BPF_HW_ARITH
      // bf_set_option "bpf_arith" = 1: log(a) = log2(a) ln 2 on the transcendental unit (outputs the polynomial would not
      // otherwise define are given a value: the block's operands are all early-clobber outputs)
      "v_log_f32 %[ra], %[ra]\n\t"
      "v_log_f32 %[rb], %[rb]\n\t"
      "v_mov_b32 %[ya], 0\n\t"
      "v_mov_b32 %[yb], 0\n\t"
      "v_mov_b32 %[za], 0\n\t"
      "v_mov_b32 %[zb], 0\n\t"
      "v_mul_f32 %[ra], 0x3f317218, %[ra]\n\t"
      "v_mul_f32 %[rb], 0x3f317218, %[rb]\n\t"
#else
      // canon_log_core polynomial, both chains
      "v_mov_b32 %[ya], 0x3d9021bb\n\t"
      "v_mov_b32 %[yb], 0x3d9021bb\n\t"
      BF_HORNER2(ya, fa, yb, fb, 0xbdebd1b8) BF_HORNER2(ya, fa, yb, fb, 0x3def251a) BF_HORNER2(ya, fa, yb, fb, 0xbdfe5d4f)
      BF_HORNER2(ya, fa, yb, fb, 0x3e11e9bf) BF_HORNER2(ya, fa, yb, fb, 0xbe2aae50) BF_HORNER2(ya, fa, yb, fb, 0x3e4cceac)
      BF_HORNER2(ya, fa, yb, fb, 0xbe7ffffc) BF_HORNER2(ya, fa, yb, fb, 0x3eaaaaaa)
      "v_mul_f32 %[ya], %[ya], %[fa]\n\t"
      "v_mul_f32 %[yb], %[yb], %[fb]\n\t"
      "v_mul_f32 %[pa], %[fa], %[fa]\n\t"
      "v_mul_f32 %[pb], %[fb], %[fb]\n\t"
      "v_mul_f32 %[ya], %[ya], %[pa]\n\t"
      "v_mul_f32 %[yb], %[yb], %[pb]\n\t"
      "v_fmac_f32 %[ya], 0xb95e8083, %[za]\n\t"
      "v_fmac_f32 %[yb], 0xb95e8083, %[zb]\n\t"
      "v_fmac_f32 %[ya], -0.5, %[pa]\n\t"
      "v_fmac_f32 %[yb], -0.5, %[pb]\n\t"
      "v_add_f32 %[ra], %[fa], %[ya]\n\t"
      "v_add_f32 %[rb], %[fb], %[yb]\n\t"
      "v_fmac_f32 %[ra], 0x3f318000, %[za]\n\t"
      "v_fmac_f32 %[rb], 0x3f318000, %[zb]\n\t"
#endif
      // w = -2.5 - log(a); central erf_inv polynomial, both chains
      "v_sub_f32 %[fa], 0xc0200000, %[ra]\n\t"
      "v_sub_f32 %[fb], 0xc0200000, %[rb]\n\t"
      "v_mov_b32 %[pa], 0x32f16588\n\t"
      "v_mov_b32 %[pb], 0x32f16588\n\t"
      BF_HORNER2(pa, fa, pb, fb, 0x34b84b36) BF_HORNER2(pa, fa, pb, fb, 0xb66c7357) BF_HORNER2(pa, fa, pb, fb, 0xb6935ac1)
      BF_HORNER2(pa, fa, pb, fb, 0x396532db) BF_HORNER2(pa, fa, pb, fb, 0xbaa45408) BF_HORNER2(pa, fa, pb, fb, 0xbb88e4ef)
      BF_HORNER2(pa, fa, pb, fb, 0x3e7c8f63) BF_HORNER2(pa, fa, pb, fb, 0x3fc02e2f)
      : [x0] "=&v"(x0), [x1] "=&v"(x1), [ua] "=&v"(ua), [ra] "=&v"(ra), [fa] "=&v"(fa), [ya] "=&v"(ya), [za] "=&v"(za), [pa] "=&v"(pa),
        [ub] "=&v"(ub), [rb] "=&v"(rb), [fb] "=&v"(fb), [yb] "=&v"(yb), [zb] "=&v"(zb), [pb] "=&v"(pb)
      : [k0] "v"(k0), [k1] "v"(k1), [k2] "v"(k2), [c0] "n"(C0), [c1] "n"(C1)
      : "vcc");
  const float wa = -ra, wb = -rb;
  if (__builtin_expect(!(wa < 5.0f), 0)) pa = erfinv_tail_poly(wa);
  if (__builtin_expect(!(wb < 5.0f), 0)) pb = erfinv_tail_poly(wb);
  z0 = 1.41421356237309515f * (pa * ua);
  z1 = 1.41421356237309515f * (pb * ub);
}
#undef BF_TFR
#undef BF_TFA
#undef BF_TFB
#undef BF_TFI
#undef BF_NRM_HEAD
#undef BF_NRM_FREXP
#undef BF_2
#undef BF_HORNER2
#endif

}  // namespace bf
