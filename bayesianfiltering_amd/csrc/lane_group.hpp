// Cross-lane primitives for "lane groups": NL = 1,2,4,...,64 consecutive lanes of a wave64
// cooperate on one filter chain (lane j of the group owns column j of the covariance).
// Everything here is register-to-register on gfx950: DPP modifiers inside quads and rows of
// 16, ds_swizzle (LDS crossbar, no memory) up to 32 lanes, v_readlane / ds_bpermute for 64.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace bf {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(float, r);
}

template <int PATTERN>
__device__ __forceinline__ float swizzle(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), PATTERN));
}

// value of lane K of the caller's NL-lane group
template <int NL, int K>
__device__ __forceinline__ float group_bcast(float v) {
  static_assert(K >= 0 && K < NL, "lane index out of group");
  if constexpr (NL == 1) {
    return v;
  } else if constexpr (NL == 2) {
    return dpp_mov<(K) | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6)>(v);
  } else if constexpr (NL == 4) {
    return dpp_mov<K * 0x55>(v);
  } else if constexpr (NL == 8) {
    return swizzle<0x18 | (K << 5)>(v);  // bitmode: and 0b11000, or K
  } else if constexpr (NL == 16) {
    return swizzle<0x10 | (K << 5)>(v);
  } else if constexpr (NL == 32) {
    return swizzle<0x00 | (K << 5)>(v);
  } else {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), K));
  }
}

// value of the lane R places further round the caller's NL-lane group: lane j reads lane (j + R) mod NL
template <int NL, int R>
__device__ __forceinline__ float group_rot(float v) {
  static_assert(NL == 1 || NL == 2 || NL == 4, "rotations are DPP quad permutations");
  constexpr int r = ((R % NL) + NL) % NL;
  if constexpr (r == 0) {
    return v;
  } else if constexpr (NL == 2) {
    return dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
  } else {
    return dpp_mov<(r & 3) | (((r + 1) & 3) << 2) | (((r + 2) & 3) << 4) | (((r + 3) & 3) << 6)>(v);
  }
}

// op(v, value of lane ^ 16) / op(v, value of lane ^ 32) in every lane through the gfx950 row / half
// swaps (v_permlane16_swap, v_permlane32_swap: VALU, no LDS crossbar trip like ds_bpermute).  The
// swap of (v, v) leaves {even-row copy, odd-row copy} in the two results, so op must be symmetric.
template <class Op>
__device__ __forceinline__ float xor16_combine(float v, Op op) {
  float a = v, b = v;  // the instruction rewrites both operands in place: two registers
  // explicit wait states on both sides (VALU write -> swap read, swap write -> VALU read)
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return op(a, b);
}
template <class Op>
__device__ __forceinline__ float xor32_combine(float v, Op op) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return op(a, b);
}

// Two different values at once: x + (x of the lane 16 away) lands in the even rows of 16 lanes, y + (y of the lane 16 away)
// in the odd rows -- the swap exchanges the first operand's odd rows with the second's even rows, so ONE swap and ONE add
// reduce two values where xor16_combine spends them on one.  Same operands in every sum, so the same bits.
__device__ __forceinline__ float pair16_reduce_scatter(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}
// likewise across the two halves of the wave: the lower 32 lanes end with x's total, the upper 32 with y's
__device__ __forceinline__ float pair32_reduce_scatter(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}

// sum over the NL lanes of the group, result in every lane (xor butterfly: 1, 2, 4, ...)
template <int NL>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (NL >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (NL >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (NL >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror: i <-> 7-i
  if constexpr (NL >= 16) v += dpp_mov<0x140>(v);  // row_mirror: i <-> 15-i
  if constexpr (NL >= 32) v += swizzle<0x401F>(v);  // xor 16 within 32
  if constexpr (NL >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int NL>
__device__ __forceinline__ float group_max(float v) {
  if constexpr (NL >= 2) v = fmaxf(v, dpp_mov<0xB1>(v));
  if constexpr (NL >= 4) v = fmaxf(v, dpp_mov<0x4E>(v));
  if constexpr (NL >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if constexpr (NL >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  if constexpr (NL >= 32) v = fmaxf(v, swizzle<0x401F>(v));
  if constexpr (NL >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, END)
template <int I, int END, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < END) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, END>(f);
  }
}

constexpr int next_pow2(int n) { return n <= 1 ? 1 : 2 * next_pow2((n + 1) / 2); }

}  // namespace bf
