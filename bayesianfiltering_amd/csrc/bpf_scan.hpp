// bpf_scan: batched bootstrap particle filter, one workgroup per trajectory.
//
// Replaces the lax.scan body of bootstrap_particle_filter (gaussfiltax/inference.py:1330-1377):
//   keys = split(key, N+1)                                     :1342
//   x_i  = f(x_i, q0 + chol(Q) normal(keys[1+i]), u)            :1344-1345, models.py:82-84
//   lls  = emission_distribution_log_prob(x_i, y, u)            :1348-1349
//   lls -= max; w = exp(lls) * w; w /= sum(w)                   :1350-1353
//   if 1 / sum(w^2) < ess_threshold * N: _resample              :1356-1357, utils.py:207-214
// and the initial draw  x_i ~ MVN(m0, P0) with keys[1+i] of split(key, N+1)  (:1369-1373).
//
// Mapping (gfx950).  The N particles of a trajectory live in the VGPRs of one workgroup:
// thread tid owns the PPT consecutive particles tid*PPT .. tid*PPT+PPT-1 (state never leaves
// registers except to be gathered on a resample).  All randomness is counter-based Threefry
// evaluated per lane (bf_rng.hpp) with JAX's split / bits layout, so the stream depends only
// on (key, N), not on the launch geometry.
//   * max / sum over particles: thread-local adjacent-pair tree, xor-butterfly across lanes
//     (adjacent lanes first), then across waves through LDS -- the adjacent-pair tree of the
//     oracle's sum, so the normalised weights feed the CDF with identical rounding;
//   * CDF: workgroup Brent-Kung scan == lax.associative_scan(add) order (the cumsum_assoc of the test oracle
//     cumsum_assoc), kept in LDS; inverse-CDF draw r_i = c[N-1] * (1 - u_i) and a binary search
//     per slot (searchsorted side='left');
//   * gather x <- x[idx]: through an LDS tile, DCH state dimensions per pass.
// Outputs: FULL (weights (N,T), particles (N,T,n) per trajectory, the reference's return value,
// inference.py:1359-1362,1378) and/or SUMMARY per step (weighted mean, ESS, log-evidence
// increment, resampled flag) -- the full history of cfg4 is 4.6 TB and does not fit HBM.
#pragma once
#ifndef BF_JIT
#include <cstring>
#include <cstdlib>
#include "bf_common.hpp"
#endif
#include <type_traits>
#include "kf_math.hpp"
#include "scan_common.hpp"
#include "bf_rng.hpp"
#include "models.hpp"
#include "ssm_device.hpp"

namespace bf {

struct BpfCarry {
  const float* x_in;       // [B][NP][n] or NULL (draw from N(m0, P0))
  const float* w_in;       // [B][NP]
  const uint32_t* key_in;  // [B][2] or NULL (use the launch key for every trajectory)
  float* x_out;
  float* w_out;
  uint32_t* key_out;
};

struct BpfOut {
  float* w;                // weights, element (b, i, t) at b*w_sB + i*w_sN + t*w_sT
  long long w_sB, w_sN, w_sT;
  float* x;                // particles, element (b, i, t, d) at b*x_sB + i*x_sN + t*x_sT + d
  long long x_sB, x_sN, x_sT;
  int* anc;                // ancestors (b, i, t), same strides as w; NULL = not emitted
  float* mean;             // [B][T][n]   sum_i w_i x_i after the resampling decision
  float* ess;              // [B][T]      1 / sum w^2 before resampling
  float* logz;             // [B][T]      log sum_i w_{t-1,i} p(y_t | x_i)
  float* resampled;        // [B][T]      1.0 if resampled at t
};

// Ancestor indices of utils.py:210: idx = choice(key, N, (N,), p = w) = searchsorted(cumsum(w), c[N-1] * (1 - u)),
// with the CDF in lax.associative_scan order (== Brent-Kung: thread tree, wave up-sweep,
// cross-wave scan, down-sweeps) so that it matches the test oracle (cumsum_assoc) bit for bit.
// resampler 1 = systematic positions (i + u0) / N instead of N independent uniforms.
// CDF entry i lives at LDS word i + (i >> 6): the bisection's first levels probe entries a power of two apart -- in a plain
// array those are different addresses of ONE bank (levels 1 ... 5: 2- to 32-way conflicts on every probe); the skew of one
// word per 64 spreads them over the banks.
__device__ __forceinline__ int cdf_at(int i) { return i + (i >> 6); }
__host__ __device__ constexpr int cdf_words(int cap) { return cap + (cap >> 6); }

template <int PPT, int NW>
__device__ __forceinline__ void resample_indices(const float* wn, const bool* valid, int NP, U32x2 kc, int resampler,
                                                 float* cdf, float* red, int* anc) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // CDF in lax.associative_scan order == Brent-Kung: thread tree, wave up-sweep, cross-wave scan, down-sweep
  float tsum[PPT];  // partial sums of the thread tree: tsum[p] = sum of the aligned block ending at p
  BF_UNROLL for (int p = 0; p < PPT; ++p) tsum[p] = wn[p];
  BF_UNROLL for (int s = 1; s < PPT; s <<= 1) BF_UNROLL for (int p = 2 * s - 1; p < PPT; p += 2 * s) tsum[p] += tsum[p - s];
  float v = tsum[PPT - 1];  // thread total
  BF_UNROLL for (int d = 0; d < 6; ++d) {  // wave up-sweep
    const float o = __shfl_up(v, 1 << d, 64);
    if (((lane + 1) & ((2 << d) - 1)) == 0) v += o;
  }
  float excl_wave = 0.f;  // inclusive scan value at the end of the previous wave
  if constexpr (NW > 1) {
    // (red[16 ...) was last read in the previous call, barriers ago: no barrier before the write)
    if (lane == 63) red[16 + wave] = v;
    lds_barrier();
    float r = (lane < NW) ? red[16 + lane] : 0.f;
    BF_UNROLL for (int d = 0; (1 << d) < NW; ++d) {  // up-sweep over the NW wave totals
      const float o = __shfl_up(r, 1 << d, 64);
      if (lane < NW && ((lane + 1) & ((2 << d) - 1)) == 0) r += o;
    }
    BF_UNROLL for (int d = 4; d >= 1; --d) {  // down-sweep
      if ((1 << d) <= NW) {
        const float o = __shfl_up(r, 1 << (d - 1), 64);
        if (lane < NW && lane >= (1 << d) && ((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) r += o;
      }
    }
    const float mine = __shfl(r, wave, 64);
    const float prev = __shfl(r, wave > 0 ? wave - 1 : 0, 64);
    excl_wave = wave > 0 ? prev : 0.f;
    if (lane == 63) v = mine;
  }
  BF_UNROLL for (int d = 6; d >= 1; --d) {  // wave down-sweep (virtual lane -1 = excl_wave)
    const float o = __shfl_up(v, 1 << (d - 1), 64);
    if (((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) v += (lane >= (1 << (d - 1))) ? o : excl_wave;
  }
  // v = inclusive scan at the thread's last slot; E = exclusive prefix of the thread
  float E = __shfl_up(v, 1, 64);
  if (lane == 0) E = excl_wave;
  float c[PPT];
  c[PPT - 1] = v;
  // down-sweep inside the thread: node at p (end of a left half-block of size s) += prefix before its block
  BF_UNROLL for (int p = 0; p < PPT - 1; ++p) c[p] = tsum[p];
  BF_UNROLL for (int s = PPT / 2; s >= 1; s >>= 1) BF_UNROLL for (int p = s - 1; p < PPT - 1; p += 2 * s) {
    // prefix before the block of size 2s that contains p: E for the first block, else c[block_start - 1]
    const int bs = (p / (2 * s)) * (2 * s);
    c[p] = ((bs == 0) ? E : c[bs - 1]) + tsum[p];
  }
  // (the CDF's last readers -- the previous call's searches -- are barriers back: no barrier before the write)
  BF_UNROLL for (int p = 0; p < PPT; ++p) cdf[cdf_at(tid * PPT + p)] = c[p];
  lds_barrier();
  const float total = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cdf[cdf_at(NP - 1)])));
  float u_sys = 0.f;
  if (resampler == 1) u_sys = bits_to_unit(threefry_bits(kc.x, kc.y, 0u, 1u));
  float r[PPT];
  BF_UNROLL for (int p = 0; p < PPT; ++p) {
    const uint32_t i = (uint32_t)(tid * PPT + p);
    if (resampler == 1) r[p] = (((float)i + u_sys) / (float)NP) * total;
    else r[p] = total * (1.0f - bits_to_unit(threefry_bits(kc.x, kc.y, valid[p] ? i : 0u, (uint32_t)NP)));
  }
  // first index with cdf[idx] >= r (searchsorted side='left').  The CDF is non-decreasing and the slots beyond NP hold
  // its total (>= r), so the lower bound over the CAP = 2^k slots is the same index; found by the fixed-trip probe
  // sequence below instead of a bisection loop per draw: no data-dependent branches, and the PPT searches of a thread
  // advance together, so their LDS reads (random addresses, ~100 cycles each) overlap instead of queueing.
  constexpr int CAP = 64 * NW * PPT;
  int pos[PPT];
  BF_UNROLL for (int p = 0; p < PPT; ++p) pos[p] = 0;
  for (int step = CAP >> 1; step >= 1; step >>= 1) {
    float probe[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) probe[p] = cdf[cdf_at(pos[p] + step - 1)];
    BF_UNROLL for (int p = 0; p < PPT; ++p) pos[p] += (probe[p] < r[p]) ? step : 0;
  }
  BF_UNROLL for (int p = 0; p < PPT; ++p) anc[p] = pos[p] < NP - 1 ? pos[p] : NP - 1;
}

// Sum over the 64 lanes of a wave, total in lane 63: six DPP adds on the vector ALU (quad permutes, rotations inside the
// 16-lane rows, the two row broadcasts) instead of six ds_bpermute round trips through the LDS crossbar.  Not the
// adjacent-pair tree order: for sums whose rounding is not pinned by the oracle (the weighted-mean summary).
__device__ __forceinline__ float wave_sum_dpp_lane63(float v) {
  auto dpp = [](float x, auto ctrl, auto rowmask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                 decltype(rowmask)::value, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});   // quad_perm [1, 0, 3, 2]
  v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});   // quad_perm [2, 3, 0, 1]
  v += dpp(v, std::integral_constant<int, 0x124>{}, std::integral_constant<int, 0xf>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{}, std::integral_constant<int, 0xf>{});  // row_ror:8
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1, 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2, 3
  return v;
}

// The adjacent-pair tree over the 64 lanes of a wave, result in lane 63, on the vector ALU: the xor-butterfly's partners
// at distance 1, 2, 4, 8 are reached by quad permutes and the two row mirrors (each lane of a finished group holds the
// group's value, so any lane of the neighbouring group serves), the rows by the two row broadcasts.  op is commutative
// (float add, nanmax), so the bits are those of the butterfly -- without its six trips through the LDS crossbar.
template <class OP>
__device__ __forceinline__ float wave_tree_dpp_lane63(float v, OP op) {
  auto dpp = [](float x, auto ctrl, auto rowmask, float keep) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, decltype(rowmask)::value, 0xf, false));
  };
  v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{}, v));   // quad_perm [1, 0, 3, 2]
  v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{}, v));   // quad_perm [2, 3, 0, 1]
  v = op(v, dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{}, v));  // row_half_mirror
  v = op(v, dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{}, v));  // row_mirror
  // rows 1 and 3 take the total of rows 0 and 2; then row 3 takes that of row 1 (the other rows keep what they have)
  const float r15 = dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}, v);
  v = ((threadIdx.x >> 4) & 1) ? op(v, r15) : v;
  const float r31 = dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}, v);
  v = ((threadIdx.x & 63) >= 48) ? op(v, r31) : v;
  return v;
}

// x_i ~ MVN(m0, P0) with key number i + 1 of split(key, N + 1) (inference.py:1369-1373): z = normal(key_i, (n,)),
// x = m0 + chol(P0) z, the row's fma chain with c ascending -- the arithmetic of the inlined form, in rolled loops.
template <int N, int DQ, int M>
__device__ __attribute__((noinline)) void draw_initial_particle(const BpfModel<N, DQ, M>* __restrict__ mdl, uint32_t k0, uint32_t k1,
                                                                uint32_t i, uint32_t NP, float* __restrict__ out) {
  const U32x2 ki = threefry_split(k0, k1, i + 1u, NP + 1u);
  float z[N];
#pragma unroll 1
  for (int d = 0; d < N; ++d) z[d] = bits_to_normal(threefry_bits(ki.x, ki.y, (uint32_t)d, (uint32_t)N));
#pragma unroll 1
  for (int d = 0; d < N; ++d) {
    float s = 0.f;
#pragma unroll 1
    for (int c = 0; c <= d; ++c) s = __builtin_fmaf(mdl->L0[d * N + c], z[c], s);
    out[d] = mdl->m0[d] + s;
  }
}

// The xor-butterfly (adjacent-pair tree) over the first NWV <= 16 lanes of a row, result in every one of them, on the vector
// ALU: partners at distance 1, 2 by quad permutes, 4, 8 by the two row mirrors (every lane of a finished group holds the
// group's value, so any lane of the neighbouring group serves).  op is commutative: the bits of the butterfly.
template <int NWV, class OP>
__device__ __forceinline__ float row_tree_dpp(float v, OP op) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                                 decltype(ctrl)::value, 0xf, 0xf, false));
  };
  if constexpr (NWV > 1) v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // quad_perm [1, 0, 3, 2]
  if constexpr (NWV > 2) v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // quad_perm [2, 3, 0, 1]
  if constexpr (NWV > 4) v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  if constexpr (NWV > 8) v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));  // row_mirror
  return v;
}

template <int J, int H, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < H) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, H>(f);
  }
}

// NaN-propagating maximum (jnp.max semantics)
__device__ __forceinline__ float nanmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// Every launch parameter in ONE struct: it is the kernel's only argument, so its layout IS the kernarg segment's, and the
// time loop re-reads the fields it needs from there (scalar loads through the constant cache, at the point of use) instead
// of holding ~75 scalar registers of pointers and strides live across a 20 000-instruction loop body -- which the register
// allocator answered with 700+ scalar spills into vector lanes and, from there, vector spills to scratch.
template <int N, int DQ, int M>
struct BpfArgs {
  CView y;
  const float* uptr;
  long long u_sB, u_sT;
  BpfCarry carry;
  BpfOut out;
  long long B, T;
  int NP;
  float ess_threshold;
  int resampler;
  uint32_t key0, key1;
};

// The kernel proper, as a device function: every kernel entry that runs it -- bpf_scan_kernel below, the `extern "C"` entries
// of a build compiled at run time with a caller's functions (user_model.hip) -- has the parameter list
// (const BpfModel*, BpfArgs by value), which is what the kernarg-segment reads below rely on.
template <int N, int DQ, int M, int PPT, int NW, class SP = SpecRuntime>
__device__ __forceinline__ void bpf_scan_body(const BpfModel<N, DQ, M>* __restrict__ mdlp) {
  // `ka[fresh()]`: the argument struct as it lies in the kernarg segment (behind the model pointer, which stays a
  // `__restrict__` parameter of its own: that is what lets the compiler read the model with scalar loads), behind an
  // offset the compiler cannot see through (always 0), so that a field read inside the time loop is a fresh scalar load
  // there and not a register held since entry
  static_assert(alignof(BpfArgs<N, DQ, M>) == 8, "kernarg layout: the struct follows the 8-byte model pointer");
  typedef const char __attribute__((address_space(4))) * KernargBytes;   // (the constant address space: scalar loads)
  const BpfArgs<N, DQ, M>* const ka = (const BpfArgs<N, DQ, M>*)((KernargBytes)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  auto fresh = []() __attribute__((always_inline)) {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
  };
  const long long T = ka->T;
  const int NP = ka->NP;
  constexpr int NT = 64 * NW;
  constexpr int CAP = NT * PPT;                 // particle slots (power of two)
  constexpr int DCH = (PPT >= 16) ? 1 : ((N >= 8) ? 8 : N);  // state dimensions gathered per LDS pass
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long b = blockIdx.x;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CW = (cdf_words(CAP) + 3) & ~3;
  float* cdf = lds;                 // CAP floats, skewed (cdf_at)
  float* red = lds + CW;            // cross-wave scratch: [0, 16) and [32, 48) the two slot groups of block_reduce, [16, 32) the CDF's wave totals
  float* mpart = lds + CW + 64;     // NW * N floats: per-wave partial sums of the weighted-mean summary
  float* tile = mpart + ((NW * N + 3) & ~3);   // CAP * DCH floats (gather tile)

  // ---- workgroup reductions in the oracle's adjacent-pair tree order
  auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  auto uniform_u = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  // ONE barrier per reduction: the wave totals go to one of two slot groups, alternately -- a wave that runs ahead into the
  // next reduction writes the OTHER group, and by the time a group is written again every wave has passed the barrier that
  // followed its last reads
  int rslot = 0;
  auto block_reduce = [&](float v, auto op) {  // v already reduced over the thread's own slots
    v = wave_tree_dpp_lane63(v, op);
    if constexpr (NW > 1) {
      float* rs = red + 32 * rslot;
      rslot ^= 1;
      if (lane == 63) rs[wave] = v;
      lds_barrier();
      float r = (lane < NW) ? rs[lane] : rs[0];
      v = uniform(row_tree_dpp<NW>(r, op));   // the xor-butterfly over the NW wave totals, on the vector ALU; lane 0's copy
    } else {
      v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    }
    return v;   // the same value in every lane, and known to the compiler as such (a scalar register)
  };
  auto thread_tree = [&](const float* e, auto op) {  // adjacent-pair tree over the PPT own slots
    float t[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) t[p] = e[p];
    BF_UNROLL for (int s = 1; s < PPT; s <<= 1) BF_UNROLL for (int p = 0; p + s < PPT; p += 2 * s) t[p] = op(t[p], t[p + s]);
    return t[0];
  };
  auto fadd = [](float a, float c) { return a + c; };

  // ---- state
  float x[PPT][N], w[PPT];
  uint32_t k0, k1;
  bool valid[PPT];
  BF_UNROLL for (int p = 0; p < PPT; ++p) valid[p] = (tid * PPT + p) < NP;
  {
    const BpfCarry carry = ka->carry;
    if (carry.key_in) {
      k0 = carry.key_in[b * 2];
      k1 = carry.key_in[b * 2 + 1];
    } else {
      k0 = ka->key0;
      k1 = ka->key1;
    }
    k0 = uniform_u(k0);
    k1 = uniform_u(k1);
  }
  if (ka->carry.x_in) {
    const BpfCarry carry = ka->carry;
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const int i = valid[p] ? tid * PPT + p : 0;
      BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = carry.x_in[(b * NP + i) * N + d];
      w[p] = valid[p] ? carry.w_in[b * NP + i] : 0.f;
    }
  } else {
    // inference.py:1369-1373: keys = split(key, N+1); next_key = keys[0]; x_i ~ MVN(m0, P0) with keys[1+i]
    // (once per trajectory, in a real function with rolled loops: inlined and unrolled its N (N + 1) / 2 scalar loads of
    // chol(P0) alone cost the whole kernel several vector registers of spilled scalars)
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const uint32_t i = valid[p] ? (uint32_t)(tid * PPT + p) : 0u;
      float x0[N];
      draw_initial_particle<N, DQ, M>(mdlp, k0, k1, i, (uint32_t)NP, x0);
      BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = x0[d];
      w[p] = valid[p] ? 1.0f / (float)NP : 0.f;
    }
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
    k0 = nk.x;
    k1 = nk.y;
  }

#ifdef BF_BPF_PHASE_TIMERS  // debug build: per-phase wall-clock ticks (10 ns) of every wave of workgroup 0 (scripts/bpf_phase_probe.py)
  long long tim[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = (long long)__builtin_amdgcn_s_memrealtime();
#define BF_TICK(I_) { const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); tim[I_] += now_ - tlast; tlast = now_; }
#else
#define BF_TICK(I_)
#endif
  for (long long t = 0; t < T; ++t) {
    // canonical arithmetic of the weight path (bf_canon_math.hpp): products are rounded before they enter a tree or a sum
#pragma clang fp contract(off)
    BF_TICK(7)
    float yv[M];
    float u0;
    {
      const BpfArgs<N, DQ, M>& a_ = ka[fresh()];
      const CView y = a_.y;
      BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = y.p[b * y.sB + t * y.sT + a * y.sE];
      u0 = a_.uptr ? a_.uptr[b * a_.u_sB + t * a_.u_sT] : 0.f;
    }

    // ---- propagate (inference.py:1342-1345, models.py:82-84) and log-weight (:1348-1349)
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);  // next_key = keys[0]
    float ll[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const uint32_t i = valid[p] ? (uint32_t)(tid * PPT + p) : 0u;
      const U32x2 ki = threefry_split(k0, k1, i + 1u, (uint32_t)NP + 1u);
      float xn[N];
      // the model is re-read (scalar loads through the constant cache) for every particle instead of being held: hoisted
      // out of the time loop its matrices occupy several hundred scalar registers, which spill through vector lanes and
      // from there to scratch (1024 x 4 geometry: 492 -> 82 spilled VGPRs, 52 -> 42 ms at cfg4's shape, B = 1024, T = 100)
      int zoff = 0;
      asm volatile("" : "+s"(zoff));
      const BpfModel<N, DQ, M>& mdl = mdlp[zoff];
      float llp;
      if constexpr (SP::fixed && DQ == N) {
        if constexpr (SP::g_identity && SP::lq_diag) {
          // identity noise input, diagonal chol(Q): the noise-free part first (the old state dies there), then every Threefry
          // block's two normals are scaled and added as they arrive -- x' = g(x) + (q0 + L_dd z_d), the operations of
          // draw_dynamics_noise + dyn_value one by one -- so that neither a q vector nor a z vector is ever live; the
          // scheduling fences keep the compiler from starting all blocks at once (16 chains in flight cost ~60 VGPRs)
          dyn_base_t<N, DQ, BpfModel<N, DQ, M>, SP>(mdl, x[p], u0, xn);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int h = (DQ + 1) / 2;
          if constexpr (SP::impl == 0) {
            // the hand-scheduled block (bf_rng.hpp): Threefry + both normals in 12 registers, constants as literals
            const uint32_t ks2 = ki.x ^ ki.y ^ 0x1BD11BDAu;
            static_for<0, h>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              float za, zb;
              threefry_two_normals_gfx950<j, (h + j < DQ) ? h + j : 0>(ki.x, ki.y, ks2, za, zb);
              x[p][j] = xn[j] + (mdl.q0[j] + mdl.LQd[j] * za);
              if constexpr (h + j < DQ) x[p][h + j] = xn[h + j] + (mdl.q0[h + j] + mdl.LQd[h + j] * zb);
              // the two results are consumed HERE: the empty asm pins the new state entries to this point (without it the
              // compiler sinks the last three operations towards their first use, past the following blocks, and keeps every
              // block's polynomial values and u alive -- 32 registers -- until then)
              if constexpr (h + j < DQ) asm volatile("" : "+v"(x[p][j]), "+v"(x[p][h + j]));
              else asm volatile("" : "+v"(x[p][j]));
            });
          } else {   // the same in plain C++ (kept for reference builds: SpecFixed<..., 1>)
            BF_UNROLL for (int j = 0; j < h; ++j) {
              const U32x2 o = threefry2x32(ki.x, ki.y, (uint32_t)j, (h + j < DQ) ? (uint32_t)(h + j) : 0u);
              x[p][j] = xn[j] + (mdl.q0[j] + mdl.LQd[j] * bits_to_normal(o.x));
              if (h + j < DQ) x[p][h + j] = xn[h + j] + (mdl.q0[h + j] + mdl.LQd[h + j] * bits_to_normal(o.y));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          llp = emission_loglik<N, DQ, M, SP>(mdl, x[p], u0, yv);
        } else {
          float q[DQ];
          draw_dynamics_noise<N, DQ, M, SP>(mdl, ki, q);
          dyn_value<N, DQ, M, SP>(mdl, x[p], q, u0, xn);
          BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = xn[d];
          llp = emission_loglik<N, DQ, M, SP>(mdl, xn, u0, yv);
        }
      } else {
        float q[DQ];
        draw_dynamics_noise<N, DQ, M, SP>(mdl, ki, q);
        dyn_value<N, DQ, M, SP>(mdl, x[p], q, u0, xn);
        BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = xn[d];
        llp = emission_loglik<N, DQ, M, SP>(mdl, xn, u0, yv);
      }
      ll[p] = valid[p] ? llp : -__builtin_inff();
      // one particle at a time: interleaving the PPT independent Threefry / erfinv chains overruns the
      // 128-VGPR budget of the 1024-thread geometry and spills
      __builtin_amdgcn_sched_barrier(0);
    }

    BF_TICK(0)
    // ---- reweight (inference.py:1350-1353)
    // (workgroup-wide results are the same in every lane: telling the compiler so keeps them -- and everything computed from
    // them and the key, the three `split`s of this step included -- in scalar registers and on the scalar unit)
    const float mx = block_reduce(thread_tree(ll, nanmax), nanmax);
    BF_TICK(1)
    float e[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) e[p] = valid[p] ? canon_exp(ll[p] - mx) * w[p] : 0.f;
    const float tot = block_reduce(thread_tree(e, fadd), fadd);
    BF_TICK(2)
    float wn[PPT], w2[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      wn[p] = valid[p] ? e[p] / tot : 0.f;
      w2[p] = wn[p] * wn[p];
    }
    const float ess = 1.0f / block_reduce(thread_tree(w2, fadd), fadd);
    const bool do_resample = ess < ka[fresh()].ess_threshold * (float)NP;  // inference.py:1356 (NaN compares false)

    BF_TICK(3)
    int anc[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) anc[p] = tid * PPT + p;
    if (do_resample) {
      // ---- utils.py:207-214: keys = split(key, 2); idx = choice(keys[0], N, (N,), p = w); next_key = keys[1]
      const U32x2 kc = threefry_split(nk.x, nk.y, 0u, 2u);
      const U32x2 kn = threefry_split(nk.x, nk.y, 1u, 2u);
      // gather through LDS, DCH dimensions per pass.  The first pass is staged BEFORE the ancestors are drawn: those
      // DCH x PPT state registers are then dead while the CDF, the uniforms and the searches run.
      // Tile layout: DCH / 4 planes of 16-byte records, particle i = tid * PPT + p at record p * NT + tid of every plane --
      // the lanes of a wave write consecutive 16-byte records (conflict-free; with the planes interleaved record by record
      // the 32-byte lane stride made every store two-way conflicted), and a drawn ancestor is fetched as DCH / 4 16-byte
      // reads instead of DCH dword reads DCH banks apart.  (Dimensions that are not multiples of four keep the
      // dimension-major dword layout.)
      constexpr bool VEC = (DCH % 4 == 0) && (N % 4 == 0);
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      auto put = [&](int d0) __attribute__((always_inline)) {
        if constexpr (VEC) {
          f32x4* t4 = reinterpret_cast<f32x4*>(tile);
          BF_UNROLL for (int p = 0; p < PPT; ++p) BF_UNROLL for (int v = 0; v < DCH / 4; ++v)
              if (d0 + 4 * v < N)
                t4[v * CAP + p * NT + tid] = f32x4{x[p][d0 + 4 * v], x[p][d0 + 4 * v + 1], x[p][d0 + 4 * v + 2], x[p][d0 + 4 * v + 3]};
        } else {
          BF_UNROLL for (int p = 0; p < PPT; ++p) BF_UNROLL for (int d = 0; d < DCH; ++d)
              if (d0 + d < N) tile[d * CAP + tid * PPT + p] = x[p][d0 + d];
        }
      };
      auto get = [&](int d0) __attribute__((always_inline)) {
        if constexpr (VEC) {
          const f32x4* t4 = reinterpret_cast<const f32x4*>(tile);
          BF_UNROLL for (int p = 0; p < PPT; ++p) {
            const int rec = (anc[p] % PPT) * NT + anc[p] / PPT;
            BF_UNROLL for (int v = 0; v < DCH / 4; ++v)
                if (d0 + 4 * v < N) {
                  const f32x4 t = t4[v * CAP + rec];
                  x[p][d0 + 4 * v] = t.x; x[p][d0 + 4 * v + 1] = t.y; x[p][d0 + 4 * v + 2] = t.z; x[p][d0 + 4 * v + 3] = t.w;
                }
          }
        } else {
          BF_UNROLL for (int p = 0; p < PPT; ++p) BF_UNROLL for (int d = 0; d < DCH; ++d)
              if (d0 + d < N) x[p][d0 + d] = tile[d * CAP + anc[p]];
        }
      };
      // (the tile's last readers -- the previous resampling step's second gather pass -- are at least three barriers back)
      put(0);
      resample_indices<PPT, NW>(wn, valid, NP, kc, ka[fresh()].resampler, cdf, red, anc);  // (its barriers publish the tile)
      BF_TICK(4)
      get(0);
      BF_UNROLL for (int d0 = DCH; d0 < N; d0 += DCH) {
        lds_barrier();
        put(d0);
        lds_barrier();
        get(d0);
      }
      BF_UNROLL for (int p = 0; p < PPT; ++p) w[p] = valid[p] ? 1.0f / (float)NP : 0.f;
      k0 = kn.x;
      k1 = kn.y;
    } else {
      BF_UNROLL for (int p = 0; p < PPT; ++p) w[p] = wn[p];
      k0 = nk.x;
      k1 = nk.y;
    }

    BF_TICK(5)
    // ---- emit
    const BpfOut out = ka[fresh()].out;
    if (out.w || out.anc || out.x) BF_UNROLL for (int p = 0; p < PPT; ++p) if (valid[p]) {
      const long long i = tid * PPT + p;
      if (out.w) out.w[b * out.w_sB + i * out.w_sN + t * out.w_sT] = w[p];
      if (out.anc) out.anc[b * out.w_sB + i * out.w_sN + t * out.w_sT] = anc[p];
      if (out.x) BF_UNROLL for (int d = 0; d < N; ++d) out.x[b * out.x_sB + i * out.x_sN + t * out.x_sT + d] = x[p][d];
    }
    if (out.mean) {
      float part[N];
      BF_UNROLL for (int d = 0; d < N; ++d) {
        float s = 0.f;
        BF_UNROLL for (int p = 0; p < PPT; ++p) s = fmaf(w[p], x[p][d], s);
        part[d] = wave_sum_dpp_lane63(s);
      }
      // (mpart's last readers are the previous step's, the three reduction barriers back)
      if (lane == 63) BF_UNROLL for (int d = 0; d < N; ++d) mpart[wave * N + d] = part[d];
      lds_barrier();
      if (tid < N) {
        float s = 0.f;
        for (int wv = 0; wv < NW; ++wv) s += mpart[wv * N + tid];
        out.mean[(b * T + t) * N + tid] = s;
      }
    }
    if (tid == 0) {
      if (out.ess) out.ess[b * T + t] = ess;
      if (out.logz) out.logz[b * T + t] = mx + canon_log(tot);
      if (out.resampled) out.resampled[b * T + t] = do_resample ? 1.0f : 0.0f;
    }
    BF_TICK(6)
  }

  const BpfCarry carry = ka[fresh()].carry;
#ifdef BF_BPF_PHASE_TIMERS
  if (b == 0 && lane == 0 && carry.w_out) {
    BF_UNROLL for (int i = 0; i < 8; ++i) carry.w_out[wave * 8 + i] = (float)tim[i];
    return;
  }
  if (b == 0) return;
#endif
  BF_UNROLL for (int p = 0; p < PPT; ++p) if (valid[p]) {
    const long long i = tid * PPT + p;
    if (carry.x_out) BF_UNROLL for (int d = 0; d < N; ++d) carry.x_out[(b * NP + i) * N + d] = x[p][d];
    if (carry.w_out) carry.w_out[b * NP + i] = w[p];
  }
  if (tid == 0 && carry.key_out) {
    carry.key_out[b * 2] = k0;
    carry.key_out[b * 2 + 1] = k1;
  }
}

template <int N, int DQ, int M, int PPT, int NW, class SP = SpecRuntime>
__global__ void __launch_bounds__(64 * NW)
bpf_scan_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, const BpfArgs<N, DQ, M> args_by_value) {
  (void)args_by_value;   // (read from the kernarg segment inside)
  bpf_scan_body<N, DQ, M, PPT, NW, SP>(mdlp);
}

#ifndef BF_JIT   // host side: launches
// ---------------------------------------------------------------------------------------
template <int N, int DQ, int M, int PPT, int NW, class SP = SpecRuntime>
static inline int launch_bpf_cfg(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B,
                          long long T, int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr,
                          const BpfOut& out, hipStream_t stream) {
  constexpr int CAP = 64 * NW * PPT;
  constexpr int DCH = (PPT >= 16) ? 1 : ((N >= 8) ? 8 : N);
  const size_t lds_bytes = sizeof(float) * (size_t)(((cdf_words(CAP) + 3) & ~3) + 64 + ((NW * N + 3) & ~3) + CAP * DCH);
  if (lds_bytes > 160 * 1024) return set_error(BF_EUNSUPPORTED, "particle tile exceeds the 160 KiB LDS");
  BpfArgs<N, DQ, M> a;
  std::memset(&a, 0, sizeof(a));
  a.y = CView{y->ptr, y->sB, y->sT, y->sE};
  a.uptr = (u && u->ptr) ? u->ptr : nullptr;
  a.u_sB = u ? u->sB : 0;
  a.u_sT = u ? u->sT : 0;
  a.carry = cr; a.out = out; a.B = B; a.T = T; a.NP = NP; a.ess_threshold = ess; a.resampler = resampler;
  a.key0 = key[0]; a.key1 = key[1];
  auto kern = bpf_scan_kernel<N, DQ, M, PPT, NW, SP>;
  if (lds_bytes > 64 * 1024)
    BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64 * NW), lds_bytes, stream, d_mdl, a);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

extern Option g_bpf_variant;   // tuning hook (bf_set_option "bpf_variant")
extern Option g_bpf_hbm_mode;  // bf_set_option "bpf_hbm_mode": 0 = choose, 1 = workgroup per trajectory, 2 = per chunk
extern Option g_bpf_spec;      // bf_set_option "bpf_spec": 1 = compile-time model structure where an instance exists (default), 0 = off

// bpf_big.hpp / bpf_wide.hpp: particle counts beyond the in-register capacities (declared here, defined after the kernels there)
template <int N, int DQ, int M>
static inline int launch_bpf_hbm_dims(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B,
                                      long long T, int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr,
                                      const BpfOut& out, hipStream_t stream);

template <int N, int DQ, int M>
static inline int launch_bpf_dims(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                           int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out,
                           hipStream_t stream) {
  BpfModel<N, DQ, M> h;
  std::memset(&h, 0, sizeof(h));  // the constant cache compares contents
  int rc = fill_bpf_model<N, DQ, M>(bp, h);
  if (rc != BF_OK) return rc;
  const void* dv = nullptr;
  rc = device_constants(&h, sizeof(h), stream, &dv);
  if (rc != BF_OK) return rc;
  const BpfModel<N, DQ, M>* d_mdl = static_cast<const BpfModel<N, DQ, M>*>(dv);
  // few trajectories with thousands of particles: one workgroup per trajectory would leave the chip idle (a step of the
  // in-register kernel takes ~19 us per 1024 particles on its one CU); the workgroup-per-chunk kernels of bpf_wide.hpp
  // spread the particles over the CUs at ~25 us of launches per step
  const bool spread = NP > 2048 && B <= 32 && g_bpf_hbm_mode != 1;
  if (spread) return launch_bpf_hbm_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  // smallest compiled particle capacity that holds NP, for the model's structure as a run-time (SpecRuntime) or compile-time
  // (SpecFixed) property
  auto by_capacity = [&](auto spec) -> int {
    using SP = decltype(spec);
    if (NP <= 64) return launch_bpf_cfg<N, DQ, M, 1, 1, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    if (NP <= 128) return launch_bpf_cfg<N, DQ, M, 1, 2, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);   // (the reference's usual 100)
    if (NP <= 256) return launch_bpf_cfg<N, DQ, M, 1, 4, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    if (NP <= 512) return launch_bpf_cfg<N, DQ, M, 1, 8, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    if (NP <= 1024) return launch_bpf_cfg<N, DQ, M, 1, 16, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    if (NP <= 4096) {
      // two geometries for the largest capacity: 1024 threads x 4 particles (128-VGPR budget, variant 0, the default) or
      // 512 threads x 8 particles (256-VGPR budget, variant 1)
      if (g_bpf_variant == 1) return launch_bpf_cfg<N, DQ, M, 8, 8, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
      return launch_bpf_cfg<N, DQ, M, 4, 16, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    }
    if (NP <= 16384 && N <= 4 && DQ <= 4) {
      // small states: 16 particles per thread still fit the registers (1024 threads x 16; the gather goes one
      // state dimension at a time so that CDF + tile stay within the LDS)
      if constexpr (N <= 4 && DQ <= 4) return launch_bpf_cfg<N, DQ, M, 16, 16, SP>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    }
    return launch_bpf_hbm_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);  // particles in HBM (bpf_big.hpp, bpf_wide.hpp)
  };
  // The structure BASELINE configs[3] has -- Lorenz-96 dynamics with identity noise input, diagonal chol(Q), an emission
  // that selects the even states, diagonal chol(R) -- as a compile-time instance (bf_set_option "bpf_spec" = 0 turns it
  // off; results are bit-identical either way: tests/test_bpf_gpu.py)
  if constexpr (N == DQ && N >= 8 && 2 * M <= N + 1) {
    if (g_bpf_spec != 0 && h.dyn_id == DYN_LORENZ96 && h.emi_id == EMI_LINEAR && h.g_identity && h.lq_diag && h.lr_diag && h.h_pick && NP <= 4096) {
      return by_capacity(SpecFixed<DYN_LORENZ96, EMI_LINEAR, true, true, true, true>{});
    }
  }
  return by_capacity(SpecRuntime{});
}

#endif  // BF_JIT

}  // namespace bf
