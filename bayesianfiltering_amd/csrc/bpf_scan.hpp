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
#include <cstring>
#include <type_traits>
#include <cstdlib>
#include "bf_common.hpp"
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
    lds_barrier();
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
  lds_barrier();
  BF_UNROLL for (int p = 0; p < PPT; ++p) cdf[tid * PPT + p] = c[p];
  lds_barrier();
  const float total = cdf[NP - 1];
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
    BF_UNROLL for (int p = 0; p < PPT; ++p) probe[p] = cdf[pos[p] + step - 1];
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

// NaN-propagating maximum (jnp.max semantics)
__device__ __forceinline__ float nanmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

template <int N, int DQ, int M, int PPT, int NW, class SP = SpecRuntime>
__global__ void __launch_bounds__(64 * NW)
bpf_scan_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB,
                long long u_sT, BpfCarry carry, BpfOut out, long long B, long long T, int NP, float ess_threshold,
                int resampler, uint32_t key0, uint32_t key1) {
  constexpr int NT = 64 * NW;
  constexpr int CAP = NT * PPT;                 // particle slots (power of two)
  constexpr int DCH = (PPT >= 16) ? 1 : ((N >= 8) ? 8 : N);  // state dimensions gathered per LDS pass
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long b = blockIdx.x;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cdf = lds;                 // CAP floats
  float* red = lds + CAP;           // 2 * 16 floats of cross-wave scratch
  float* tile = lds + CAP + 64;     // CAP * DCH floats (gather tile; also N-vector reductions)

  // ---- workgroup reductions in the oracle's adjacent-pair tree order
  auto block_reduce = [&](float v, auto op) {  // v already reduced over the thread's own slots
    v = wave_tree_dpp_lane63(v, op);
    if constexpr (NW > 1) {
      lds_barrier();
      if (lane == 63) red[wave] = v;
      lds_barrier();
      float r = (lane < NW) ? red[lane] : red[0];
      BF_UNROLL for (int off = 1; off < NW; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
      v = __shfl(r, 0, 64);
    } else {
      v = __shfl(v, 63, 64);
    }
    return v;
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
  if (carry.key_in) {
    k0 = carry.key_in[b * 2];
    k1 = carry.key_in[b * 2 + 1];
  } else {
    k0 = key0;
    k1 = key1;
  }
  if (carry.x_in) {
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const int i = valid[p] ? tid * PPT + p : 0;
      BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = carry.x_in[(b * NP + i) * N + d];
      w[p] = valid[p] ? carry.w_in[b * NP + i] : 0.f;
    }
  } else {
    // inference.py:1369-1373: keys = split(key, N+1); next_key = keys[0]; x_i ~ MVN(m0, P0) with keys[1+i]
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const uint32_t i = valid[p] ? (uint32_t)(tid * PPT + p) : 0u;
      const U32x2 ki = threefry_split(k0, k1, i + 1u, (uint32_t)NP + 1u);
      float z[N];
      BF_UNROLL for (int d = 0; d < N; ++d) z[d] = bits_to_normal(threefry_bits(ki.x, ki.y, (uint32_t)d, (uint32_t)N));
      BF_UNROLL for (int d = 0; d < N; ++d) {
        float s = 0.f;
        BF_UNROLL for (int c = 0; c <= d; ++c) s = __builtin_fmaf(mdl.L0[d * N + c], z[c], s);
        x[p][d] = mdl.m0[d] + s;
      }
      w[p] = valid[p] ? 1.0f / (float)NP : 0.f;
    }
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
    k0 = nk.x;
    k1 = nk.y;
  }

  for (long long t = 0; t < T; ++t) {
    // canonical arithmetic of the weight path (bf_canon_math.hpp): products are rounded before they enter a tree or a sum
#pragma clang fp contract(off)
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = y.p[b * y.sB + t * y.sT + a * y.sE];
    const float u0 = uptr ? uptr[b * u_sB + t * u_sT] : 0.f;

    // ---- propagate (inference.py:1342-1345, models.py:82-84) and log-weight (:1348-1349)
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);  // next_key = keys[0]
    float ll[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      const uint32_t i = valid[p] ? (uint32_t)(tid * PPT + p) : 0u;
      const U32x2 ki = threefry_split(k0, k1, i + 1u, (uint32_t)NP + 1u);
      float q[DQ], xn[N];
      // the model is re-read (scalar loads through the constant cache) for every particle instead of being held: hoisted
      // out of the time loop its matrices occupy several hundred scalar registers, which spill through vector lanes and
      // from there to scratch (1024 x 4 geometry: 492 -> 82 spilled VGPRs, 52 -> 42 ms at cfg4's shape, B = 1024, T = 100)
      // (a SpecFixed instance reads a few dozen fields only: they stay in scalar registers across particles and steps)
      int zoff = 0;
      if constexpr (!SP::fixed) asm volatile("" : "+s"(zoff));
      const BpfModel<N, DQ, M>& mdl = mdlp[zoff];
      draw_dynamics_noise<N, DQ, M, SP>(mdl, ki, q);
      dyn_value<N, DQ, M, SP>(mdl, x[p], q, u0, xn);
      BF_UNROLL for (int d = 0; d < N; ++d) x[p][d] = xn[d];
      const float llp = emission_loglik<N, DQ, M, SP>(mdl, xn, u0, yv);
      ll[p] = valid[p] ? llp : -__builtin_inff();
      // one particle at a time: interleaving the PPT independent Threefry / erfinv chains overruns the
      // 128-VGPR budget of the 1024-thread geometry and spills
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- reweight (inference.py:1350-1353)
    const float mx = block_reduce(thread_tree(ll, nanmax), nanmax);
    float e[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) e[p] = valid[p] ? canon_exp(ll[p] - mx) * w[p] : 0.f;
    const float tot = block_reduce(thread_tree(e, fadd), fadd);
    float wn[PPT], w2[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) {
      wn[p] = valid[p] ? e[p] / tot : 0.f;
      w2[p] = wn[p] * wn[p];
    }
    const float ess = 1.0f / block_reduce(thread_tree(w2, fadd), fadd);
    const bool do_resample = ess < ess_threshold * (float)NP;  // inference.py:1356 (NaN compares false)

    int anc[PPT];
    BF_UNROLL for (int p = 0; p < PPT; ++p) anc[p] = tid * PPT + p;
    if (do_resample) {
      // ---- utils.py:207-214: keys = split(key, 2); idx = choice(keys[0], N, (N,), p = w); next_key = keys[1]
      const U32x2 kc = threefry_split(nk.x, nk.y, 0u, 2u);
      const U32x2 kn = threefry_split(nk.x, nk.y, 1u, 2u);
      // gather through LDS, DCH dimensions per pass.  The first pass is staged BEFORE the ancestors are drawn: those
      // DCH x PPT state registers are then dead while the CDF, the uniforms and the searches run.
      // Tile layout: one record of DCH floats per particle, particle i = tid * PPT + p at record p * NT + tid -- the
      // lanes of a wave write consecutive records (16-byte stores, every bank once per 8 lanes), and a drawn ancestor is
      // fetched as DCH / 4 16-byte reads instead of DCH dword reads DCH banks apart.  (Dimensions that are not multiples
      // of four keep the dimension-major dword layout.)
      constexpr bool VEC = (DCH % 4 == 0) && (N % 4 == 0);
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      auto put = [&](int d0) __attribute__((always_inline)) {
        if constexpr (VEC) {
          f32x4* t4 = reinterpret_cast<f32x4*>(tile);
          BF_UNROLL for (int p = 0; p < PPT; ++p) BF_UNROLL for (int v = 0; v < DCH / 4; ++v)
              if (d0 + 4 * v < N)
                t4[(p * NT + tid) * (DCH / 4) + v] = f32x4{x[p][d0 + 4 * v], x[p][d0 + 4 * v + 1], x[p][d0 + 4 * v + 2], x[p][d0 + 4 * v + 3]};
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
                  const f32x4 t = t4[rec * (DCH / 4) + v];
                  x[p][d0 + 4 * v] = t.x; x[p][d0 + 4 * v + 1] = t.y; x[p][d0 + 4 * v + 2] = t.z; x[p][d0 + 4 * v + 3] = t.w;
                }
          }
        } else {
          BF_UNROLL for (int p = 0; p < PPT; ++p) BF_UNROLL for (int d = 0; d < DCH; ++d)
              if (d0 + d < N) x[p][d0 + d] = tile[d * CAP + anc[p]];
        }
      };
      lds_barrier();
      put(0);
      resample_indices<PPT, NW>(wn, valid, NP, kc, resampler, cdf, red, anc);  // (its barriers publish the tile)
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

    // ---- emit
    BF_UNROLL for (int p = 0; p < PPT; ++p) if (valid[p]) {
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
      lds_barrier();
      if (lane == 63) BF_UNROLL for (int d = 0; d < N; ++d) tile[wave * N + d] = part[d];
      lds_barrier();
      if (tid < N) {
        float s = 0.f;
        for (int wv = 0; wv < NW; ++wv) s += tile[wv * N + tid];
        out.mean[(b * T + t) * N + tid] = s;
      }
    }
    if (tid == 0) {
      if (out.ess) out.ess[b * T + t] = ess;
      if (out.logz) out.logz[b * T + t] = mx + canon_log(tot);
      if (out.resampled) out.resampled[b * T + t] = do_resample ? 1.0f : 0.0f;
    }
  }

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

// ---------------------------------------------------------------------------------------
template <int N, int DQ, int M, int PPT, int NW>
static inline int launch_bpf_cfg(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B,
                          long long T, int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr,
                          const BpfOut& out, hipStream_t stream) {
  constexpr int CAP = 64 * NW * PPT;
  constexpr int DCH = (PPT >= 16) ? 1 : ((N >= 8) ? 8 : N);
  const size_t lds_bytes = sizeof(float) * (size_t)(CAP + 64 + CAP * DCH);
  if (lds_bytes > 160 * 1024) return set_error(BF_EUNSUPPORTED, "particle tile exceeds the 160 KiB LDS");
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  auto kern = bpf_scan_kernel<N, DQ, M, PPT, NW>;
  if (lds_bytes > 64 * 1024)
    BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64 * NW), lds_bytes, stream, d_mdl, yv, (u && u->ptr) ? u->ptr : nullptr,
                     u ? u->sB : 0, u ? u->sT : 0, cr, out, B, T, NP, ess, resampler, key[0], key[1]);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

extern std::atomic<int> g_bpf_variant;   // tuning hook (bf_set_option "bpf_variant")
extern std::atomic<int> g_bpf_hbm_mode;  // bf_set_option "bpf_hbm_mode": 0 = choose, 1 = workgroup per trajectory, 2 = per chunk

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
  // smallest compiled particle capacity that holds NP
  if (spread) rc = launch_bpf_hbm_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  else if (NP <= 64) rc = launch_bpf_cfg<N, DQ, M, 1, 1>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  else if (NP <= 256) rc = launch_bpf_cfg<N, DQ, M, 1, 4>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  else if (NP <= 1024) rc = launch_bpf_cfg<N, DQ, M, 1, 16>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  else if (NP <= 4096) {
    // two geometries for the largest capacity: 1024 threads x 4 particles (128-VGPR budget, variant 0, the default) or
    // 512 threads x 8 particles (256-VGPR budget, variant 1).  Measured at cfg4's shape (B = 1024, T = 100, every step
    // resamples): 1024 x 4 = 41.9 ms, 512 x 8 = 50.8 ms; without the per-particle re-read of the model (see the kernel)
    // 52.3 / 44.4 ms -- the 128-VGPR geometry was the one that spilled.  Of the 42 ms, 29 are propagation + weights
    // (ess_threshold = 0) and 13 the resampling branch, two thirds of it the two gather passes through the LDS tile
    // (scripts/bpf_probe.py).  A rolled particle loop (registers rotated by one particle per trip, 16 KB of code instead
    // of 130 KB) is slower, 44 / 78 ms: the unrolled trips overlap each other's dependent Threefry / erf_inv chains.
    if (g_bpf_variant == 1) rc = launch_bpf_cfg<N, DQ, M, 8, 8>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
    else rc = launch_bpf_cfg<N, DQ, M, 4, 16>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  }
  else if (NP <= 16384 && N <= 4 && DQ <= 4) {
    // small states: 16 particles per thread still fit the registers (1024 threads x 16; the gather goes one
    // state dimension at a time so that CDF + tile stay within the LDS)
    if constexpr (N <= 4 && DQ <= 4) rc = launch_bpf_cfg<N, DQ, M, 16, 16>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  }
  else rc = launch_bpf_hbm_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);  // particles in HBM (bpf_big.hpp, bpf_wide.hpp)
  return rc;
}

}  // namespace bf
