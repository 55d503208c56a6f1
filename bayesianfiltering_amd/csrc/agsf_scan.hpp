// agsf_scan: batched "speedy" augmented Gaussian-sum filter.
//
// Replaces the lax.scan body of speedy_augmented_gaussian_sum_filter (gaussfiltax/inference.py:621-812).
// Per step every carried component i0 (N0 of them) is branched into N1 z-samples drawn from
// N(m, P - Delta), Delta = opt_args[0] P (:665-688); each is pushed through the extended-Kalman _predict
// with covariance Delta (:695-698); every prediction is branched into N2 s-samples from
// N(m-, P- - Lambda), Lambda = opt_args[1] P- (:711-726); each is updated by _condition_on with covariance
// Lambda (:735-737); the N0 N1 N2 leaves are weighted (:738-743) and N0 of them are drawn with
// jr.choice under the fixed key PRNGKey(0) (:760) to become the next carry with weights 1 / N0.
// rng_key is never advanced by the reference, so the two arrays of normals are the same at every step
// (:672, :716): each leaf draws its two n-vectors once, before the time loop.
//
// Mapping (gfx950).  One lane per leaf (i0, i1, i2); the MP = next_pow2(N0 N1 N2) <= 64 lanes of a
// trajectory are consecutive lanes of one wave, 256 / MP trajectories per workgroup.  A lane redoes its
// parents' work (the Cholesky factor of (1 - a0) P, the prediction of its (i0, i1) node) instead of
// communicating: the tree is shallow and the algebra is n^3 with n <= 8.  Cross-lane steps: max / sum of
// the leaf weights (xor butterfly, adjacent-pair tree), the cumulative sum for jr.choice in
// lax.associative_scan order (segmented Brent-Kung with __shfl_up), the inverse-CDF search and the
// gather of the drawn leaves through LDS.
#pragma once
#ifndef BF_JIT
#include <cstring>
#endif
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "scan_common.hpp"
#include "bf_rng.hpp"
#include "models.hpp"
#include "gsf_scan.hpp"   // fill_model: EkfModel from the C-ABI struct
#include "ugsf_scan.hpp"  // unscented node operations (speedy_unscented_agsf, unscented_agsf)
#include "agsf_geom.hpp"

namespace bf {

// jnp.linalg.cholesky as the reference's CPU runs evaluate it: the input is symmetrised ((A + A^T) / 2), and a failed
// factorisation (LAPACK potrf: a pivot <= 0 or NaN) returns an all-NaN factor
template <int N>
__device__ __forceinline__ void chol_lower(const float* A, float* L) {
  BF_UNROLL for (int i = 0; i < N * N; ++i) L[i] = 0.f;
  bool bad = false;
  BF_UNROLL for (int j = 0; j < N; ++j) {
    float d = A[j * N + j];
    BF_UNROLL for (int k = 0; k < j; ++k) d = fmaf(-L[j * N + k], L[j * N + k], d);
    bad |= !(d > 0.f);
    d = fast_sqrt(d);   // single instructions (1 ulp), as in the Kalman kernels' factorizations (kf_math.hpp)
    L[j * N + j] = d;
    const float inv = fast_rcp(d);
    BF_UNROLL for (int i = j + 1; i < N; ++i) {
      float s = 0.5f * (A[i * N + j] + A[j * N + i]);
      BF_UNROLL for (int k = 0; k < j; ++k) s = fmaf(-L[i * N + k], L[j * N + k], s);
      L[i * N + j] = s * inv;
    }
  }
  if (bad) {
    BF_UNROLL for (int i = 0; i < N * N; ++i) L[i] = __builtin_nanf("");
  }
}

// optimal_resampling of Fearnhead & Clifford as written in gaussfiltax/utils.py:216-244, for the M <= 64 weights
// held one per lane by the MP = next_pow2(M) lanes of a trajectory (lane l: w, 0 beyond M).  Returns, in lanes
// j < N, the index of the particle that becomes output j and its normalised weight.
//   sorted_weights, sorted_idx = sort / argsort(weights)            stable: bitonic network on (w, index) pairs
//   ps[i] = cumsum(sorted_weights)[M-N+i] / (i+1), flipped; L, p     (:226-232; running sums in index order)
//   res_idx = jr.choice(key, M, (M,), p = normalised weights below p) (:235-237; CDF in associative_scan order)
//   final_idx / final_weights, last N entries                        (:240-243)
__device__ __forceinline__ void optimal_resampling_lanes(float w, int l, int MP, int M, int N, uint32_t k0, uint32_t k1,
                                                         int& idx_out, float& w_out) {
  const int lane = threadIdx.x & 63;
  const int seg0 = lane - l;  // first lane of the trajectory's segment
  // ---- stable ascending sort of (w, l); padding lanes carry +inf and end up beyond M
  float sv = l < M ? w : __builtin_inff();
  int si = l;
  for (int k = 2; k <= MP; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const float ov = __shfl_xor(sv, j, 64);
      const int oi = __shfl_xor(si, j, 64);
      const bool up = (l & k) == 0;         // ascending block
      const bool lower = (l & j) == 0;      // this lane keeps the smaller of the pair in an ascending block
      const bool other_less = (ov < sv) || (ov == sv && oi < si);
      const bool take = (lower == up) ? other_less : !other_less;
      sv = take ? ov : sv;
      si = take ? oi : si;
    }
  // ---- running sums of the sorted weights in index order: cum[s] = ((sw0 + sw1) + ...) + sw_s
  float cum = 0.f;
  for (int c = 0; c < M; ++c) {
    const float vc = __shfl(sv, seg0 + c, 64);
    cum = (c <= l) ? cum + vc : cum;
  }
  // ---- threshold: lane s <-> ind = M - 1 - s in [1, N - 1]; ps = cum[s] / (N - ind); bounds (sw[s], sw[s + 1])
  const int ind = M - 1 - l;
  const float sw_next = __shfl_down(sv, 1, 64);
  const bool cand = ind >= 1 && ind <= N - 1 && l < M;
  const float psv = cum / (float)(N - ind);
  const bool pred = cand && (sv < psv) && (psv < sw_next);
  int Lsum = pred ? ind : 0;
  for (int off = 1; off < MP; off <<= 1) Lsum += __shfl_xor(Lsum, off, 64);
  const int Ll = Lsum >= 1 && Lsum <= N - 1 ? Lsum : 1;
  const float p_at = __shfl(psv, seg0 + (M - 1 - Ll), 64);
  const float p = Lsum == 0 ? 1.0f / (float)N : p_at;
  // ---- resample among the weights below p
  const bool below = l < M && sv < p;
  float rw = below ? sv : 0.f;
  float tot = rw;
  for (int off = 1; off < MP; off <<= 1) tot += __shfl_xor(tot, off, 64);   // adjacent-pair tree
  rw = rw / tot;
  float c = rw;                                                              // cumsum, lax.associative_scan order
  for (int d = 0; (1 << d) < MP; ++d) {
    const float o = __shfl_up(c, 1 << d, 64);
    if (((l + 1) & ((2 << d) - 1)) == 0) c += o;
  }
  for (int d = 5; d >= 1; --d)
    if ((1 << d) <= MP) {
      const float o = __shfl_up(c, 1 << (d - 1), 64);
      if (l >= (1 << d) && ((l + 1) & ((1 << d) - 1)) == (1 << (d - 1))) c += o;
    }
  const float ctot = __shfl(c, seg0 + M - 1, 64);
  const float u = bits_to_unit(threefry_bits(k0, k1, (uint32_t)(l < M ? l : 0), (uint32_t)M));
  const float r = ctot * (1.0f - u);
  int lo = 0, hi = M;  // first index with cdf[idx] >= r: every lane walks the same bounded search on shuffled values
  for (int it = 0; it < 7; ++it) {
    const int mid = (lo + hi) >> 1;
    const float cm = __shfl(c, seg0 + (mid < M ? mid : M - 1), 64);
    const bool go = lo < hi;
    if (go && cm < r) lo = mid + 1;
    else if (go) hi = mid;
  }
  // (no weight below p: 0 / 0 weights, a NaN draw -- every comparison of the search fails and the clamped result
  // is the last index)
  const int ridx = (r != r) ? M - 1 : (lo < M - 1 ? lo : M - 1);
  const int unsort = __shfl(si, seg0 + ridx, 64);
  const int fidx = below ? unsort : si;
  const float fw = below ? p : sv;
  // ---- the last N sorted positions are the output
  const bool top = l >= M - N && l < M;
  float ftot = top ? fw : 0.f;
  for (int off = 1; off < MP; off <<= 1) ftot += __shfl_xor(ftot, off, 64);
  const int srcl = M - N + (l < N ? l : 0);
  idx_out = __shfl(fidx, seg0 + srcl, 64);
  w_out = __shfl(fw, seg0 + srcl, 64) / ftot;
}


// ---- the same building blocks for one trajectory per workgroup of NW waves (thread l = threadIdx.x holds element l of
// MP = 64 NW; `red` is 64 floats of LDS scratch: [0, 16) reductions, [16, 32) scan totals, [32, 48) integer sums)
template <int NW, class OP>
__device__ __forceinline__ float block_tree_reduce(float v, float* red, OP op) {  // adjacent-pair tree over the 64 NW elements
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
  lds_barrier();
  if (lane == 0) red[wave] = v;
  lds_barrier();
  float r = red[lane < NW ? lane : 0];
  for (int off = 1; off < NW; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
  return __shfl(r, 0, 64);
}
// inclusive cumulative sum in lax.associative_scan order (Brent-Kung) over the 64 NW elements: wave up-sweep, scan of
// the wave totals, wave down-sweep
template <int NW>
__device__ __forceinline__ float block_cumsum_assoc(float c, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  BF_UNROLL for (int d = 0; d < 6; ++d) {
    const float o = __shfl_up(c, 1 << d, 64);
    if (((lane + 1) & ((2 << d) - 1)) == 0) c += o;
  }
  lds_barrier();
  if (lane == 63) red[16 + wave] = c;
  lds_barrier();
  float r = (lane < NW) ? red[16 + lane] : 0.f;
  BF_UNROLL for (int d = 0; (1 << d) < NW; ++d) {
    const float o = __shfl_up(r, 1 << d, 64);
    if (lane < NW && ((lane + 1) & ((2 << d) - 1)) == 0) r += o;
  }
  BF_UNROLL for (int d = 4; d >= 1; --d) {
    if ((1 << d) <= NW) {
      const float o = __shfl_up(r, 1 << (d - 1), 64);
      if (lane < NW && lane >= (1 << d) && ((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) r += o;
    }
  }
  const float mine = __shfl(r, wave, 64);
  const float prev = __shfl(r, wave > 0 ? wave - 1 : 0, 64);
  const float excl_wave = wave > 0 ? prev : 0.f;
  if (lane == 63) c = mine;
  BF_UNROLL for (int d = 6; d >= 1; --d) {
    const float o = __shfl_up(c, 1 << (d - 1), 64);
    if (((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) c += (lane >= (1 << (d - 1))) ? o : excl_wave;
  }
  return c;
}

// optimal_resampling (utils.py:216-244) for M <= 64 NW weights held one per thread by a whole workgroup -- the reference's
// own test runs augmented_gaussian_sum_filter_optimal on a [5, 5, 5] tree, 125 leaves (docs/tests/test_inference.py:89-92).
// Same steps and the same arithmetic orders as optimal_resampling_lanes; the sort's far exchanges, the running sums, the
// table look-ups and the gathers go through `scratch` (4 * 64 NW floats of LDS).  Every thread of the workgroup calls it.
template <int NW>
__device__ __forceinline__ void optimal_resampling_block(float w, int M, int N, uint32_t k0, uint32_t k1, float* scratch, float* red,
                                                         int& idx_out, float& w_out) {
  constexpr int MP = 64 * NW;
  const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
  float* s_v = scratch;                                  // sorted weights; later the final indices
  int* s_i = reinterpret_cast<int*>(scratch + MP);       // sorted indices
  float* s_a = scratch + 2 * MP;                         // thresholds ps; later the final weights
  float* s_b = scratch + 3 * MP;                         // cumulative resampling weights
  // ---- stable ascending sort of (w, l): bitonic network, partners within a wave by shuffle, beyond it through LDS
  float sv = l < M ? w : __builtin_inff();
  int si = l;
  for (int k = 2; k <= MP; k <<= 1)
    for (int j = k >> 1; j >= 1; j >>= 1) {
      float ov;
      int oi;
      if (j < 64) {
        ov = __shfl_xor(sv, j, 64);
        oi = __shfl_xor(si, j, 64);
      } else {
        lds_barrier();
        s_v[l] = sv;
        s_i[l] = si;
        lds_barrier();
        ov = s_v[l ^ j];
        oi = s_i[l ^ j];
      }
      const bool up = (l & k) == 0;
      const bool lower = (l & j) == 0;
      const bool other_less = (ov < sv) || (ov == sv && oi < si);
      const bool take = (lower == up) ? other_less : !other_less;
      sv = take ? ov : sv;
      si = take ? oi : si;
    }
  lds_barrier();
  s_v[l] = sv;
  s_i[l] = si;
  lds_barrier();
  // ---- running sums in index order (a broadcast read per term)
  float cum = 0.f;
  for (int c = 0; c < M; ++c) {
    const float vc = s_v[c];
    cum = (c <= l) ? cum + vc : cum;
  }
  // ---- threshold
  const int ind = M - 1 - l;
  const float sw_next = s_v[l + 1 < MP ? l + 1 : l];
  const bool cand = ind >= 1 && ind <= N - 1 && l < M;
  const float psv = cum / (float)(N - ind);
  const bool pred = cand && (sv < psv) && (psv < sw_next);
  s_a[l] = psv;
  int Lsum = pred ? ind : 0;
  for (int off = 1; off < 64; off <<= 1) Lsum += __shfl_xor(Lsum, off, 64);
  int* red_i = reinterpret_cast<int*>(red + 32);
  lds_barrier();
  if (lane == 0) red_i[wave] = Lsum;
  lds_barrier();
  Lsum = 0;
  BF_UNROLL for (int q = 0; q < NW; ++q) Lsum += red_i[q];
  const int Ll = Lsum >= 1 && Lsum <= N - 1 ? Lsum : 1;
  const float p_at = s_a[M - 1 - Ll];
  const float p = Lsum == 0 ? 1.0f / (float)N : p_at;
  // ---- resample among the weights below p
  const bool below = l < M && sv < p;
  float rw = below ? sv : 0.f;
  const float tot = block_tree_reduce<NW>(rw, red, [](float a, float b) { return a + b; });
  rw = rw / tot;
  const float c = block_cumsum_assoc<NW>(rw, red);
  s_b[l] = c;
  lds_barrier();
  const float ctot = s_b[M - 1];
  const float u = bits_to_unit(threefry_bits(k0, k1, (uint32_t)(l < M ? l : 0), (uint32_t)M));
  const float r = ctot * (1.0f - u);
  int lo = 0, hi = M;  // first index with cdf[idx] >= r
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (s_b[mid] < r) lo = mid + 1; else hi = mid;
  }
  const int ridx = (r != r) ? M - 1 : (lo < M - 1 ? lo : M - 1);
  const int unsort = s_i[ridx];
  const int fidx = below ? unsort : si;
  const float fw = below ? p : sv;
  // ---- the last N sorted positions are the output
  const bool top = l >= M - N && l < M;
  const float ftot = block_tree_reduce<NW>(top ? fw : 0.f, red, [](float a, float b) { return a + b; });
  lds_barrier();
  reinterpret_cast<int*>(s_v)[l] = fidx;
  s_a[l] = fw;
  lds_barrier();
  const int srcl = M - N + (l < N ? l : 0);
  idx_out = reinterpret_cast<int*>(s_v)[srcl];
  w_out = s_a[srcl] / ftot;
}

// What a tree node does with its Gaussian: the extended-Kalman pair _predict / _condition_on
// (inference.py:51-105) or the unscented pair (:146-174, :198-224) of speedy_unscented_agsf (:966-1156).
#ifndef BF_JIT
template <int N, int M>
struct EkfNodes {
  using Arg = EkfModel<N, M>;  // by value, in the kernel arguments
  // tq / tr: this step's F_q Q_t F_q^T / H_r R_t H_r^T when the covariances are (T, d, d) (inference.py:658-661), else NULL
  static constexpr int TVQ = N * N, TVR = M * M;  // floats per step in the tables
  static __device__ __forceinline__ void predict(const Arg& mdl, float* m, float* P, float u0, const float* tq) {
    float F[N * N], fx[N];
    dyn_linearize<N, M>(mdl, m, u0, F, fx);
    predict_cov<N>(F, tq ? tq : mdl.GQG, P);  // F P F^T + F_q Q F_q^T
    BF_UNROLL for (int i = 0; i < N; ++i) m[i] = fx[i];
  }
  static __device__ __forceinline__ float condition(const Arg& mdl, float* m, float* P, const float* yv, float u0, const float* tr) {
    float H[M * N], hx[M], HrRHr[M * M], v[M];
    emi_linearize<N, M>(mdl, m, u0, H, hx, HrRHr);
    if (tr) BF_UNROLL for (int i = 0; i < M * M; ++i) HrRHr[i] = tr[i];
    BF_UNROLL for (int a = 0; a < M; ++a) v[a] = yv[a] - hx[a];
    return condition_on<N, M>(H, HrRHr, v, m, P);
  }
};
#endif  // BF_JIT
template <int N, int DQ, int M, int DR, class SP = SpecRuntime>
struct UkfNodes {
  using Arg = const UkfModel<N, DQ, M, DR>*;  // device-resident
  // tq / tr: sqrtm(Q_t) / sqrtm(R_t) of the step, or NULL
  static constexpr int TVQ = DQ * DQ, TVR = DR * DR;
  static __device__ __forceinline__ void predict(Arg mdl, float* m, float* P, float u0, const float* tq) {
    ukf_predict<SP>(*mdl, m, P, u0, tq ? tq : mdl->sQ);
  }
  static __device__ __forceinline__ float condition(Arg mdl, float* m, float* P, const float* yv, float u0, const float* tr) {
    return ukf_condition_on<SP>(*mdl, m, P, yv, u0, tr ? tr : mdl->sR);
  }
};



// NW = 1: the MP <= 64 leaves of a trajectory are lanes of one wave, 256 / MP trajectories per 256-thread workgroup.
// NW > 1: one trajectory per workgroup of 64 NW threads (MP = 64 NW leaves, e.g. the [100, 2, 2] tree of
// BOT_Experiment_script.py:118); reductions and the cumulative sum continue across waves through LDS.
template <int N, int M, class NODES, int NW>
__device__ __forceinline__ void
agsf_scan_body(typename NODES::Arg mdl, CView y, UView uin, CarryView carry, AgsfOut out, long long B, long long T, int N0, int N1,
               int N2, int MP, float a0, float a1, uint32_t key0, uint32_t key1, int variant, int carry_records,
               const float* __restrict__ tvq, const float* __restrict__ tvr) {
#pragma clang fp contract(fast)   // (stated, not inherited: a build compiled at run time sets contraction off for the caller's functions)
  constexpr int EP = N * N;
  constexpr int REC = N + EP;  // one component record in LDS: mean, covariance
  constexpr int NT = NW == 1 ? 256 : 64 * NW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpb = NT / MP;
  const int slot = tid / MP;       // trajectory slot in the workgroup
  const int l = tid % MP;          // leaf index
  const int Mleaf = N0 * N1 * N2;
  const bool leaf_ok = l < Mleaf;
  const int lc = leaf_ok ? l : 0;  // padding lanes shadow leaf 0 (weight 0, never drawn)
  const int i0 = lc / (N1 * N2), i1 = (lc / N2) % N1, i2 = lc % N2;
  const long long b_raw = (long long)blockIdx.x * tpb + slot;
  const bool traj_ok = b_raw < B;
  const long long b = traj_ok ? b_raw : B - 1;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* leafbuf = lds;                          // [NT][REC]            updated mean / covariance of every leaf
  float* carrybuf = lds + NT * REC;              // [carry_records][REC] carried components, slot-major: [slot * MP + i0]
  float* cdfbuf = carrybuf + carry_records * REC;  // [NT]               cumulative leaf weights
  float* wbuf = cdfbuf + NT;                     // [carry_records]      weights of the carried components
  float* red = wbuf + carry_records;             // [64]                 cross-wave scratch (NW > 1)
  float* optbuf = red + 64;                      // [4 NT]               optimal_resampling_block's tables (NW > 1)

  // ---- the two standard-normal vectors of this leaf (same at every step: the reference's key is never advanced)
  float ez[N], es[N];
  if (variant == 0) {
    const U32x2 kz = threefry_split(key0, key1, 0u, 2u);   // key, subkey = jr.split(rng_key)          :672
    const U32x2 ks = threefry_split(kz.x, kz.y, 0u, 2u);   // key, _ = jr.split(key)                    :716
    const uint32_t cz = (uint32_t)(N0 * N * N1), cs = (uint32_t)(N0 * N1 * N * N2);
    BF_UNROLL for (int d = 0; d < N; ++d) {
      ez[d] = bits_to_normal(threefry_bits(kz.x, kz.y, (uint32_t)((i0 * N + d) * N1 + i1), cz));            // (N0, n, N1)
      es[d] = bits_to_normal(threefry_bits(ks.x, ks.y, (uint32_t)(((i0 * N1 + i1) * N + d) * N2 + i2), cs));  // (N0 N1, n, N2)
    }
  } else {
    // augmented_gaussian_sum_filter (inference.py:458-620): the branches come from containers._branches_from_tree1/2
    // (containers.py:63-140): keys = split(subkey, #nodes), node j draws jr.multivariate_normal(keys[j], mean,
    // cov - split_cov, (num,)) = mean + chol @ normal(keys[j], (num, n))
    const U32x2 k0s = threefry_split(key0, key1, 0u, 2u);      // key, subkey = jr.split(rng_key)       :519
    const U32x2 sub1 = threefry_split(key0, key1, 1u, 2u);
    const U32x2 sub2 = threefry_split(k0s.x, k0s.y, 1u, 2u);   // key, subkey = jr.split(key)           :545
    const U32x2 kn1 = threefry_split(sub1.x, sub1.y, (uint32_t)i0, (uint32_t)N0);
    const U32x2 kn2 = threefry_split(sub2.x, sub2.y, (uint32_t)(i0 * N1 + i1), (uint32_t)(N0 * N1));
    BF_UNROLL for (int d = 0; d < N; ++d) {
      ez[d] = bits_to_normal(threefry_bits(kn1.x, kn1.y, (uint32_t)(i1 * N + d), (uint32_t)(N1 * N)));   // (N1, n)
      es[d] = bits_to_normal(threefry_bits(kn2.x, kn2.y, (uint32_t)(i2 * N + d), (uint32_t)(N2 * N)));   // (N2, n)
    }
  }
  // uniform of the resampling draw this lane performs (lanes l < N0): uniform(PRNGKey(0), (N0,))[l]     :760
  const float udraw = bits_to_unit(threefry_bits(0u, 0u, (uint32_t)(l < N0 ? l : 0), (uint32_t)N0));

  // ---- carry -> LDS
  if (l < N0) {
    float* rec = carrybuf + (slot * MP + l) * REC;
    BF_UNROLL for (int i = 0; i < N; ++i) rec[i] = carry.m_in[(b * N0 + l) * N + i];
    BF_UNROLL for (int i = 0; i < EP; ++i) rec[N + i] = carry.P_in[(b * N0 + l) * EP + i];
  }
  float wpar = carry.w_in ? carry.w_in[b * N0 + i0] : 1.0f / (float)N0;  // weight of the parent component
  float wmine = (l < N0) ? (carry.w_in ? carry.w_in[b * N0 + l] : 1.0f / (float)N0) : 0.f;  // of component l (carry out)
  lds_barrier();

  auto seg_reduce = [&](float v, auto op) {  // over the MP leaves of the trajectory, adjacent-pair tree
    if constexpr (NW == 1) {
      for (int off = 1; off < MP; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    } else {
      for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
      lds_barrier();
      if (lane == 0) red[wave] = v;
      lds_barrier();
      float r = red[lane < NW ? lane : 0];
      for (int off = 1; off < NW; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
      v = __shfl(r, 0, 64);
    }
    return v;
  };

  // observation and input of step t + 1 are fetched while step t runs (a load issued at the top of its own step would put
  // an HBM round trip on every step's critical path)
  float ynext[M], unext;
  BF_UNROLL for (int a = 0; a < M; ++a) ynext[a] = y.p[b * y.sB + a * y.sE];
  unext = uin.p ? uin.p[b * uin.sB + 0 * uin.sT] : 0.f;
  for (long long t = 0; t < T; ++t) {
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = ynext[a];
    const float u0 = unext;
    {
      const long long tn = t + 1 < T ? t + 1 : t;
      BF_UNROLL for (int a = 0; a < M; ++a) ynext[a] = y.p[b * y.sB + tn * y.sT + a * y.sE];
      unext = uin.p ? uin.p[b * uin.sB + tn * uin.sT] : 0.f;
    }

    // ---- z-sample of the (i0, i1) node and its prediction (:675-698)
    float mz[N], P[EP];
    {
      const float* rec = carrybuf + (slot * MP + i0) * REC;
      float Pk[EP], Lz[EP];
      BF_UNROLL for (int i = 0; i < EP; ++i) Pk[i] = rec[N + i];
      float Dl[EP], Az[EP];
      BF_UNROLL for (int i = 0; i < EP; ++i) {
        Dl[i] = a0 * Pk[i];       // Delta = opt_args[0] * P                                       :665
        Az[i] = Pk[i] - Dl[i];    // filtered_covs - Deltas                                        :675
      }
      chol_lower<N>(Az, Lz);
      BF_UNROLL for (int i = 0; i < N; ++i) {
        float s = 0.f;
        BF_UNROLL for (int c = 0; c <= i; ++c) s = fmaf(Lz[i * N + c], ez[c], s);
        mz[i] = rec[i] + s;       // z = m + chol(P - Delta) eps
        if (variant != 0 && mz[i] != mz[i]) mz[i] = rec[i];  // jnp.where(isnan(new_means), mean, new_means)  containers.py:84
      }
      NODES::predict(mdl, mz, Dl, u0, tvq ? tvq + t * NODES::TVQ : nullptr);  // the node (z, Delta) through the dynamics
      BF_UNROLL for (int i = 0; i < EP; ++i) P[i] = Dl[i];
    }
    // ---- s-sample of the leaf and its update (:711-737)
    float ll;
    {
      float Lam[EP], As[EP], Ls[EP];
      BF_UNROLL for (int i = 0; i < EP; ++i) {
        Lam[i] = a1 * P[i];       // Lambda = opt_args[1] * P-                                     :711
        As[i] = P[i] - Lam[i];
      }
      chol_lower<N>(As, Ls);
      float ms[N];
      BF_UNROLL for (int i = 0; i < N; ++i) {
        float s = 0.f;
        BF_UNROLL for (int c = 0; c <= i; ++c) s = fmaf(Ls[i * N + c], es[c], s);
        ms[i] = mz[i] + s;
        if (variant != 0 && ms[i] != ms[i]) ms[i] = mz[i];        // containers.py:121
      }
      ll = NODES::condition(mdl, ms, Lam, yv, u0, tvr ? tvr + t * NODES::TVR : nullptr);  // the leaf (s, Lambda) conditioned on y
      BF_UNROLL for (int i = 0; i < N; ++i) mz[i] = ms[i];
      BF_UNROLL for (int i = 0; i < EP; ++i) P[i] = Lam[i];
    }
    // ---- leaf weights (:738-743): carried weight / N1 / N2, times exp(ll - max), normalised
    const float wleaf = (wpar / (float)N1) / (float)N2;
    const float llm = leaf_ok ? ll : -__builtin_inff();
    const float mx = seg_reduce(llm, [](float a, float c) { return (a != a || c != c) ? __builtin_nanf("") : fmaxf(a, c); });
    const float e = leaf_ok ? expf(ll - mx) * wleaf : 0.f;
    const float tot = seg_reduce(e, [](float a, float c) { return a + c; });
    const float w = leaf_ok ? e / tot : 0.f;

    // ---- jr.choice(PRNGKey(0), arange(M), (N0,), p = w) (:760): cumsum in associative_scan order, inverse-CDF search
    float c = w;
    if constexpr (NW == 1) {
      for (int d = 0; (1 << d) < MP; ++d) {        // up-sweep
        const float o = __shfl_up(c, 1 << d, 64);
        if (((l + 1) & ((2 << d) - 1)) == 0) c += o;
      }
      for (int d = 5; d >= 1; --d) {               // down-sweep
        if ((1 << d) <= MP) {
          const float o = __shfl_up(c, 1 << (d - 1), 64);
          if (l >= (1 << d) && ((l + 1) & ((1 << d) - 1)) == (1 << (d - 1))) c += o;
        }
      }
    } else {
      c = block_cumsum_assoc<NW>(c, red);  // the same Brent-Kung order over 64 NW leaves
    }
    lds_barrier();  // previous step's readers of leafbuf / cdfbuf are done
    cdfbuf[tid] = c;
    {
      float* rec = leafbuf + tid * REC;
      BF_UNROLL for (int i = 0; i < N; ++i) rec[i] = mz[i];
      BF_UNROLL for (int i = 0; i < EP; ++i) rec[N + i] = P[i];
    }
    lds_barrier();
    int idx = 0;
    float wnew = 1.0f / (float)N0;  // weights = ones / N0                                          :765
    if (NW == 1 && variant == 2) {
      // augmented_gaussian_sum_filter_optimal (:1157-1300): utils.optimal_resampling(weights, N0, key) with the
      // key left by the two splits (:1203, :1229); the drawn components keep unequal weights
      const U32x2 kz = threefry_split(key0, key1, 0u, 2u);
      const U32x2 ko = threefry_split(kz.x, kz.y, 0u, 2u);
      optimal_resampling_lanes(w, l, MP, Mleaf, N0, ko.x, ko.y, idx, wnew);
    } else if (NW > 1 && variant == 2) {
      const U32x2 kz = threefry_split(key0, key1, 0u, 2u);
      const U32x2 ko = threefry_split(kz.x, kz.y, 0u, 2u);
      if constexpr (NW > 1) optimal_resampling_block<NW>(w, Mleaf, N0, ko.x, ko.y, optbuf, red, idx, wnew);
    } else if (l < N0) {
      const float* cd = cdfbuf + slot * MP;
      const float r = cd[Mleaf - 1] * (1.0f - udraw);
      int lo = 0, hi = Mleaf;  // first index with cdf[idx] >= r (searchsorted side='left')
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cd[mid] < r) lo = mid + 1; else hi = mid;
      }
      idx = lo < Mleaf - 1 ? lo : Mleaf - 1;
    }
    if (l < N0) {
      const float* src = leafbuf + (slot * MP + idx) * REC;
      float* dst = carrybuf + (slot * MP + l) * REC;
      // drawn leaf -> component l of the next carry, and out
      BF_UNROLL for (int i = 0; i < REC; ++i) dst[i] = src[i];
      wbuf[slot * MP + l] = wnew;
      wmine = wnew;
      if (traj_ok) {
        if (out.m.p) BF_UNROLL for (int i = 0; i < N; ++i) out.m.p[b * out.m.sB + l * out.m.sK + t * out.m.sT + i * out.m.sE] = src[i];
        if (out.P.p) BF_UNROLL for (int i = 0; i < EP; ++i) out.P.p[b * out.P.sB + l * out.P.sK + t * out.P.sT + i * out.P.sE] = src[N + i];
        if (out.w.p) out.w.p[b * out.w.sB + l * out.w.sK + t * out.w.sT] = wnew;
        if (out.anc) out.anc[(b * T + t) * N0 + l] = idx;
      }
    }
    lds_barrier();
    wpar = wbuf[slot * MP + i0];
  }

  if (traj_ok && l < N0) {
    const float* rec = carrybuf + (slot * MP + l) * REC;
    if (carry.m_out) BF_UNROLL for (int i = 0; i < N; ++i) carry.m_out[(b * N0 + l) * N + i] = rec[i];
    if (carry.P_out) BF_UNROLL for (int i = 0; i < EP; ++i) carry.P_out[(b * N0 + l) * EP + i] = rec[N + i];
    if (carry.w_out) carry.w_out[b * N0 + l] = wmine;
  }
}

template <int N, int M, class NODES, int NW>
__global__ void __launch_bounds__(NW == 1 ? 256 : 64 * NW)
agsf_scan_kernel(typename NODES::Arg mdl, CView y, UView uin, CarryView carry, AgsfOut out, long long B, long long T, int N0, int N1,
                 int N2, int MP, float a0, float a1, uint32_t key0, uint32_t key1, int variant, int carry_records,
                 const float* __restrict__ tvq, const float* __restrict__ tvr) {
  agsf_scan_body<N, M, NODES, NW>(mdl, y, uin, carry, out, B, T, N0, N1, N2, MP, a0, a1, key0, key1, variant, carry_records, tvq, tvr);
}

#ifndef BF_JIT
template <int N, int M, class NODES, int NW>
static inline int launch_agsf_geom(typename NODES::Arg arg, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                                   const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry,
                                   const bf_out_desc* out, int* d_leaf_idx, int variant, int MP, const float* d_tvq, const float* d_tvr,
                                   hipStream_t stream) {
  constexpr int NT = NW == 1 ? 256 : 64 * NW;
  const int carry_records = NW == 1 ? 256 : ((nc[0] + 3) & ~3);
  const size_t lds_bytes = agsf_lds_bytes(N, NW, nc[0]);
  if (lds_bytes > 160 * 1024)
    return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: %d leaves and %d components of dimension %d exceed the 160 KiB LDS",
                     nc[0] * nc[1] * nc[2], nc[0], N);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  UView uv{u && u->ptr ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  AgsfOut ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs), d_leaf_idx};
  auto kern = agsf_scan_kernel<N, M, NODES, NW>;
  if (lds_bytes > 64 * 1024)
    BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int tpb = NT / MP;
  hipLaunchKernelGGL(kern, dim3((unsigned)((B + tpb - 1) / tpb)), dim3(NT), lds_bytes, stream, arg, yv, uv, cv, ov, B, T, nc[0],
                     nc[1], nc[2], MP, opt[0], opt[1], key[0], key[1], variant, carry_records, d_tvq, d_tvr);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

template <int N, int M, class NODES>
static inline int launch_agsf_nodes(typename NODES::Arg arg, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                                    const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry,
                                    const bf_out_desc* out, int* d_leaf_idx, int variant, const float* d_tvq, const float* d_tvr,
                                    hipStream_t stream) {
  const long long Mleaf = (long long)nc[0] * nc[1] * nc[2];
  if (Mleaf > 1024)
    return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: %lld leaves per trajectory exceed one workgroup (1024)", Mleaf);
  int MP = 1;
  while (MP < Mleaf) MP <<= 1;
  if (out->pred_means.ptr || out->pred_covs.ptr || out->coll_mean.ptr || out->coll_cov.ptr || out->loglik.ptr)
    return set_error(BF_EINVAL, "the augmented filter emits weights, means and covariances only (inference.py:771-775)");
#define BF_GEOM(NW_) return launch_agsf_geom<N, M, NODES, NW_>(arg, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, MP, d_tvq, d_tvr, stream)
  if (MP <= 64) BF_GEOM(1);
  if constexpr (N <= 4) {  // the multi-wave geometries are built for the small state dimensions only (build time, LDS)
    if (MP <= 128) {  // e.g. the [5, 5, 5] tree of the reference's own test (docs/tests/test_inference.py): 125 leaves on 2 waves
      MP = 128;
      BF_GEOM(2);
    }
    if (MP <= 256) {
      MP = 256;
      BF_GEOM(4);
    }
    if (MP <= 512) {  // e.g. the [100, 2, 2] tree of BOT_Experiment_script.py:118: 400 leaves on 8 waves, two trajectories per CU
      MP = 512;
      BF_GEOM(8);
    }
    MP = 1024;
    BF_GEOM(16);
  } else {
    return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: more than 64 leaves per trajectory need state_dim <= 4");
  }
#undef BF_GEOM
}

template <int N, int M>
static inline int launch_agsf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                              const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry,
                              const bf_out_desc* out, int* d_leaf_idx, int variant, hipStream_t stream) {
  EkfModel<N, M> e;
  std::vector<float> tvq, tvr;
  int rc = fill_model<N, M>(p, e, &tvq, &tvr);
  if (rc != BF_OK) return rc;
  if (p->flags != 0) return set_error(BF_EUNSUPPORTED, "legacy-class flags do not apply to the augmented filter");
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  const float *d_tvq = nullptr, *d_tvr = nullptr;
  if ((rc = upload_table(tvq, stream, &d_tvq)) != BF_OK || (rc = upload_table(tvr, stream, &d_tvr)) != BF_OK) return rc;
  return launch_agsf_nodes<N, M, EkfNodes<N, M>>(e, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, d_tvq, d_tvr, stream);
}

// unscented nodes (speedy_unscented_agsf / unscented_agsf, inference.py:966-1156 / 813-965)
template <int N, int DQ, int M, int DR>
static inline int launch_uagsf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B,
                               long long T, const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry,
                               const bf_out_desc* out, int* d_leaf_idx, int variant, hipStream_t stream) {
  UkfModel<N, DQ, M, DR> h;
  std::memset(&h, 0, sizeof(h));  // the constant cache compares contents
  std::vector<float> tvsq, tvsr;
  int rc = fill_ukf_model<N, DQ, M, DR>(p, up, h, &tvsq, &tvsr);
  if (rc != BF_OK) return rc;
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  const void* dv = nullptr;
  rc = device_constants(&h, sizeof(h), stream, &dv);
  if (rc != BF_OK) return rc;
  const UkfModel<N, DQ, M, DR>* d_mdl = static_cast<const UkfModel<N, DQ, M, DR>*>(dv);
  const float *d_tvq = nullptr, *d_tvr = nullptr;
  if ((rc = upload_table(tvsq, stream, &d_tvq)) != BF_OK || (rc = upload_table(tvsr, stream, &d_tvr)) != BF_OK) return rc;
  return launch_agsf_nodes<N, M, UkfNodes<N, DQ, M, DR>>(d_mdl, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, d_tvq, d_tvr,
                                                         stream);
}

#endif  // BF_JIT

}  // namespace bf
