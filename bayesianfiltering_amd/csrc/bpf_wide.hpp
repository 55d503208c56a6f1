// bpf_wide: the particles-in-HBM bootstrap particle filter (bpf_big.hpp) with one workgroup per CHUNK of 1024
// particles instead of one per trajectory -- for few trajectories with many particles (the reference's own use:
// one trajectory, 5e4 ... 5e5 particles), where a workgroup per trajectory leaves the chip idle.
//
// A step is six launches on the caller's stream, the stream order being the grid-wide synchronisation between
// the passes; the per-chunk partial results (maxima, sums, CDF chunk totals) go through small arrays in HBM and
// every workgroup recombines them itself with the same tree as bpf_big_kernel, so the two kernels return the same
// bits.  Nothing is read back by the host: the resampling decision is recomputed by each workgroup from the
// partials, the PRNG key and the current-buffer flag of a trajectory are double-buffered by step parity.
//   K1 propagate   particles through f with their noise draw, log-weights, chunk maxima
//   K2 weights     exp(ll - max) * w, chunk sums
//   K3 normalise   w / sum, chunk sums of w^2, chunk totals in CDF up-sweep order
//   K4 cdf         (if ESS < threshold * N) scan of the chunk totals, CDF of the chunk
//   K5 resample    inverse-CDF draw + gather (or pass the weights on), emit, chunk partials of the mean
//   K6 finish      one workgroup per trajectory: summaries, next key, buffer flag
#pragma once
#include "bpf_big.hpp"

namespace bf {

struct WideScratch {
  BigScratch s;
  float* cmax;     // [B][1024] chunk maxima of the log-weights
  float* csum;     // [B][1024] chunk sums of the unnormalised weights
  float* cw2;      // [B][1024] chunk sums of the squared normalised weights
  float* ctot;     // [B][1024] chunk totals of the normalised weights (up-sweep order)
  float* cmean;    // [B][n][1024] chunk partials of sum_i w_i x_i
  uint32_t* keys;  // [2][B][2]  PRNG key of the trajectory, by step parity
  int* cur;        // [2][B]     which of xa / xb holds the particles, by step parity
};

template <class Op>
__device__ __forceinline__ float wide_block_reduce(float v, Op op, float* red) {  // == block_reduce of bpf_big_kernel
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  BF_UNROLL for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
  lds_barrier();
  if (lane == 0) red[32 + wave] = v;
  lds_barrier();
  float r = red[32 + (lane < 16 ? lane : 0)];
  BF_UNROLL for (int off = 1; off < 16; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
  return __shfl(r, 0, 64);
}

// the tree continued over the nchp (power of two <= 1024) chunk partials src[0 .. nchp) in global memory
// (== chunks_reduce of bpf_big_kernel)
template <class Op>
__device__ __forceinline__ float wide_chunks_reduce(const float* src, int nchp, Op op, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float v = src[tid < nchp ? tid : 0];
  for (int off = 1; off < nchp && off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
  if (nchp > 64) {
    lds_barrier();
    if (lane == 0) red[32 + wave] = v;
    lds_barrier();
    const int nw = nchp / 64;
    float r = red[32 + (lane < nw ? lane : 0)];
    for (int off = 1; off < nw; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
    v = __shfl(r, 0, 64);
  } else {
    lds_barrier();
    if (tid == 0) red[48] = v;
    lds_barrier();
    v = red[48];
  }
  lds_barrier();
  return v;
}

struct WideAdd {
  __device__ __forceinline__ float operator()(float a, float c) const { return a + c; }
};
struct WideMax {
  __device__ __forceinline__ float operator()(float a, float c) const { return nanmax(a, c); }
};

template <int N, int DQ, int M>
__global__ void __launch_bounds__(BIG_NT)
wide_init_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, BpfCarry carry, WideScratch sc, long long B, int NP, int nch, int nchp,
                 uint32_t key0, uint32_t key1) {
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int i = c * BIG_NT + tid;
  uint32_t k0 = key0, k1 = key1;
  if (carry.key_in) {
    k0 = carry.key_in[b * 2];
    k1 = carry.key_in[b * 2 + 1];
  }
  float* xa = sc.s.xa + b * (long long)NP * N;
  float* gw = sc.s.w + b * (long long)NP;
  if (i < NP) {
    if (carry.x_in) {
      BF_UNROLL for (int d = 0; d < N; ++d) xa[(long long)i * N + d] = carry.x_in[(b * NP + i) * N + d];
      gw[i] = carry.w_in[b * NP + i];
    } else {
      const U32x2 ki = threefry_split(k0, k1, (uint32_t)i + 1u, (uint32_t)NP + 1u);
      float z[N];
      BF_UNROLL for (int d = 0; d < N; ++d) z[d] = bits_to_normal(threefry_bits(ki.x, ki.y, (uint32_t)d, (uint32_t)N));
      BF_UNROLL for (int d = 0; d < N; ++d) {
        float s = 0.f;
        BF_UNROLL for (int cc = 0; cc <= d; ++cc) s = fmaf(mdl.L0[d * N + cc], z[cc], s);
        xa[(long long)i * N + d] = mdl.m0[d] + s;
      }
      gw[i] = 1.0f / (float)NP;
    }
  }
  if (c == 0) {
    if (tid == 0) {
      if (!carry.x_in) {
        const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
        k0 = nk.x;
        k1 = nk.y;
      }
      sc.keys[b * 2] = k0;
      sc.keys[b * 2 + 1] = k1;
      sc.cur[b] = 0;
    }
    // padding of the partial arrays up to the power of two: identity elements, never overwritten
    for (int cc = nch + tid; cc < nchp; cc += BIG_NT) {
      sc.cmax[b * BIG_MAXCH + cc] = -__builtin_inff();
      sc.csum[b * BIG_MAXCH + cc] = 0.f;
      sc.cw2[b * BIG_MAXCH + cc] = 0.f;
      sc.ctot[b * BIG_MAXCH + cc] = 0.f;
      BF_UNROLL for (int d = 0; d < N; ++d) sc.cmean[(b * N + d) * BIG_MAXCH + cc] = 0.f;
    }
  }
}

template <int N, int DQ, int M>
__global__ void __launch_bounds__(BIG_NT)
wide_propagate_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB, long long u_sT,
                      WideScratch sc, long long B, long long t, int NP) {
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  __shared__ float red[64];
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int p = (int)(t & 1);
  const int i = c * BIG_NT + tid;
  const uint32_t k0 = sc.keys[(p * B + b) * 2], k1 = sc.keys[(p * B + b) * 2 + 1];
  float* x = (sc.cur[p * B + b] ? sc.s.xb : sc.s.xa) + b * (long long)NP * N;
  float yv[M];
  BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = y.p[b * y.sB + t * y.sT + a * y.sE];
  const float u0 = uptr ? uptr[b * u_sB + t * u_sT] : 0.f;
  float ll = -__builtin_inff();
  if (i < NP) {
    const U32x2 ki = threefry_split(k0, k1, (uint32_t)i + 1u, (uint32_t)NP + 1u);
    ll = propagate_particle<N, DQ, M>(mdl, ki, x + (long long)i * N, u0, yv);
    sc.s.ll[b * NP + i] = ll;
  }
  const float cm = wide_block_reduce(ll, WideMax(), red);
  if (tid == 0) sc.cmax[b * BIG_MAXCH + c] = cm;
}

static __global__ void __launch_bounds__(BIG_NT) wide_weights_kernel(WideScratch sc, int NP, int nchp) {
  __shared__ float red[64];
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int i = c * BIG_NT + tid;
  const float mx = wide_chunks_reduce(sc.cmax + b * BIG_MAXCH, nchp, WideMax(), red);
  float e = 0.f;
  if (i < NP) {
#pragma clang fp contract(off)
    e = canon_exp(sc.s.ll[b * NP + i] - mx) * sc.s.w[b * NP + i];
    sc.s.ll[b * NP + i] = e;
  }
  const float cs = wide_block_reduce(e, WideAdd(), red);
  if (tid == 0) sc.csum[b * BIG_MAXCH + c] = cs;
}

static __global__ void __launch_bounds__(BIG_NT) wide_normalise_kernel(WideScratch sc, int NP, int nchp) {
  __shared__ float red[64];
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int i = c * BIG_NT + tid;
  const float tot = wide_chunks_reduce(sc.csum + b * BIG_MAXCH, nchp, WideAdd(), red);
  float wn = 0.f;
  if (i < NP) {
    wn = sc.s.ll[b * NP + i] / tot;
    sc.s.ll[b * NP + i] = wn;
  }
  float w2;
  {
#pragma clang fp contract(off)  // the product is rounded before it enters the tree (as in bpf_big_kernel and the oracle)
    w2 = wn * wn;
  }
  const float cs = wide_block_reduce(w2, WideAdd(), red);
  lds_barrier();
  const float ct = chunk_total_bk(wn, red);
  if (tid == 0) {
    sc.cw2[b * BIG_MAXCH + c] = cs;
    sc.ctot[b * BIG_MAXCH + c] = ct;
  }
}

static __global__ void __launch_bounds__(BIG_NT) wide_cdf_kernel(WideScratch sc, int NP, int nchp, float ess_threshold) {
  __shared__ float red[64];
  __shared__ float cpart[BIG_MAXCH];
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int i = c * BIG_NT + tid;
  const float ess = 1.0f / wide_chunks_reduce(sc.cw2 + b * BIG_MAXCH, nchp, WideAdd(), red);
  if (!(ess < ess_threshold * (float)NP)) return;  // same value in every workgroup of the trajectory
  const float v = sc.ctot[b * BIG_MAXCH + (tid < nchp ? tid : 0)];
  const float sv = chunk_scan_bk(tid < nchp ? v : 0.f, 0.f, 0.f, false, red);
  lds_barrier();
  if (tid < nchp) cpart[tid] = sv;  // inclusive value at the end of chunk tid
  lds_barrier();
  const float excl0 = c > 0 ? cpart[c - 1] : 0.f;
  const float cv = chunk_scan_bk(i < NP ? sc.s.ll[b * NP + i] : 0.f, excl0, cpart[c], true, red);
  if (i < NP) sc.s.cdf[b * NP + i] = cv;
}

template <int N>
__global__ void __launch_bounds__(BIG_NT)
wide_resample_kernel(WideScratch sc, BpfOut out, long long B, long long t, int NP, int nchp, float ess_threshold, int resampler) {
  __shared__ float red[64];
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int p = (int)(t & 1);
  const int i = c * BIG_NT + tid;
  const float ess = 1.0f / wide_chunks_reduce(sc.cw2 + b * BIG_MAXCH, nchp, WideAdd(), red);
  const bool do_resample = ess < ess_threshold * (float)NP;
  const int cur = sc.cur[p * B + b];
  const float* xcur = (cur ? sc.s.xb : sc.s.xa) + b * (long long)NP * N;
  float* xoth = (cur ? sc.s.xa : sc.s.xb) + b * (long long)NP * N;
  const float* gc = sc.s.cdf + b * (long long)NP;
  float wi = 0.f, xi[N];
  int ai = i;
  BF_UNROLL for (int d = 0; d < N; ++d) xi[d] = 0.f;
  if (i < NP) {
    if (do_resample) {
      const uint32_t k0 = sc.keys[(p * B + b) * 2], k1 = sc.keys[(p * B + b) * 2 + 1];
      const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
      const U32x2 kc = threefry_split(nk.x, nk.y, 0u, 2u);
      const float total = gc[NP - 1];
      float r;
      if (resampler == 1) r = (((float)i + bits_to_unit(threefry_bits(kc.x, kc.y, 0u, 1u))) / (float)NP) * total;
      else r = total * (1.0f - bits_to_unit(threefry_bits(kc.x, kc.y, (uint32_t)i, (uint32_t)NP)));
      int lo = 0, hi = NP;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (gc[mid] < r) lo = mid + 1; else hi = mid;
      }
      ai = lo < NP - 1 ? lo : NP - 1;
      BF_UNROLL for (int d = 0; d < N; ++d) {
        xi[d] = xcur[(long long)ai * N + d];
        xoth[(long long)i * N + d] = xi[d];
      }
      wi = 1.0f / (float)NP;
    } else {
      BF_UNROLL for (int d = 0; d < N; ++d) xi[d] = xcur[(long long)i * N + d];
      wi = sc.s.ll[b * NP + i];
    }
    sc.s.w[b * NP + i] = wi;
    if (out.w) out.w[b * out.w_sB + (long long)i * out.w_sN + t * out.w_sT] = wi;
    if (out.anc) out.anc[b * out.w_sB + (long long)i * out.w_sN + t * out.w_sT] = ai;
    if (out.x) BF_UNROLL for (int d = 0; d < N; ++d) out.x[b * out.x_sB + (long long)i * out.x_sN + t * out.x_sT + d] = xi[d];
  }
  if (out.mean) {
    BF_UNROLL for (int d = 0; d < N; ++d) {
      const float s = wide_block_reduce(wi * xi[d], WideAdd(), red);
      if (tid == 0) sc.cmean[(b * N + d) * BIG_MAXCH + c] = s;
      lds_barrier();
    }
  }
}

template <int N>
__global__ void __launch_bounds__(BIG_NT)
wide_finish_kernel(WideScratch sc, BpfOut out, long long B, long long T, long long t, int NP, int nchp, float ess_threshold) {
  __shared__ float red[64];
  const int tid = threadIdx.x;
  const long long b = blockIdx.x;
  const int p = (int)(t & 1);
  const float mx = wide_chunks_reduce(sc.cmax + b * BIG_MAXCH, nchp, WideMax(), red);
  const float tot = wide_chunks_reduce(sc.csum + b * BIG_MAXCH, nchp, WideAdd(), red);
  const float ess = 1.0f / wide_chunks_reduce(sc.cw2 + b * BIG_MAXCH, nchp, WideAdd(), red);
  const bool do_resample = ess < ess_threshold * (float)NP;
  if (out.mean) {
    BF_UNROLL for (int d = 0; d < N; ++d) {
      const float s = wide_chunks_reduce(sc.cmean + (b * N + d) * BIG_MAXCH, nchp, WideAdd(), red);
      if (tid == 0) out.mean[(b * T + t) * N + d] = s;
    }
  }
  if (tid == 0) {
    if (out.ess) out.ess[b * T + t] = ess;
    if (out.logz) out.logz[b * T + t] = mx + canon_log(tot);
    if (out.resampled) out.resampled[b * T + t] = do_resample ? 1.0f : 0.0f;
    const uint32_t k0 = sc.keys[(p * B + b) * 2], k1 = sc.keys[(p * B + b) * 2 + 1];
    U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
    if (do_resample) nk = threefry_split(nk.x, nk.y, 1u, 2u);
    const int q = p ^ 1;
    sc.keys[(q * B + b) * 2] = nk.x;
    sc.keys[(q * B + b) * 2 + 1] = nk.y;
    const int cur = sc.cur[p * B + b];
    sc.cur[q * B + b] = do_resample ? (cur ^ 1) : cur;
  }
}

template <int N>
__global__ void __launch_bounds__(BIG_NT) wide_carry_kernel(WideScratch sc, BpfCarry carry, long long B, long long T, int NP) {
  const int tid = threadIdx.x, c = blockIdx.x;
  const long long b = blockIdx.y;
  const int p = (int)(T & 1);
  const int i = c * BIG_NT + tid;
  const float* x = (sc.cur[p * B + b] ? sc.s.xb : sc.s.xa) + b * (long long)NP * N;
  if (i < NP) {
    if (carry.x_out) BF_UNROLL for (int d = 0; d < N; ++d) carry.x_out[(b * NP + i) * N + d] = x[(long long)i * N + d];
    if (carry.w_out) carry.w_out[b * NP + i] = sc.s.w[b * NP + i];
  }
  if (c == 0 && tid == 0 && carry.key_out) {
    carry.key_out[b * 2] = sc.keys[(p * B + b) * 2];
    carry.key_out[b * 2 + 1] = sc.keys[(p * B + b) * 2 + 1];
  }
}

template <int N, int DQ, int M>
static inline int launch_bpf_wide_dims(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                                       int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out,
                                       hipStream_t stream) {
  const int nch = (NP + BIG_NT - 1) / BIG_NT;
  int nchp = 1;
  while (nchp < nch) nchp <<= 1;
  const size_t per = (size_t)B * NP;
  const size_t floats = per * (2 * N + 3) + (size_t)B * BIG_MAXCH * (4 + N);
  const size_t bytes = sizeof(float) * floats + sizeof(int) * per + sizeof(uint32_t) * 4 * (size_t)B + sizeof(int) * 2 * (size_t)B;
  float* buf = nullptr;
  BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&buf), bytes, stream));
  WideScratch sc;
  sc.s.xa = buf;
  sc.s.xb = sc.s.xa + per * N;
  sc.s.w = sc.s.xb + per * N;
  sc.s.ll = sc.s.w + per;
  sc.s.cdf = sc.s.ll + per;
  sc.cmax = sc.s.cdf + per;
  sc.csum = sc.cmax + (size_t)B * BIG_MAXCH;
  sc.cw2 = sc.csum + (size_t)B * BIG_MAXCH;
  sc.ctot = sc.cw2 + (size_t)B * BIG_MAXCH;
  sc.cmean = sc.ctot + (size_t)B * BIG_MAXCH;
  sc.s.anc = reinterpret_cast<int*>(sc.cmean + (size_t)B * BIG_MAXCH * N);
  sc.keys = reinterpret_cast<uint32_t*>(sc.s.anc + per);
  sc.cur = reinterpret_cast<int*>(sc.keys + 4 * (size_t)B);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  const float* up = (u && u->ptr) ? u->ptr : nullptr;
  const dim3 grid((unsigned)nch, (unsigned)B), blk(BIG_NT);
  hipLaunchKernelGGL((wide_init_kernel<N, DQ, M>), grid, blk, 0, stream, d_mdl, cr, sc, B, NP, nch, nchp, key[0], key[1]);
  for (long long t = 0; t < T; ++t) {
    hipLaunchKernelGGL((wide_propagate_kernel<N, DQ, M>), grid, blk, 0, stream, d_mdl, yv, up, u ? u->sB : 0, u ? u->sT : 0, sc, B, t, NP);
    hipLaunchKernelGGL(wide_weights_kernel, grid, blk, 0, stream, sc, NP, nchp);
    hipLaunchKernelGGL(wide_normalise_kernel, grid, blk, 0, stream, sc, NP, nchp);
    hipLaunchKernelGGL(wide_cdf_kernel, grid, blk, 0, stream, sc, NP, nchp, ess);
    hipLaunchKernelGGL((wide_resample_kernel<N>), grid, blk, 0, stream, sc, out, B, t, NP, nchp, ess, resampler);
    hipLaunchKernelGGL((wide_finish_kernel<N>), dim3((unsigned)B), blk, 0, stream, sc, out, B, T, t, NP, nchp, ess);
  }
  if (cr.x_out || cr.w_out || cr.key_out) hipLaunchKernelGGL((wide_carry_kernel<N>), grid, blk, 0, stream, sc, cr, B, T, NP);
  const hipError_t le = hipGetLastError();
  const hipError_t fe = hipFreeAsync(buf, stream);
  BF_HIP_CHECK(le);
  BF_HIP_CHECK(fe);
  return BF_OK;
}

// particle counts beyond the in-register capacities: one workgroup per trajectory (bpf_big_kernel) when the batch
// alone fills the chip, one workgroup per chunk (this file) otherwise.  bf_set_option("bpf_hbm_mode"): 0 = choose,
// 1 = workgroup per trajectory, 2 = workgroup per chunk.
template <int N, int DQ, int M>
static inline int launch_bpf_hbm_dims(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                                      int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out,
                                      hipStream_t stream) {
  if (NP > BIG_NT * BIG_MAXCH)
    return set_error(BF_EUNSUPPORTED, "bootstrap particle filter: %d particles exceed the capacity of %d per trajectory", NP,
                     BIG_NT * BIG_MAXCH);
  const int hbm_mode = g_bpf_hbm_mode.load();
  const bool wide = B <= 65535 && (hbm_mode == 2 || (hbm_mode == 0 && B < 128));
  if (wide) return launch_bpf_wide_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
  return launch_bpf_big_dims<N, DQ, M>(d_mdl, y, u, B, T, NP, ess, resampler, key, cr, out, stream);
}

}  // namespace bf
