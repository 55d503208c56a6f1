// bpf_big: bootstrap particle filter for particle counts beyond the in-register kernel (bpf_scan.hpp).
//
// Same recursion as bpf_scan_kernel -- the lax.scan body of bootstrap_particle_filter
// (gaussfiltax/inference.py:1330-1377) with _resample (utils.py:207-214) -- for N up to 2^20 particles
// per trajectory, the counts the reference's own scripts use (5e4 in BOT_Experiment_script.py:138,
// 5e5 in Experiment_TSP_2023.ipynb).  The particles of a trajectory live in HBM (two buffers, the
// resampling gather goes from one to the other); one 1024-thread workgroup per trajectory walks them
// in chunks of 1024, so every reduction and the cumulative sum keep the binary-tree order of the small
// kernel and of the oracle:
//   * max / sum: adjacent-pair tree inside a chunk (xor butterfly, then across waves), then the same tree over
//     the chunk results (N' = next_pow2 chunks, zero / -inf padded);
//   * CDF = lax.associative_scan order (Brent-Kung) over N' elements: per chunk the up-sweep yields the chunk
//     total; the totals are scanned by the same algorithm; then every chunk redoes its up-sweep and runs the
//     down-sweep with the inclusive value at the end of the previous chunk as the element "before" it;
//   * inverse-CDF draw by binary search over the CDF in global memory, gather through global memory.
// The workgroup of a trajectory is alone on its data: no grid-wide synchronisation, any batch size.
#pragma once
#include "bpf_scan.hpp"

namespace bf {

struct BigScratch {
  float* xa;     // [B][NP][n]   particles
  float* xb;     // [B][NP][n]   gather target
  float* w;      // [B][NP]      weights
  float* ll;     // [B][NP]      log-likelihoods, then unnormalised weights, then normalised weights
  float* cdf;    // [B][NP]
  int* anc;      // [B][NP]
};

constexpr int BIG_NT = 1024;       // threads per trajectory
constexpr int BIG_MAXCH = 1024;    // chunks per trajectory (N <= 2^20)

// Brent-Kung inclusive scan of one 1024-element chunk (thread tid holds v): `excl0` is the inclusive value at
// the end of the previous chunk (the element "before" this one, 0 for the first chunk); when `fix_last`, the
// chunk's last element is `incl_last` (the scanned chunk total).  red: 32 floats of LDS scratch.
__device__ __forceinline__ float chunk_scan_bk(float v, float excl0, float incl_last, bool fix_last, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  BF_UNROLL for (int d = 0; d < 6; ++d) {  // wave up-sweep
    const float o = __shfl_up(v, 1 << d, 64);
    if (((lane + 1) & ((2 << d) - 1)) == 0) v += o;
  }
  lds_barrier();
  if (lane == 63) red[wave] = v;
  lds_barrier();
  float r = (lane < 16) ? red[lane] : 0.f;
  BF_UNROLL for (int d = 0; d < 4; ++d) {  // up-sweep over the 16 wave totals
    const float o = __shfl_up(r, 1 << d, 64);
    if (lane < 16 && ((lane + 1) & ((2 << d) - 1)) == 0) r += o;
  }
  const float total = __shfl(r, 15, 64);  // chunk total (before any prefix)
  if (fix_last && lane == 15) r = incl_last;
  BF_UNROLL for (int d = 4; d >= 1; --d) {  // down-sweep over the wave totals; the node before wave 0 is excl0
    const float o = __shfl_up(r, 1 << (d - 1), 64);
    if (lane < 16 && ((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) r += (lane >= (1 << d)) ? o : excl0;
  }
  const float mine = __shfl(r, wave, 64);
  const float prev = __shfl(r, wave > 0 ? wave - 1 : 0, 64);
  const float excl_wave = wave > 0 ? prev : excl0;
  if (lane == 63) v = mine;
  BF_UNROLL for (int d = 6; d >= 1; --d) {  // wave down-sweep (virtual lane -1 = excl_wave)
    const float o = __shfl_up(v, 1 << (d - 1), 64);
    if (((lane + 1) & ((1 << d) - 1)) == (1 << (d - 1))) v += (lane >= (1 << (d - 1))) ? o : excl_wave;
  }
  (void)total;
  return v;
}

// total of one chunk in the same up-sweep order (== the chunk's last scan element without prefix)
__device__ __forceinline__ float chunk_total_bk(float v, float* red) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  BF_UNROLL for (int d = 0; d < 6; ++d) {
    const float o = __shfl_up(v, 1 << d, 64);
    if (((lane + 1) & ((2 << d) - 1)) == 0) v += o;
  }
  lds_barrier();
  if (lane == 63) red[wave] = v;
  lds_barrier();
  float r = (lane < 16) ? red[lane] : 0.f;
  BF_UNROLL for (int d = 0; d < 4; ++d) {
    const float o = __shfl_up(r, 1 << d, 64);
    if (lane < 16 && ((lane + 1) & ((2 << d) - 1)) == 0) r += o;
  }
  return __shfl(r, 15, 64);
}

// one particle through the dynamics with its own noise draw (inference.py:1342-1345: q = q0 + chol(Q) z from the
// particle's key), written back in place, and its emission log-density (:1348-1349)
template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ float propagate_particle(const BpfModel<N, DQ, M>& mdl, U32x2 ki, float* xp, float u0, const float* yv) {
  float x[N], q[DQ], xn[N];
  BF_UNROLL for (int d = 0; d < N; ++d) x[d] = xp[d];
  draw_dynamics_noise<N, DQ, M, SP>(mdl, ki, q);
  dyn_value<N, DQ, M, SP>(mdl, x, q, u0, xn);       // (SP::user_dyn / user_emi / user_lp: the caller's functions from source)
  BF_UNROLL for (int d = 0; d < N; ++d) xp[d] = xn[d];
  return emission_loglik<N, DQ, M, SP>(mdl, xn, u0, yv);
}

template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ void
bpf_big_body(const BpfModel<N, DQ, M>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB, long long u_sT,
             BpfCarry carry, BpfOut out, BigScratch sc, long long B, long long T, int NP, float ess_threshold, int resampler,
             uint32_t key0, uint32_t key1) {
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long b = blockIdx.x;
  const int nch = (NP + BIG_NT - 1) / BIG_NT;
  int nchp = 1;
  while (nchp < nch) nchp <<= 1;

  __shared__ float red[64];
  __shared__ float cpart[BIG_MAXCH];  // per-chunk partial results (max / sums / CDF totals)
  float* xa = sc.xa + b * (long long)NP * N;
  float* xb = sc.xb + b * (long long)NP * N;
  float* gw = sc.w + b * (long long)NP;
  float* gl = sc.ll + b * (long long)NP;
  float* gc = sc.cdf + b * (long long)NP;
  int* ganc = sc.anc + b * (long long)NP;

  auto block_reduce = [&](float v, auto op) {  // adjacent-pair tree over the 1024 values of a chunk
    BF_UNROLL for (int off = 1; off < 64; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    lds_barrier();
    if (lane == 0) red[32 + wave] = v;
    lds_barrier();
    float r = red[32 + (lane < 16 ? lane : 0)];
    BF_UNROLL for (int off = 1; off < 16; off <<= 1) r = op(r, __shfl_xor(r, off, 64));
    return __shfl(r, 0, 64);
  };
  // the same tree continued over the chunk partials cpart[0 .. nchp), nchp a power of two <= 1024: one value per
  // thread; the butterfly of the first nchp threads only pairs them among themselves
  auto chunks_reduce = [&](auto op) {
    lds_barrier();
    float v = cpart[tid < nchp ? tid : 0];
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
    return v;
  };
  auto fadd = [](float a, float c) { return a + c; };

  uint32_t k0, k1;
  if (carry.key_in) {
    k0 = carry.key_in[b * 2];
    k1 = carry.key_in[b * 2 + 1];
  } else {
    k0 = key0;
    k1 = key1;
  }
  // ---- initial particles
  if (carry.x_in) {
    for (int i = tid; i < NP; i += BIG_NT) {
      BF_UNROLL for (int d = 0; d < N; ++d) xa[(long long)i * N + d] = carry.x_in[(b * NP + i) * N + d];
      gw[i] = carry.w_in[b * NP + i];
    }
  } else {
    for (int i = tid; i < NP; i += BIG_NT) {
      const U32x2 ki = threefry_split(k0, k1, (uint32_t)i + 1u, (uint32_t)NP + 1u);
      float z[N];
      BF_UNROLL for (int d = 0; d < N; ++d) z[d] = bits_to_normal(threefry_bits(ki.x, ki.y, (uint32_t)d, (uint32_t)N));
      BF_UNROLL for (int d = 0; d < N; ++d) {
        float s = 0.f;
        BF_UNROLL for (int c = 0; c <= d; ++c) s = fmaf(mdl.L0[d * N + c], z[c], s);
        xa[(long long)i * N + d] = mdl.m0[d] + s;
      }
      gw[i] = 1.0f / (float)NP;
    }
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);
    k0 = nk.x;
    k1 = nk.y;
  }
  __syncthreads();  // global-memory state written by other threads is read below

  for (long long t = 0; t < T; ++t) {
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = y.p[b * y.sB + t * y.sT + a * y.sE];
    const float u0 = uptr ? uptr[b * u_sB + t * u_sT] : 0.f;
    const U32x2 nk = threefry_split(k0, k1, 0u, (uint32_t)NP + 1u);

    // ---- pass 1: propagate, log-weight, chunk maxima
    for (int c = 0; c < nch; ++c) {
      const int i = c * BIG_NT + tid;
      const bool valid = i < NP;
      float ll = -__builtin_inff();
      if (valid) {
        const U32x2 ki = threefry_split(k0, k1, (uint32_t)i + 1u, (uint32_t)NP + 1u);
        ll = propagate_particle<N, DQ, M, SP>(mdl, ki, xa + (long long)i * N, u0, yv);
        gl[i] = ll;
      }
      const float cm = block_reduce(ll, nanmax);
      if (tid == 0) cpart[c] = cm;
    }
    for (int c = nch + tid; c < nchp; c += BIG_NT) cpart[c] = -__builtin_inff();
    const float mx = chunks_reduce(nanmax);

    // ---- pass 2: unnormalised weights and their sum
    for (int c = 0; c < nch; ++c) {
      const int i = c * BIG_NT + tid;
      float e = 0.f;
      if (i < NP) {
#pragma clang fp contract(off)
        e = canon_exp(gl[i] - mx) * gw[i];
        gl[i] = e;
      }
      const float cs = block_reduce(e, fadd);
      lds_barrier();
      if (tid == 0) cpart[c] = cs;
    }
    for (int c = nch + tid; c < nchp; c += BIG_NT) cpart[c] = 0.f;
    const float tot = chunks_reduce(fadd);

    // ---- pass 3: normalise, effective sample size
    for (int c = 0; c < nch; ++c) {
      const int i = c * BIG_NT + tid;
      float w2 = 0.f;
      if (i < NP) {
#pragma clang fp contract(off)  // the product is rounded before it enters the tree
        const float wn = gl[i] / tot;
        gl[i] = wn;
        w2 = wn * wn;
      }
      const float cs = block_reduce(w2, fadd);
      lds_barrier();
      if (tid == 0) cpart[c] = cs;
    }
    for (int c = nch + tid; c < nchp; c += BIG_NT) cpart[c] = 0.f;
    const float ess = 1.0f / chunks_reduce(fadd);
    const bool do_resample = ess < ess_threshold * (float)NP;

    float* xcur = xa;
    if (do_resample) {
      const U32x2 kc = threefry_split(nk.x, nk.y, 0u, 2u);
      const U32x2 kn = threefry_split(nk.x, nk.y, 1u, 2u);
      // ---- pass 4a: chunk totals of the normalised weights (up-sweep order)
      for (int c = 0; c < nch; ++c) {
        const int i = c * BIG_NT + tid;
        const float ct = chunk_total_bk(i < NP ? gl[i] : 0.f, red);
        lds_barrier();
        if (tid == 0) cpart[c] = ct;
      }
      for (int c = nch + tid; c < nchp; c += BIG_NT) cpart[c] = 0.f;
      lds_barrier();
      // scan of the chunk totals (nchp <= 1024 values, one per thread, same algorithm)
      {
        const float v = cpart[tid < nchp ? tid : 0];
        const float sv = chunk_scan_bk(tid < nchp ? v : 0.f, 0.f, 0.f, false, red);
        lds_barrier();
        if (tid < nchp) cpart[tid] = sv;  // inclusive value at the end of chunk tid
        lds_barrier();
      }
      // ---- pass 4b: the CDF chunk by chunk, prefixed by the end of the previous chunk
      for (int c = 0; c < nch; ++c) {
        const int i = c * BIG_NT + tid;
        const float excl0 = c > 0 ? cpart[c - 1] : 0.f;
        const float cv = chunk_scan_bk(i < NP ? gl[i] : 0.f, excl0, cpart[c], true, red);
        if (i < NP) gc[i] = cv;
      }
      __syncthreads();  // the CDF is read by every thread below
      // ---- pass 5: inverse-CDF draw (searchsorted side='left') and gather
      const float total = gc[NP - 1];
      float u_sys = 0.f;
      if (resampler == 1) u_sys = bits_to_unit(threefry_bits(kc.x, kc.y, 0u, 1u));
      for (int i = tid; i < NP; i += BIG_NT) {
        float r;
        if (resampler == 1) r = (((float)i + u_sys) / (float)NP) * total;
        else r = total * (1.0f - bits_to_unit(threefry_bits(kc.x, kc.y, (uint32_t)i, (uint32_t)NP)));
        int lo = 0, hi = NP;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (gc[mid] < r) lo = mid + 1; else hi = mid;
        }
        const int a = lo < NP - 1 ? lo : NP - 1;
        ganc[i] = a;
        BF_UNROLL for (int d = 0; d < N; ++d) xb[(long long)i * N + d] = xa[(long long)a * N + d];
        gw[i] = 1.0f / (float)NP;
      }
      xcur = xb;
      k0 = kn.x;
      k1 = kn.y;
    } else {
      for (int i = tid; i < NP; i += BIG_NT) {
        gw[i] = gl[i];
        ganc[i] = i;
      }
      k0 = nk.x;
      k1 = nk.y;
    }
    __syncthreads();

    // ---- emit
    if (out.w || out.x || out.anc) {
      for (int i = tid; i < NP; i += BIG_NT) {
        if (out.w) out.w[b * out.w_sB + (long long)i * out.w_sN + t * out.w_sT] = gw[i];
        if (out.anc) out.anc[b * out.w_sB + (long long)i * out.w_sN + t * out.w_sT] = ganc[i];
        if (out.x) BF_UNROLL for (int d = 0; d < N; ++d)
            out.x[b * out.x_sB + (long long)i * out.x_sN + t * out.x_sT + d] = xcur[(long long)i * N + d];
      }
    }
    if (out.mean) {
      float part[N];
      BF_UNROLL for (int d = 0; d < N; ++d) part[d] = 0.f;
      for (int i = tid; i < NP; i += BIG_NT)
        BF_UNROLL for (int d = 0; d < N; ++d) part[d] = fmaf(gw[i], xcur[(long long)i * N + d], part[d]);
      BF_UNROLL for (int d = 0; d < N; ++d) {
        const float s = block_reduce(part[d], fadd);
        if (tid == 0) out.mean[(b * T + t) * N + d] = s;
      }
    }
    if (tid == 0) {
      if (out.ess) out.ess[b * T + t] = ess;
      if (out.logz) out.logz[b * T + t] = mx + canon_log(tot);
      if (out.resampled) out.resampled[b * T + t] = do_resample ? 1.0f : 0.0f;
    }
    if (do_resample) {  // the gather target becomes the current buffer
      float* tmp = xa;
      xa = xb;
      xb = tmp;
    }
    __syncthreads();
  }

  for (int i = tid; i < NP; i += BIG_NT) {
    if (carry.x_out) BF_UNROLL for (int d = 0; d < N; ++d) carry.x_out[(b * NP + i) * N + d] = xa[(long long)i * N + d];
    if (carry.w_out) carry.w_out[b * NP + i] = gw[i];
  }
  if (tid == 0 && carry.key_out) {
    carry.key_out[b * 2] = k0;
    carry.key_out[b * 2 + 1] = k1;
  }
}

template <int N, int DQ, int M>
__global__ void __launch_bounds__(BIG_NT)
bpf_big_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB, long long u_sT,
               BpfCarry carry, BpfOut out, BigScratch sc, long long B, long long T, int NP, float ess_threshold, int resampler,
               uint32_t key0, uint32_t key1) {
  bpf_big_body<N, DQ, M, SpecRuntime>(mdlp, y, uptr, u_sB, u_sT, carry, out, sc, B, T, NP, ess_threshold, resampler, key0, key1);
}

#ifndef BF_JIT
template <int N, int DQ, int M>
static inline int launch_bpf_big_dims(const BpfModel<N, DQ, M>* d_mdl, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                                 int NP, float ess, int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out,
                                 hipStream_t stream) {
  if (NP > BIG_NT * BIG_MAXCH)
    return set_error(BF_EUNSUPPORTED, "bootstrap particle filter: %d particles exceed the capacity of %d per trajectory", NP,
                     BIG_NT * BIG_MAXCH);
  const size_t per = (size_t)B * NP;
  float* buf = nullptr;
  const size_t floats = per * (2 * N + 3);
  BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&buf), sizeof(float) * floats + sizeof(int) * per, stream));
  BigScratch sc;
  sc.xa = buf;
  sc.xb = sc.xa + per * N;
  sc.w = sc.xb + per * N;
  sc.ll = sc.w + per;
  sc.cdf = sc.ll + per;
  sc.anc = reinterpret_cast<int*>(sc.cdf + per);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  hipLaunchKernelGGL((bpf_big_kernel<N, DQ, M>), dim3((unsigned)B), dim3(BIG_NT), 0, stream, d_mdl, yv, (u && u->ptr) ? u->ptr : nullptr,
                     u ? u->sB : 0, u ? u->sT : 0, cr, out, sc, B, T, NP, ess, resampler, key[0], key[1]);
  const hipError_t le = hipGetLastError();
  const hipError_t fe = hipFreeAsync(buf, stream);
  BF_HIP_CHECK(le);
  BF_HIP_CHECK(fe);
  return BF_OK;
}
#endif  // BF_JIT

}  // namespace bf
