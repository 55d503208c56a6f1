// kf_scan_small: batched Kalman filter, one trajectory per lane, whole time loop in-kernel.
//
// Replaces, for linear f/h and one component, the lax.scan body of gaussian_sum_filter
// (gaussfiltax/inference.py:333-371): per step  _condition_on (:72-105)  ->  reweight
// (:347-350)  ->  _predict (:51-70), emitting the five posterior streams of :357-363.
//
// Mapping (gfx950): state (m, P) of one chain lives in the VGPRs of one lane; the model
// matrices are kernel arguments (SGPR / scalar-cache operands, wave-uniform); a wave
// therefore advances 64 chains per step with no cross-lane traffic.  The kernel is bound by
// the posterior streams it writes (172 B per chain-step at n=4, m=2 against ~600 flop), so
// everything here is about how those bytes reach HBM:
//   EMIT_SCALAR  any strides; one dword store per element.  With the batch-inner layout
//                (sB = 1) each store instruction writes 256 contiguous bytes.
//   EMIT_ROWVEC  reference layout, per-lane 16-byte row stores (each instruction touches 64
//                different cache lines -- correct but TA-bound; kept as the simple baseline).
//   EMIT_STAGED  reference layout through an LDS time-transpose: every lane appends its
//                rows for TS consecutive steps to a per-wave LDS tile, then the wave writes
//                each chain's TS*E floats as whole 128-byte lines with dwordx4 stores.
#include "bf_common.hpp"
#include "kf_math.hpp"

namespace bf {

template <int N, int M>
struct KFConst {
  float A[N * N];    // F_x
  float H[M * N];    // H_x
  float GQG[N * N];  // F_q Q F_q^T
  float DRD[M * M];  // H_r R H_r^T
  float Gq0[N];      // F_q q0
  float Dr0[M];      // H_r r0
};

enum { EMIT_SCALAR = 0, EMIT_ROWVEC = 1, EMIT_STAGED = 2 };

template <int E>
__device__ __forceinline__ void store_scalar(const SView& s, long long b, long long t, const float* v) {
  if (s.p == nullptr) return;
  float* q = s.p + b * s.sB + t * s.sT;
  BF_UNROLL for (int e = 0; e < E; ++e) q[e * s.sE] = v[e];
}

template <int E>
__device__ __forceinline__ void store_rowvec(const SView& s, long long b, long long t, const float* v) {
  if (s.p == nullptr) return;
  float* q = s.p + b * s.sB + t * s.sT;
  if constexpr (E % 4 == 0) {
    BF_UNROLL for (int e = 0; e < E; e += 4)
        *reinterpret_cast<float4*>(q + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
  } else {
    BF_UNROLL for (int e = 0; e < E; ++e) q[e] = v[e];
  }
}

// ---------------------------------------------------------------------------------------
// LDS time-transpose for the reference layout.
//
// One tile per (wave, stream): 64 rows (one per lane/chain) x W floats, W = TS*E the
// number of floats a chain produces for this stream in TS steps (a multiple of 32 = one
// 128-byte line).  The producer lane writes float4 chunk c of its row at chunk column
// (c ^ (lane & (CH-1))) -- an XOR swizzle over the CH = W/4 chunks of the row, so the 8-lane
// groups of ds_write_b128 fall on distinct banks although the row pitch is a power of two.
// The flush reads chunk (lane % CH) of row (lane / CH + 64/CH * i) with the same swizzle and
// stores 16 B per lane: 64/CH complete rows (each W*4 contiguous bytes) per instruction.
template <int E, int TS>
struct Stage {
  static constexpr int W = E * TS;       // floats per row
  static constexpr int CH = W / 4;       // float4 chunks per row
  static constexpr int TILE = 64 * W;    // floats per wave tile
  // a row must be a whole number of float4 chunks, and the chunk count a power of two <= 64
  static constexpr bool OK = (W % 4 == 0) && (CH >= 1) && ((CH & (CH - 1)) == 0) && (CH <= 64);

  // append E floats of step-slot ts (0..TS-1) for this lane
  static __device__ __forceinline__ void put(float* tile, int lane, int ts, const float* v) {
    if constexpr (E % 4 == 0) {
      BF_UNROLL for (int e = 0; e < E; e += 4) {
        int c = (ts * E + e) >> 2;
        int cs = c ^ (lane & (CH - 1));
        *reinterpret_cast<float4*>(tile + lane * W + cs * 4) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
      }
    } else {
      BF_UNROLL for (int e = 0; e < E; ++e) {
        int f = ts * E + e;
        int cs = (f >> 2) ^ (lane & (CH - 1));
        tile[lane * W + cs * 4 + (f & 3)] = v[e];
      }
    }
  }

  // write rows [0,64) of the tile: row r goes to dst + (b0 + r)*sB + t0*E, W floats
  static __device__ __forceinline__ void flush(const float* tile, int lane, float* dst, long long b0,
                                               long long sB, long long t0, long long B) {
    constexpr int RPI = 64 / CH;  // rows per store instruction
    const int c = lane & (CH - 1);
    const int r0 = lane / CH;
    BF_UNROLL for (int i = 0; i < CH; ++i) {
      int r = r0 + i * RPI;
      int cs = c ^ (r & (CH - 1));
      float4 v = *reinterpret_cast<const float4*>(tile + r * W + cs * 4);
      if (b0 + r < B) *reinterpret_cast<float4*>(dst + (b0 + r) * sB + t0 * E + c * 4) = v;
    }
  }
};

// Order LDS traffic between the lanes of ONE wave: the hardware executes a wave's DS
// instructions in order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int N, int M>
struct StagedCfg {
  // steps staged per flush, per stream: rows of >= 128 B for the matrix streams
  static constexpr int TS_P = (N * N >= 32) ? 1 : 32 / (N * N);
  static constexpr int TS_M = (N >= 32) ? 1 : 32 / N;  // means: 128-byte rows
  static constexpr int TS_W = 16;                     // weights / loglik: 64-byte rows (LDS budget)
  static constexpr int TS_Y = (M >= 32) ? 1 : 32 / M;  // observations prefetched per 128-byte row
  static constexpr bool OK = Stage<N * N, TS_P>::OK && Stage<N, TS_M>::OK && Stage<1, TS_W>::OK;
};

template <int N, int M, int MODE, bool TV, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
kf_scan_small_kernel(KFConst<N, M> c, const float* __restrict__ gqg_t, const float* __restrict__ drd_t,
                     CView y, CarryView carry, OutViews out, long long B, long long T, int lds_per_wave) {
  const long long b_raw = (long long)blockIdx.x * (64 * WAVES) + threadIdx.x;
  const bool active = b_raw < B;
  const long long b = active ? b_raw : (B - 1);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long b0 = b_raw - lane;  // first chain of this wave

  float m[N], P[N * N], w;
  BF_UNROLL for (int i = 0; i < N; ++i) m[i] = carry.m_in[b * N + i];
  BF_UNROLL for (int i = 0; i < N * N; ++i) P[i] = carry.P_in[b * N * N + i];
  w = carry.w_in ? carry.w_in[b] : 1.0f;

  // LDS tiles for the staged emitter
  using SC = StagedCfg<N, M>;
  using StP = Stage<N * N, SC::TS_P>;
  using StM = Stage<N, SC::TS_M>;
  using StW = Stage<1, SC::TS_W>;
  // per-wave tiles carved from dynamic LDS; only enabled streams take space (host sizes it
  // with staged_lds_floats_per_wave)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tP = nullptr, *tpP = nullptr, *tM = nullptr, *tpM = nullptr, *tW = nullptr, *tL = nullptr;
  if constexpr (MODE == EMIT_STAGED) {
    float* q = lds + wave * lds_per_wave;
    tP = q;  q += out.P.p ? StP::TILE : 0;
    tpP = q; q += out.pP.p ? StP::TILE : 0;
    tM = q;  q += out.m.p ? StM::TILE : 0;
    tpM = q; q += out.pm.p ? StM::TILE : 0;
    tW = q;  q += out.w.p ? StW::TILE : 0;
    tL = q;
  }

  float yv[M];
  BF_UNROLL for (int e = 0; e < M; ++e) yv[e] = y.p[b * y.sB + e * y.sE];

  for (long long t = 0; t < T; ++t) {
    // prefetch next observation (latency hidden behind this step's arithmetic)
    float yn[M];
    const long long tn = (t + 1 < T) ? t + 1 : t;
    BF_UNROLL for (int e = 0; e < M; ++e) yn[e] = y.p[b * y.sB + tn * y.sT + e * y.sE];

    const float* GQG = c.GQG;
    const float* DRD = c.DRD;
    float gq[N * N], dr[M * M];
    if constexpr (TV) {
      if (gqg_t) {
        BF_UNROLL for (int i = 0; i < N * N; ++i) gq[i] = gqg_t[t * N * N + i];
        GQG = gq;
      }
      if (drd_t) {
        BF_UNROLL for (int i = 0; i < M * M; ++i) dr[i] = drd_t[t * M * M + i];
        DRD = dr;
      }
    }

    // ---- _condition_on: innovation v = y - (H m + H_r r0)
    float v[M];
    mv<M, N>(c.H, m, v);
    BF_UNROLL for (int a = 0; a < M; ++a) v[a] = yv[a] - (v[a] + c.Dr0[a]);
    float ll = condition_on<N, M>(c.H, DRD, v, m, P);

    // ---- reweight (K = 1): lls -= max; w = exp(lls) * w; w /= sum(w)
    w = reweight_single(ll, w);

    if constexpr (MODE == EMIT_SCALAR) {
      if (active) {
        store_scalar<N>(out.m, b, t, m);
        store_scalar<N * N>(out.P, b, t, P);
        store_scalar<1>(out.w, b, t, &w);
        store_scalar<1>(out.ll, b, t, &ll);
      }
    } else if constexpr (MODE == EMIT_ROWVEC) {
      if (active) {
        store_rowvec<N>(out.m, b, t, m);
        store_rowvec<N * N>(out.P, b, t, P);
        store_rowvec<1>(out.w, b, t, &w);
        store_rowvec<1>(out.ll, b, t, &ll);
      }
    } else {
      if (out.m.p) StM::put(tM, lane, int(t % SC::TS_M), m);
      if (out.P.p) StP::put(tP, lane, int(t % SC::TS_P), P);
      if (out.w.p) StW::put(tW, lane, int(t % SC::TS_W), &w);
      if (out.ll.p) StW::put(tL, lane, int(t % SC::TS_W), &ll);
    }

    // ---- _predict: m <- A m + G q0 ; P <- A P A^T + G Q G^T
    float mp[N];
    mv<N, N>(c.A, m, mp);
    BF_UNROLL for (int i = 0; i < N; ++i) m[i] = mp[i] + c.Gq0[i];
    predict_cov<N>(c.A, GQG, P);

    if constexpr (MODE == EMIT_SCALAR) {
      if (active) {
        store_scalar<N>(out.pm, b, t, m);
        store_scalar<N * N>(out.pP, b, t, P);
      }
    } else if constexpr (MODE == EMIT_ROWVEC) {
      if (active) {
        store_rowvec<N>(out.pm, b, t, m);
        store_rowvec<N * N>(out.pP, b, t, P);
      }
    } else {
      if (out.pm.p) StM::put(tpM, lane, int(t % SC::TS_M), m);
      if (out.pP.p) StP::put(tpP, lane, int(t % SC::TS_P), P);
      // flush whichever tiles completed a row at this step (T is a multiple of every TS: host-checked)
      const long long t1 = t + 1;
      if (t1 % SC::TS_P == 0) {
        wave_sync();
        if (out.P.p) StP::flush(tP, lane, out.P.p, b0, out.P.sB, t1 - SC::TS_P, B);
        if (out.pP.p) StP::flush(tpP, lane, out.pP.p, b0, out.pP.sB, t1 - SC::TS_P, B);
      }
      if (t1 % SC::TS_M == 0) {
        wave_sync();
        if (out.m.p) StM::flush(tM, lane, out.m.p, b0, out.m.sB, t1 - SC::TS_M, B);
        if (out.pm.p) StM::flush(tpM, lane, out.pm.p, b0, out.pm.sB, t1 - SC::TS_M, B);
      }
      if (t1 % SC::TS_W == 0) {
        wave_sync();
        if (out.w.p) StW::flush(tW, lane, out.w.p, b0, out.w.sB, t1 - SC::TS_W, B);
        if (out.ll.p) StW::flush(tL, lane, out.ll.p, b0, out.ll.sB, t1 - SC::TS_W, B);
      }
      wave_sync();
    }

    BF_UNROLL for (int e = 0; e < M; ++e) yv[e] = yn[e];
  }

  if (active) {
    if (carry.m_out) BF_UNROLL for (int i = 0; i < N; ++i) carry.m_out[b * N + i] = m[i];
    if (carry.P_out) BF_UNROLL for (int i = 0; i < N * N; ++i) carry.P_out[b * N * N + i] = P[i];
    if (carry.w_out) carry.w_out[b] = w;
  }
}

// ---------------------------------------------------------------------------------------
template <int N, int M>
static void fill_const(const bf_lgssm* p, KFConst<N, M>& c) {
  const int dq = p->dq, dr = p->dr;
  auto Gat = [&](int i, int k) { return p->G ? p->G[i * dq + k] : (i == k ? 1.f : 0.f); };
  auto Dat = [&](int i, int k) { return p->D ? p->D[i * dr + k] : (i == k ? 1.f : 0.f); };
  for (int i = 0; i < N * N; ++i) c.A[i] = p->A[i];
  for (int i = 0; i < M * N; ++i) c.H[i] = p->H[i];
  // (G @ Q) @ G^T and (D @ R) @ D^T in fp32, association as written in inference.py:69,:100
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int l = 0; l < dq; ++l) {
        float gq = 0.f;
        for (int k = 0; k < dq; ++k) gq = fmaf(Gat(i, k), p->Q[k * dq + l], gq);
        s = fmaf(gq, Gat(j, l), s);
      }
      c.GQG[i * N + j] = s;
    }
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      float s = 0.f;
      for (int l = 0; l < dr; ++l) {
        float dq_ = 0.f;
        for (int k = 0; k < dr; ++k) dq_ = fmaf(Dat(i, k), p->R[k * dr + l], dq_);
        s = fmaf(dq_, Dat(j, l), s);
      }
      c.DRD[i * M + j] = s;
    }
  for (int i = 0; i < N; ++i) {
    float s = 0.f;
    for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), p->q0 ? p->q0[k] : 0.f, s);
    c.Gq0[i] = s;
  }
  for (int i = 0; i < M; ++i) {
    float s = 0.f;
    for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), p->r0 ? p->r0[k] : 0.f, s);
    c.Dr0[i] = s;
  }
}

static bool stream_is_reference(const bf_stream& s, long long E, long long T) {
  return s.ptr == nullptr || (s.sE == 1 && s.sT == E && s.sB == T * E && (reinterpret_cast<uintptr_t>(s.ptr) % 16 == 0));
}

template <int N, int M>
static int launch_nm(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                     const bf_out_desc* out, hipStream_t stream, int force_mode) {
  KFConst<N, M> c;
  fill_const<N, M>(p, c);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};

  using SC = StagedCfg<N, M>;
  const bool ref_layout = stream_is_reference(out->weights, 1, T) && stream_is_reference(out->loglik, 1, T) &&
                          stream_is_reference(out->means, N, T) && stream_is_reference(out->pred_means, N, T) &&
                          stream_is_reference(out->covs, N * N, T) && stream_is_reference(out->pred_covs, N * N, T);
  const bool staged_ok = SC::OK && ref_layout && (T % SC::TS_W == 0) && (T % SC::TS_M == 0) && (T % SC::TS_P == 0);
  int mode = EMIT_SCALAR;
  if (ref_layout && N % 4 == 0) mode = EMIT_ROWVEC;
  if (staged_ok) mode = EMIT_STAGED;
  if (force_mode >= 0) {
    if (force_mode == EMIT_STAGED && !staged_ok)
      return set_error(BF_EINVAL, "staged emitter needs the reference layout and T %% %d == 0", SC::TS_W);
    if (force_mode == EMIT_ROWVEC && !(ref_layout && N % 4 == 0))
      return set_error(BF_EINVAL, "row-vector emitter needs the reference layout and n %% 4 == 0");
    mode = force_mode;
  }

  // time-varying covariances: per-step G Q_t G^T / D R_t D^T tables on the device
  float* d_gqg = nullptr;
  float* d_drd = nullptr;
  const bool tv = (p->Q_steps > 1) || (p->R_steps > 1);
  if (tv) {
    if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
      return set_error(BF_EINVAL, "time-varying Q/R need exactly T=%lld matrices", T);
    bf_lgssm pt = *p;
    if (p->Q_steps > 1) {
      float* h = new float[T * N * N];
      for (long long t = 0; t < T; ++t) {
        pt.Q = p->Q + t * p->dq * p->dq;
        KFConst<N, M> ct;
        fill_const<N, M>(&pt, ct);
        for (int i = 0; i < N * N; ++i) h[t * N * N + i] = ct.GQG[i];
      }
      pt.Q = p->Q;
      hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&d_gqg), sizeof(float) * T * N * N, stream);
      if (e == hipSuccess) e = hipMemcpyAsync(d_gqg, h, sizeof(float) * T * N * N, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      delete[] h;
      BF_HIP_CHECK(e);
    }
    if (p->R_steps > 1) {
      float* h = new float[T * M * M];
      for (long long t = 0; t < T; ++t) {
        pt.R = p->R + t * p->dr * p->dr;
        KFConst<N, M> ct;
        fill_const<N, M>(&pt, ct);
        for (int i = 0; i < M * M; ++i) h[t * M * M + i] = ct.DRD[i];
      }
      hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&d_drd), sizeof(float) * T * M * M, stream);
      if (e == hipSuccess) e = hipMemcpyAsync(d_drd, h, sizeof(float) * T * M * M, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      delete[] h;
      BF_HIP_CHECK(e);
    }
  }

  constexpr int WAVES = 1;
  int lds_per_wave = 0;
  if constexpr (SC::OK) if (mode == EMIT_STAGED) {
    using StP = Stage<N * N, SC::TS_P>;
    using StM = Stage<N, SC::TS_M>;
    using StW = Stage<1, SC::TS_W>;
    lds_per_wave = (out->covs.ptr ? StP::TILE : 0) + (out->pred_covs.ptr ? StP::TILE : 0) +
                   (out->means.ptr ? StM::TILE : 0) + (out->pred_means.ptr ? StM::TILE : 0) +
                   (out->weights.ptr ? StW::TILE : 0) + (out->loglik.ptr ? StW::TILE : 0);
  }
  const size_t lds_bytes = sizeof(float) * (size_t)lds_per_wave * WAVES;
  dim3 block(64 * WAVES);
  dim3 grid((unsigned)((B + 64 * WAVES - 1) / (64 * WAVES)));
#define BF_LAUNCH(MODE_, TV_)                                                                      \
  hipLaunchKernelGGL((kf_scan_small_kernel<N, M, MODE_, TV_, WAVES>), grid, block, lds_bytes, stream, c, d_gqg, \
                     d_drd, yv, cv, ov, B, T, lds_per_wave)
  if (tv) {
    if (mode == EMIT_SCALAR) BF_LAUNCH(EMIT_SCALAR, true);
    else if (mode == EMIT_ROWVEC) BF_LAUNCH(EMIT_ROWVEC, true);
    else if constexpr (SC::OK) BF_LAUNCH(EMIT_STAGED, true);
  } else {
    if (mode == EMIT_SCALAR) BF_LAUNCH(EMIT_SCALAR, false);
    else if (mode == EMIT_ROWVEC) BF_LAUNCH(EMIT_ROWVEC, false);
    else if constexpr (SC::OK) BF_LAUNCH(EMIT_STAGED, false);
  }
#undef BF_LAUNCH
  BF_HIP_CHECK(hipGetLastError());
  if (d_gqg) BF_HIP_CHECK(hipFreeAsync(d_gqg, stream));
  if (d_drd) BF_HIP_CHECK(hipFreeAsync(d_drd, stream));
  return BF_OK;
}

// (n, m) pairs compiled into the register-resident kernel
int launch_kf_small(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                    const bf_out_desc* out, hipStream_t stream, int force_mode) {
#define BF_CASE(N_, M_) \
  if (p->n == N_ && p->m == M_) return launch_nm<N_, M_>(p, y, B, T, carry, out, stream, force_mode)
  BF_CASE(1, 1);
  BF_CASE(2, 1);
  BF_CASE(2, 2);
  BF_CASE(3, 1);
  BF_CASE(3, 3);
  BF_CASE(4, 1);
  BF_CASE(4, 2);
  BF_CASE(4, 4);
  BF_CASE(8, 4);
#undef BF_CASE
  return set_error(BF_EUNSUPPORTED, "kalman filter: (n=%d, m=%d) is not compiled into the register-resident kernel",
                   p->n, p->m);
}

}  // namespace bf
