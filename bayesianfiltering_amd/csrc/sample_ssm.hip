// sample_ssm: synthetic trajectories of a registry state-space model on the device.
//
// Replaces NonlinearSSM.sample (gaussfiltax/models.py:240-289): split the key three ways, draw
// z_1 ~ N(m0, P0) and r_1, then for t >= 2 draw (q_t, r_t) with the two halves of
// split(next_keys[t-1]) and apply f, h -- the step before the filtering path in every
// experiment script (docs/experiments/BOT_Experiment_script.py:95).  One lane per trajectory
// (the recursion is sequential in t); all randomness is per-lane Threefry with JAX's layout, so
// trajectory b depends only on keys[b].
#include <cstring>
#include "bf_common.hpp"
#include "bf_rng.hpp"
#include "ssm_device.hpp"

namespace bf {

template <int M>
struct EmissionNoise {
  int d_identity, emi_sv, pad0, pad1;
  float Dm[M * M];   // H_r (constant case)
  float LRn[M * M];  // chol(R), lower
  float r0[M];
};

template <int D>
__device__ __forceinline__ void mvn_draw(uint32_t k0, uint32_t k1, const float* loc, const float* L, float* out) {
  float z[D];
  constexpr int h = (D + 1) / 2;
  BF_UNROLL for (int j = 0; j < h; ++j) {
    const U32x2 o = threefry2x32(k0, k1, (uint32_t)j, (h + j < D) ? (uint32_t)(h + j) : 0u);
    z[j] = bits_to_normal(o.x);
    if (h + j < D) z[h + j] = bits_to_normal(o.y);
  }
  BF_UNROLL for (int d = 0; d < D; ++d) {
    float s = 0.f;
    BF_UNROLL for (int c = 0; c <= d; ++c) s = fmaf(L[d * D + c], z[c], s);
    out[d] = loc[d] + s;
  }
}

template <int N, int DQ, int M>
__global__ void __launch_bounds__(64)
sample_ssm_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, const EmissionNoise<M>* __restrict__ enp,
                  const uint32_t* __restrict__ keys, const float* __restrict__ uptr, long long u_sB, long long u_sT,
                  float* __restrict__ states, float* __restrict__ emis, long long B, long long T) {
  const long long b = (long long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const BpfModel<N, DQ, M>& mdl = *mdlp;
  const EmissionNoise<M>& en = *enp;
  const uint32_t k0 = keys[b * 2], k1 = keys[b * 2 + 1];
  const U32x2 key1 = threefry_split(k0, k1, 0u, 3u), key2 = threefry_split(k0, k1, 1u, 3u), key3 = threefry_split(k0, k1, 2u, 3u);
  float x[N], r[M], q[DQ];
  mvn_draw<N>(key1.x, key1.y, mdl.m0, mdl.L0, x);
  mvn_draw<M>(key2.x, key2.y, en.r0, en.LRn, r);
  for (long long t = 0; t < T; ++t) {
    const float u0 = uptr ? uptr[b * u_sB + t * u_sT] : 0.f;
    if (t > 0) {
      const U32x2 kt = threefry_split(key3.x, key3.y, (uint32_t)(t - 1), (uint32_t)(T - 1));
      const U32x2 ka = threefry_split(kt.x, kt.y, 0u, 2u), kb = threefry_split(kt.x, kt.y, 1u, 2u);
      mvn_draw<DQ>(ka.x, ka.y, mdl.q0, mdl.LQ, q);
      mvn_draw<M>(kb.x, kb.y, en.r0, en.LRn, r);
      float xn[N];
      dyn_value<N, DQ, M>(mdl, x, q, u0, xn);
      BF_UNROLL for (int d = 0; d < N; ++d) x[d] = xn[d];
    }
    float hx[M];
    if (en.emi_sv) {
      if constexpr (M == N) {
        const float sigma = mdl.eth[0], beta = mdl.eth[1], c = mdl.eth[2];
        BF_UNROLL for (int i = 0; i < N; ++i) hx[i] = u0 * beta * expf(x[i] / sigma) * r[i] + (1.f - u0) * (c * x[i] + r[i]);
      }
    } else {
      emi_value<N, DQ, M>(mdl, x, u0, hx);  // h(x, 0, u)
      if (en.d_identity) {
        BF_UNROLL for (int a = 0; a < M; ++a) hx[a] += r[a];
      } else {
        BF_UNROLL for (int a = 0; a < M; ++a) {
          float s = 0.f;
          BF_UNROLL for (int c = 0; c < M; ++c) s = fmaf(en.Dm[a * M + c], r[c], s);
          hx[a] += s;
        }
      }
    }
    if (states) BF_UNROLL for (int d = 0; d < N; ++d) states[(b * T + t) * N + d] = x[d];
    if (emis) BF_UNROLL for (int a = 0; a < M; ++a) emis[(b * T + t) * M + a] = hx[a];
  }
}

template <int N, int DQ, int M>
static int launch_sample_dims(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                              float* d_states, float* d_emis, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
  if (p->dr != M) return set_error(BF_EUNSUPPORTED, "sample_ssm: emission noise dimension must equal the emission dimension");
  struct Pack {
    BpfModel<N, DQ, M> mdl;
    EmissionNoise<M> en;
  } h;
  std::memset(&h, 0, sizeof(h));  // the constant cache compares contents: no indeterminate padding
  // reuse the particle-filter model fill with the emission-noise covariance standing in for the
  // log-density covariance and r_eval = 0; the stochastic-volatility emission is evaluated here
  bf_bpf_model tmp = *bp;
  tmp.lp_cov = p->R;
  tmp.r_eval = nullptr;
  const bool sv = p->emi_id == EMI_STOCH_VOL;
  bf_model ssm2 = *p;
  int rc;
  if (sv) {
    // fill the dynamics side through a linear-emission placeholder of the right shape
    float* zeros = new float[(size_t)M * N + (size_t)M * M]();
    for (int i = 0; i < M; ++i) zeros[(size_t)M * N + i * M + i] = 1.f;
    ssm2.emi_id = EMI_LINEAR;
    ssm2.emi_theta = zeros;
    ssm2.n_emi_theta = M * N + M * M;
    tmp.ssm = ssm2;
    rc = fill_bpf_model<N, DQ, M>(&tmp, h.mdl);
    delete[] zeros;
    if (rc == BF_OK) {
      if (p->n_emi_theta != 3 || M != N) return set_error(BF_EINVAL, "stoch_vol: m = n, theta = (sigma, beta, c)");
      h.mdl.emi_id = EMI_STOCH_VOL;
      for (int i = 0; i < 3; ++i) h.mdl.eth[i] = p->emi_theta[i];
    }
  } else {
    rc = fill_bpf_model<N, DQ, M>(&tmp, h.mdl);
  }
  if (rc != BF_OK) return rc;
  h.en.emi_sv = sv ? 1 : 0;
  h.en.d_identity = 1;
  if (p->emi_id == EMI_LINEAR) {
    h.en.d_identity = 0;
    for (int i = 0; i < M * M; ++i) h.en.Dm[i] = p->emi_theta[M * N + i];
  }
  for (int i = 0; i < M * M; ++i) h.en.LRn[i] = h.mdl.LR[i];
  for (int i = 0; i < M; ++i) h.en.r0[i] = p->r0 ? p->r0[i] : 0.f;
  const void* dv = nullptr;
  rc = device_constants(&h, sizeof(h), stream, &dv);
  if (rc != BF_OK) return rc;
  const Pack* d = static_cast<const Pack*>(dv);
  hipLaunchKernelGGL((sample_ssm_kernel<N, DQ, M>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, stream, &d->mdl, &d->en, d_keys,
                     (u && u->ptr) ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0, d_states, d_emis, B, T);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

int launch_sample_generic(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                          float* d_states, float* d_emis, hipStream_t stream);

int launch_sample_ssm(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                      float* d_states, float* d_emis, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
#define BF_CASE(N_, DQ_, M_) \
  if (p->n == N_ && p->dq == DQ_ && p->m == M_ && p->dr == M_) return launch_sample_dims<N_, DQ_, M_>(bp, d_keys, u, B, T, d_states, d_emis, stream)
  BF_CASE(1, 1, 1);
  BF_CASE(2, 2, 1);
  BF_CASE(2, 2, 2);
  BF_CASE(3, 3, 1);
  BF_CASE(3, 3, 3);
  BF_CASE(4, 4, 1);
  BF_CASE(4, 4, 2);
  BF_CASE(4, 2, 2);
  BF_CASE(4, 2, 1);
  BF_CASE(6, 6, 3);
  BF_CASE(8, 8, 4);
  BF_CASE(16, 16, 8);
  BF_CASE(4, 4, 4);
#undef BF_CASE
  // any other shape: the run-time-dimension kernel (generic_scan.hip), one wave per trajectory
  return launch_sample_generic(bp, d_keys, u, B, T, d_states, d_emis, stream);
}

}  // namespace bf
