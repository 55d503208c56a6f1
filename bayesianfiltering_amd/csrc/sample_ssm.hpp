// sample_ssm.hpp: the device side of NonlinearSSM.sample (gaussfiltax/models.py:240-289) -- see sample_ssm.hip.  A header so
// that user_model.hip can hand the same code to hiprtc with the caller's f / h compiled in (SpecUser).
#pragma once
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

template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ void
sample_ssm_body(const BpfModel<N, DQ, M>* __restrict__ mdlp, const EmissionNoise<M>* __restrict__ enp,
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
      dyn_value<N, DQ, M, SP>(mdl, x, q, u0, xn);   // (SP::user_dyn: the caller's f(x, q, u) from source)
      BF_UNROLL for (int d = 0; d < N; ++d) x[d] = xn[d];
    }
    float hx[M];
#ifdef BF_USER_EMI
    if constexpr (SP::user_emi) {
      bfu::emission<float>(x, r, u0, mdl.uth_emi, hx);   // the caller's h(x, r, u) from source
    } else
#endif
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
__global__ void __launch_bounds__(64)
sample_ssm_kernel(const BpfModel<N, DQ, M>* __restrict__ mdlp, const EmissionNoise<M>* __restrict__ enp,
                  const uint32_t* __restrict__ keys, const float* __restrict__ uptr, long long u_sB, long long u_sT,
                  float* __restrict__ states, float* __restrict__ emis, long long B, long long T) {
  sample_ssm_body<N, DQ, M, SpecRuntime>(mdlp, enp, keys, uptr, u_sB, u_sT, states, emis, B, T);
}

}  // namespace bf
