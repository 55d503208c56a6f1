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
#include "sample_ssm.hpp"

namespace bf {

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

int launch_sample_user_impl(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                            float* d_states, float* d_emis, hipStream_t stream);   // user_model.hip

int launch_sample_ssm(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                      float* d_states, float* d_emis, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
  if (p->user)   // functions from the caller's source: the same kernel compiled at run time around them
    return launch_sample_user_impl(bp, d_keys, u, B, T, d_states, d_emis, stream);
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
