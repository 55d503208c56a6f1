// Dispatch of the bootstrap particle filter over the compiled (n, dq, m) table (instantiations in
// bpf_group_{a,b,c}.hip) and the stand-alone resampler.
#include "bpf_scan.hpp"

namespace bf {

Option g_bpf_variant{0, OPT_BPF_VARIANT};   // tuning hook (bf_set_option "bpf_variant")
Option g_bpf_hbm_mode{0, OPT_BPF_HBM_MODE};  // bf_set_option "bpf_hbm_mode"
Option g_bpf_spec{1, OPT_BPF_SPEC};      // bf_set_option "bpf_spec"
Option g_bpf_arith{0, OPT_BPF_ARITH};    // bf_set_option "bpf_arith": 0 = canonical fp32 arithmetic (bit-exact ancestry), 1 = hardware v_exp_f32 / v_log_f32

// Stand-alone resampler: idx[b][:] = choice(key_b, N, (N,), p = w[b]) (the index draw of utils.py:210)
template <int PPT, int NW>
__global__ void __launch_bounds__(64 * NW)
resample_kernel(const float* __restrict__ w, const uint32_t* __restrict__ keys, int NP, int resampler, int* __restrict__ idx) {
  constexpr int CAP = 64 * NW * PPT;
  __shared__ float cdf[cdf_words(CAP)];
  __shared__ float red[64];
  const int tid = threadIdx.x;
  const long long b = blockIdx.x;
  float wn[PPT];
  bool valid[PPT];
  int anc[PPT];
  BF_UNROLL for (int p = 0; p < PPT; ++p) {
    valid[p] = tid * PPT + p < NP;
    wn[p] = valid[p] ? w[b * NP + tid * PPT + p] : 0.f;
  }
  resample_indices<PPT, NW>(wn, valid, NP, U32x2{keys[b * 2], keys[b * 2 + 1]}, resampler, cdf, red, anc);
  BF_UNROLL for (int p = 0; p < PPT; ++p) if (valid[p]) idx[b * NP + tid * PPT + p] = anc[p];
}

int launch_resample(const float* d_w, const uint32_t* d_keys, long long B, int NP, int resampler, int* d_idx,
                    hipStream_t stream) {
  if (NP <= 64) hipLaunchKernelGGL((resample_kernel<1, 1>), dim3((unsigned)B), dim3(64), 0, stream, d_w, d_keys, NP, resampler, d_idx);
  else if (NP <= 256) hipLaunchKernelGGL((resample_kernel<1, 4>), dim3((unsigned)B), dim3(256), 0, stream, d_w, d_keys, NP, resampler, d_idx);
  else if (NP <= 1024) hipLaunchKernelGGL((resample_kernel<1, 16>), dim3((unsigned)B), dim3(1024), 0, stream, d_w, d_keys, NP, resampler, d_idx);
  else if (NP <= 4096) hipLaunchKernelGGL((resample_kernel<4, 16>), dim3((unsigned)B), dim3(1024), 0, stream, d_w, d_keys, NP, resampler, d_idx);
  else return set_error(BF_EUNSUPPORTED, "resample: %d particles exceed the compiled capacity of 4096", NP);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

#define BF_DECL(F_)                                                                                                   \
  int F_(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess, \
         int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out, hipStream_t stream, bool* matched)
BF_DECL(launch_bpf_group_a);
BF_DECL(launch_bpf_group_b);
BF_DECL(launch_bpf_group_c);
#undef BF_DECL

int launch_bpf_user_impl(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess,
                         int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream);

const bf_user_model* registry_jit_handle(const bf_model* p, bool hw_arith);   // user_model.hip
int launch_bpf_hw_arith_impl(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess,
                             int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream);

int launch_bpf(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP,
               float ess, int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o,
               hipStream_t stream) {
  BpfCarry cr{carry ? carry->x_in : nullptr, carry ? carry->w_in : nullptr, carry ? carry->key_in : nullptr,
              carry ? carry->x_out : nullptr, carry ? carry->w_out : nullptr, carry ? carry->key_out : nullptr};
  BpfOut out{o->weights, o->w_sB, o->w_sN, o->w_sT, o->particles, o->x_sB, o->x_sN, o->x_sT, o->ancestors,
             o->mean, o->ess, o->logz, o->resampled};
  if (bp->ssm.user)   // functions from the caller's source: the kernel compiled at run time for this model (user_model.hip)
    return launch_bpf_user_impl(bp, y, u, B, T, NP, ess, resampler, key, carry, o, stream);
  if (bp->ssm.dyn_id == BF_FN_USER || bp->ssm.emi_id == BF_FN_USER)
    return set_error(BF_EINVAL, "dyn_id / emi_id = BF_FN_USER needs bf_model.user (bf_user_model_create)");
  if (g_bpf_arith == 1)   // the same kernel with the hardware's transcendentals, compiled at run time (user_model.hip)
    return launch_bpf_hw_arith_impl(bp, y, u, B, T, NP, ess, resampler, key, carry, o, stream);
  bool matched = false;
  int rc = launch_bpf_group_a(bp, y, u, B, T, NP, ess, resampler, key, cr, out, stream, &matched);
  if (matched) return rc;
  rc = launch_bpf_group_b(bp, y, u, B, T, NP, ess, resampler, key, cr, out, stream, &matched);
  if (matched) return rc;
  rc = launch_bpf_group_c(bp, y, u, B, T, NP, ess, resampler, key, cr, out, stream, &matched);
  if (matched) return rc;
  // no compiled instance for these dimensions: the same kernel, compiled now (needs hiprtc; in-register particle counts)
  bf_bpf_model jit = *bp;
  jit.ssm.user = registry_jit_handle(&bp->ssm, false);
  if (!jit.ssm.user) return set_error(BF_ENOGPU, "no current device");
  return launch_bpf_user_impl(&jit, y, u, B, T, NP, ess, resampler, key, carry, o, stream);
}

}  // namespace bf
