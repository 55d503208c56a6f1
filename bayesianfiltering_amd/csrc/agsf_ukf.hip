// Instances and dispatch of the augmented Gaussian-sum filter kernel with unscented nodes (agsf_scan.hpp).
#include "agsf_scan.hpp"

namespace bf {

int launch_agsf_user_impl(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                          const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out,
                          int* d_leaf_idx, int variant, hipStream_t stream);  // user_model.hip

const bf_user_model* registry_jit_handle(const bf_model* p, bool hw_arith);   // user_model.hip

int launch_agsf_ukf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                    const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out,
                    int* d_leaf_idx, int variant, hipStream_t stream) {
  if (p->user) return launch_agsf_user_impl(p, up, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
#define BF_CASE(N_, DQ_, M_, DR_)                                     \
  if (p->n == N_ && p->dq == DQ_ && p->m == M_ && p->dr == DR_)       \
    return launch_uagsf<N_, DQ_, M_, DR_>(p, up, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
  BF_CASE(1, 1, 1, 1);
  BF_CASE(3, 3, 1, 1);
  BF_CASE(4, 2, 1, 1);
  BF_CASE(4, 2, 2, 2);
  BF_CASE(4, 4, 2, 2);
#undef BF_CASE
  // no compiled instance for these dimensions: the same kernel, compiled now (needs hiprtc; dimensions up to 8)
  bf_model jit = *p;
  jit.user = registry_jit_handle(p, false);
  if (!jit.user) return set_error(BF_ENOGPU, "no current device");
  return launch_agsf_user_impl(&jit, up, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
}

}  // namespace bf
