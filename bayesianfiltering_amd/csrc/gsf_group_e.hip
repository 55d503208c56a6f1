// Structure-aware instances of the Gaussian-sum / EKF kernel: Lorenz-96 dynamics with the
// even-state-picking emission (gsf_scan.hpp, SPEC_L96_PICK).
#include "gsf_scan.hpp"

namespace bf {

int launch_gsf_group_e(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
        const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes, bool* matched) {
  *matched = false;
  if (!gsf_is_l96_pick(p)) return BF_OK;
#define BF_CASE(N_, M_, NL_)                                                                       \
  if (p->n == N_ && p->m == M_ && (lanes == 0 || lanes == NL_)) {                                  \
    *matched = true;                                                                               \
    return launch_gsf<N_, M_, NL_, SPEC_L96_PICK>(p, y, u, B, T, K, carry, out, stream, force_mode); \
  }
  BF_CASE(8, 4, 2);
  BF_CASE(8, 4, 1);
  BF_CASE(8, 4, 4);
  BF_CASE(4, 2, 2);
  BF_CASE(4, 2, 1);
  BF_CASE(6, 3, 2);
#undef BF_CASE
  return BF_OK;
}

}  // namespace bf
