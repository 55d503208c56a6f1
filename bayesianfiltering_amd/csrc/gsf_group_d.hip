// Instantiations of the Gaussian-sum / EKF kernel (gsf_scan.hpp) for a slice of the
// (n, m, lanes-per-chain) table; split over several translation units to build in parallel.
#include "gsf_scan.hpp"

namespace bf {

int launch_gsf_group_d(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
        const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes, bool* matched) {
#define BF_CASE(N_, M_, NL_)                                                        \
  if (p->n == N_ && p->m == M_ && (lanes == 0 || lanes == NL_)) {                   \
    *matched = true;                                                                \
    return launch_gsf<N_, M_, NL_>(p, y, u, B, T, K, carry, out, stream, force_mode); \
  }
  BF_CASE(4, 4, 2);
  BF_CASE(5, 1, 1);
  BF_CASE(5, 4, 1);
  BF_CASE(6, 2, 2);
  BF_CASE(7, 3, 1);
  BF_CASE(8, 4, 2);
#undef BF_CASE
  *matched = false;
  return BF_OK;
}

}  // namespace bf
