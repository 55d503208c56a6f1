// Instantiations of the lane-group Kalman kernel (kf_scan_group.hpp) for a slice of the
// (n, m, lanes-per-trajectory) table; split over several translation units to build in parallel.
#include "kf_scan_group.hpp"

namespace bf {

int launch_kf_group_c(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
        const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes, bool* matched) {
#define BF_CASE(N_, M_, NL_)                                                     \
  if (p->n == N_ && p->m == M_ && (lanes == 0 || lanes == NL_)) {                \
    *matched = true;                                                             \
    return launch_nml<N_, M_, NL_>(p, y, B, T, carry, out, stream, force_mode); \
  }
  BF_CASE(3, 1, 2);
  BF_CASE(3, 3, 2);
  BF_CASE(4, 2, 1);
  BF_CASE(5, 3, 4);
  BF_CASE(6, 3, 4);
  BF_CASE(7, 3, 4);
  BF_CASE(8, 3, 4);
#undef BF_CASE
  *matched = false;
  return BF_OK;
}

}  // namespace bf
