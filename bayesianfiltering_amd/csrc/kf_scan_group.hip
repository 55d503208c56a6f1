// Dispatch of the lane-group Kalman kernel over the compiled (n, m, lanes) table
// (instantiations live in kf_group_{a,b,c,d}.hip).
#include "bf_common.hpp"

namespace bf {

#define BF_DECL(F_)                                                                                              \
  int F_(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,                \
         const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes, bool* matched)
BF_DECL(launch_kf_group_a);
BF_DECL(launch_kf_group_b);
BF_DECL(launch_kf_group_c);
BF_DECL(launch_kf_group_d);
#undef BF_DECL

// n = 1..8 with m = 1..min(n, 4); `lanes` = 0 picks the default lanes per trajectory.
int launch_kf_group(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                    const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes) {
  if (lanes == 0 && p->n >= 1 && p->n <= 8) lanes = p->n <= 2 ? 1 : (p->n <= 4 ? 2 : 4);  // default lanes per trajectory
  bool matched = false;
  int rc = launch_kf_group_a(p, y, B, T, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_kf_group_b(p, y, B, T, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_kf_group_c(p, y, B, T, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_kf_group_d(p, y, B, T, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  return set_error(BF_EUNSUPPORTED,
                   "kalman filter: (n=%d, m=%d, lanes=%d) is not compiled in (n = 1..8 with m = 1..min(n,4); n = 64, m = 32)",
                   p->n, p->m, lanes);
}

}  // namespace bf
