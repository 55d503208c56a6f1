// Dispatch of the Gaussian-sum / EKF kernel over the compiled (n, m, lanes) table
// (instantiations live in gsf_group_{a,b,c,d}.hip).
#include "bf_common.hpp"

namespace bf {

#define BF_DECL(F_)                                                                                               \
  int F_(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,            \
         const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes, bool* matched)
BF_DECL(launch_gsf_group_a);
BF_DECL(launch_gsf_group_b);
BF_DECL(launch_gsf_group_c);
BF_DECL(launch_gsf_group_d);
BF_DECL(launch_gsf_group_e);
#undef BF_DECL

// n = 1..8 with m = 1..min(n, 4); `lanes` = 0 picks the default lanes per chain.
Option g_gsf_structured{1, OPT_GSF_STRUCTURED};  // tuning / test hook (bf_set_option "gsf_structured")

int launch_gsf_ekf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                   const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes) {
  if (lanes == 0) {  // default lanes per chain: the largest column block per lane that divides n (fewest redundant VALU ops)
    static const int kDefault[9] = {0, 1, 1, 1, 2, 1, 2, 1, 2};
    if (p->n >= 1 && p->n <= 8) lanes = kDefault[p->n];
  }
  bool matched = false;
  int rc = BF_OK;
  if (g_gsf_structured) {  // structure-aware instances first (Lorenz-96 + pick-even emission)
    rc = launch_gsf_group_e(p, y, u, B, T, K, carry, out, stream, force_mode, lanes, &matched);
    if (matched) return rc;
  }
  rc = launch_gsf_group_a(p, y, u, B, T, K, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_gsf_group_b(p, y, u, B, T, K, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_gsf_group_c(p, y, u, B, T, K, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  rc = launch_gsf_group_d(p, y, u, B, T, K, carry, out, stream, force_mode, lanes, &matched);
  if (matched) return rc;
  return set_error(BF_EUNSUPPORTED, "gaussian-sum filter: (n=%d, m=%d, lanes=%d) is not compiled in (n = 1..8 with m = 1..min(n,4))",
                   p->n, p->m, lanes);
}

}  // namespace bf
