// Instantiations of the bootstrap particle filter kernel (bpf_scan.hpp) for a slice of the
// (n, dq, m) table; split over several translation units to build in parallel.
#include "bpf_wide.hpp"

namespace bf {

int launch_bpf_group_c(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP,
        float ess, int resampler, const uint32_t key[2], const BpfCarry& cr, const BpfOut& out, hipStream_t stream, bool* matched) {
  const bf_model* p = &bp->ssm;
#define BF_CASE(N_, DQ_, M_)                                                                        \
  if (p->n == N_ && p->dq == DQ_ && p->m == M_) {                                                     \
    *matched = true;                                                                                  \
    return launch_bpf_dims<N_, DQ_, M_>(bp, y, u, B, T, NP, ess, resampler, key, cr, out, stream); \
  }
  BF_CASE(2, 2, 1);
  BF_CASE(4, 2, 2);
  BF_CASE(4, 4, 1);
  BF_CASE(6, 6, 3);
#undef BF_CASE
  *matched = false;
  return BF_OK;
}

}  // namespace bf
