// Instances and dispatch of the unscented Gaussian-sum filter kernel (ugsf_scan.hpp) over the compiled
// (n, dq, m, dr) table.
#include "ugsf_scan.hpp"

namespace bf {

int launch_ugsf_user_impl(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                          int K, const bf_carry* carry, const bf_out_desc* out, hipStream_t stream);

const bf_user_model* registry_jit_handle(const bf_model* p, bool hw_arith);   // user_model.hip

int launch_ugsf_ukf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B,
                    long long T, int K, const bf_carry* carry, const bf_out_desc* out, hipStream_t stream) {
  if (p->user)   // functions from the caller's source: the kernel compiled at run time for this model (user_model.hip)
    return launch_ugsf_user_impl(p, up, y, u, B, T, K, carry, out, stream);
  if (p->dyn_id == BF_FN_USER || p->emi_id == BF_FN_USER)
    return set_error(BF_EINVAL, "dyn_id / emi_id = BF_FN_USER needs bf_model.user (bf_user_model_create)");
#define BF_CASE(N_, DQ_, M_, DR_)                                                      \
  if (p->n == N_ && p->dq == DQ_ && p->m == M_ && p->dr == DR_)                        \
    return launch_ugsf<N_, DQ_, M_, DR_>(p, up, y, u, B, T, K, carry, out, stream);
  BF_CASE(1, 1, 1, 1);   // growth / sine + quadratic (Experiment_TSP_2023.ipynb cell 2)
  BF_CASE(2, 2, 1, 1);
  BF_CASE(2, 2, 2, 2);   // stochastic volatility (adaptive_experiment.py:51-54)
  BF_CASE(3, 3, 1, 1);   // Lorenz-63 + quadratic (exp_lorentz63.py)
  BF_CASE(3, 3, 3, 3);
  BF_CASE(4, 2, 1, 1);   // manoeuvring target + bearing only (docs/tests/test_inference.py)
  BF_CASE(4, 2, 2, 2);   // manoeuvring target + bearing / range, constant-velocity models (BOT_Experiment_script.py)
  BF_CASE(4, 4, 2, 2);
  BF_CASE(8, 8, 4, 4);   // Lorenz-96 with the even-state emission (nonlinearities.py:37-50)
#undef BF_CASE
  // no compiled instance for these dimensions: the same kernel, compiled now (needs hiprtc; dimensions up to 8)
  bf_model jit = *p;
  jit.user = registry_jit_handle(p, false);
  if (!jit.user) return set_error(BF_ENOGPU, "no current device");
  return launch_ugsf_user_impl(&jit, up, y, u, B, T, K, carry, out, stream);
}

}  // namespace bf
