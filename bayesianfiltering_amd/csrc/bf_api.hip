// C-ABI entry points (include/bayesfilt.h).  Validation and dispatch only; the kernels live in
// the kf_*/gsf_*/bpf_* translation units.
#include <cstring>
#include "bf_common.hpp"
#include "bf_rng.hpp"

namespace bf {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int launch_kf_group(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                    const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes);

int launch_kf_mfma(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                   const bf_out_desc* out, hipStream_t stream, int K, bool multi, int dyn_kind, const float* dth);
int launch_kf_bf32(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry, const bf_out_desc* out,
                   hipStream_t stream, int K, bool multi, int dyn_kind, const float* dth, bool two_per_wave);
int launch_kf_generic(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                      const bf_out_desc* out, hipStream_t stream);
int launch_gsf_generic(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                       const bf_carry* carry, const bf_out_desc* out, hipStream_t stream);
int launch_collapse(const bf_stream* w, const bf_stream* m, const bf_stream* P, long long B, long long T, int K, int n,
                    float* mean_out, float* cov_out, hipStream_t stream);
int launch_gsf_ekf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                   const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode, int lanes);

int launch_ugsf_ukf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B,
                    long long T, int K, const bf_carry* carry, const bf_out_desc* out, hipStream_t stream);
int launch_agsf_ekf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, const int32_t nc[3],
                    const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out, int* d_leaf_idx,
                    int variant, hipStream_t stream);
int launch_agsf_ukf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                    const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out,
                    int* d_leaf_idx, int variant, hipStream_t stream);
int launch_optimal_resample(const float* d_w, const uint32_t key[2], long long B, int M, int N, int* d_idx, float* d_wout,
                            hipStream_t stream);
int launch_bpf(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP,
               float ess, int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o,
               hipStream_t stream);
int launch_sample_ssm(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                      float* d_states, float* d_emis, hipStream_t stream);
int launch_resample(const float* d_w, const uint32_t* d_keys, long long B, int NP, int resampler, int* d_idx,
                    hipStream_t stream);

CallOverrides& call_overrides() {
  static thread_local CallOverrides c = {};
  return c;
}

// tuning options: process-wide defaults (atomics) with per-call overrides (bf_common.hpp: Option)
extern Option g_bpf_variant;
bool gsf_user_regs_eligible(const bf_model* p, int K, const bf_out_desc* out);   // user_model.hip
int launch_gsf_user_regs_impl(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K, const bf_carry* carry,
                              const bf_out_desc* out, hipStream_t stream);
extern Option g_bpf_hbm_mode;
extern Option g_bpf_spec;
extern Option g_bpf_arith;
extern Option g_gsf_structured;
extern Option g_kf_mfma_variant;
static Option g_kf_emit_mode{-1, OPT_KF_EMIT_MODE};  // -1 = choose from the layout
static Option g_kf_lanes{0, OPT_KF_LANES};       // 0 = default lanes per trajectory for the (n, m) pair
// op 0 log, 1 exp, 2 bits -> normal, 3 sin, 4 cos, 5 atan2(in[i], in[n + i])
__host__ __device__ inline float canon_eval_one(int op, const float* in, long long n, long long i) {
  const float x = in[i];
  float s, c;
  switch (op) {
    case 0: return canon_log(x);
    case 1: return canon_exp(x);
    case 2: return bits_to_normal(canon_f_bits(x));
    case 3: canon_sincos(x, &s, &c); return s;
    case 4: canon_sincos(x, &s, &c); return c;
    default: return canon_atan2(x, in[n + i]);
  }
}
__global__ void canon_eval_kernel(int op, const float* __restrict__ in, long long n, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = canon_eval_one(op, in, n, i);
}

static Option g_kf_small_mode{1, OPT_KF_SMALL_MODE};   // bf_set_option "kf_small_mode": 1 = one-wave matrix-core kernel for 9 <= n <= 32 (default), 2 = its two-chains-per-wave variant (same bits, measured slower), 0 = off
static Option g_force_generic{0, OPT_FORCE_GENERIC};  // 1 = run the run-time-dimension kernel even where a compiled instance exists

// A shape / option the compiled instances do not cover falls through to the run-time-dimension kernel
// (generic_scan.hip); if that cannot run it either, both reasons are reported.
template <class F>
static int with_generic_fallback(int rc, F&& generic) {
  if (rc != BF_EUNSUPPORTED) return rc;
  char first[512];
  std::snprintf(first, sizeof(first), "%s", last_error_buf());
  const int rc2 = generic();
  if (rc2 == BF_OK) return BF_OK;
  char second[512];
  std::snprintf(second, sizeof(second), "%s", last_error_buf());
  return set_error(rc2, "%.240s; generic kernel: %.240s", first, second);
}

}  // namespace bf

extern "C" {

int bf_version(void) { return BF_VERSION; }

int bf_abi_check(int32_t header_version, size_t sizeof_out_desc, size_t sizeof_lgssm, size_t sizeof_model,
                 size_t sizeof_bpf_model, size_t sizeof_bpf_out) {
  if (header_version / 100 != BF_VERSION / 100)
    return bf::set_error(BF_EINVAL, "binding was written against header version %d, the library is %d", (int)header_version, BF_VERSION);
#define BF_ABI_SIZE(NAME_, T_)                                                                                  \
  if (NAME_ != 0 && NAME_ != sizeof(T_)) \
    return bf::set_error(BF_EINVAL, "binding's sizeof(" #T_ ") = %zu, the library's is %zu: the struct layouts differ", NAME_, sizeof(T_));
  BF_ABI_SIZE(sizeof_out_desc, bf_out_desc)
  BF_ABI_SIZE(sizeof_lgssm, bf_lgssm)
  BF_ABI_SIZE(sizeof_model, bf_model)
  BF_ABI_SIZE(sizeof_bpf_model, bf_bpf_model)
  BF_ABI_SIZE(sizeof_bpf_out, bf_bpf_out)
#undef BF_ABI_SIZE
  return BF_OK;
}

const char* bf_last_error(void) { return bf::last_error_buf(); }

int bf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

static int set_option_impl(const char* name, int value, bool this_call_only) {
  auto assign = [&](bf::Option& o) {
    if (this_call_only) {
      bf::CallOverrides& c = bf::call_overrides();
      c.value[o.id] = value;
      c.armed[o.id] = true;
    } else {
      o = value;
    }
    return BF_OK;
  };
  if (name && std::strcmp(name, "kf_emit_mode") == 0) {
    if (value < -1 || value > 2) return bf::set_error(BF_EINVAL, "kf_emit_mode must be -1..2");
    return assign(bf::g_kf_emit_mode);
  }
  if (name && std::strcmp(name, "kf_lanes") == 0) {
    if (value < 0 || value > 64 || (value & (value - 1)) != 0) return bf::set_error(BF_EINVAL, "kf_lanes must be 0 or a power of two <= 64");
    return assign(bf::g_kf_lanes);
  }
  if (name && std::strcmp(name, "kf_mfma_variant") == 0) {
    if (value < 1 || value > 5) return bf::set_error(BF_EINVAL, "kf_mfma_variant must be 1 ... 5");
    return assign(bf::g_kf_mfma_variant);
  }
  if (name && std::strcmp(name, "kf_small_mode") == 0) {
    if (value < 0 || value > 2) return bf::set_error(BF_EINVAL, "kf_small_mode must be 0, 1 or 2");
    return assign(bf::g_kf_small_mode);
  }
  if (name && std::strcmp(name, "force_generic") == 0) {
    if (value < 0 || value > 1) return bf::set_error(BF_EINVAL, "force_generic must be 0 or 1");
    return assign(bf::g_force_generic);
  }
  if (name && std::strcmp(name, "gsf_structured") == 0) {
    if (value < 0 || value > 1) return bf::set_error(BF_EINVAL, "gsf_structured must be 0 or 1");
    return assign(bf::g_gsf_structured);
  }
  if (name && std::strcmp(name, "bpf_variant") == 0) {
    if (value < 0 || value > 1) return bf::set_error(BF_EINVAL, "bpf_variant must be 0 or 1");
    return assign(bf::g_bpf_variant);
  }
  if (name && std::strcmp(name, "bpf_hbm_mode") == 0) {
    if (value < 0 || value > 2) return bf::set_error(BF_EINVAL, "bpf_hbm_mode must be 0, 1 or 2");
    return assign(bf::g_bpf_hbm_mode);
  }
  if (name && std::strcmp(name, "bpf_arith") == 0) {
    if (value < 0 || value > 1) return bf::set_error(BF_EINVAL, "bpf_arith must be 0 or 1");
    return assign(bf::g_bpf_arith);
  }
  if (name && std::strcmp(name, "bpf_spec") == 0) {
    if (value < 0 || value > 1) return bf::set_error(BF_EINVAL, "bpf_spec must be 0 or 1");
    return assign(bf::g_bpf_spec);
  }
  return bf::set_error(BF_EINVAL, "unknown option '%s'", name ? name : "(null)");
}

int bf_set_option(const char* name, int value) { return set_option_impl(name, value, false); }

int bf_set_call_option(const char* name, int value) { return set_option_impl(name, value, true); }

int64_t bf_bytes_per_step(int32_t n, int32_t m, int32_t K, const bf_out_desc* out) {
  int64_t per = 0;
  if (!out) {
    per = 1 + 2 * (int64_t)n + 2 * (int64_t)n * n;
  } else {
    if (out->weights.ptr) per += 1;
    if (out->means.ptr) per += n;
    if (out->covs.ptr) per += (int64_t)n * n;
    if (out->pred_means.ptr) per += n;
    if (out->pred_covs.ptr) per += (int64_t)n * n;
    if (out->loglik.ptr) per += 1;
  }
  int64_t once = 0;  // collapsed streams: one Gaussian per step whatever K is
  if (out && out->coll_mean.ptr) once += n;
  if (out && out->coll_cov.ptr) once += (int64_t)n * n;
  return 4 * (int64_t)m + 4 * (int64_t)K * per + 4 * once;
}

int bf_kalman_filter_f32(const bf_lgssm* model, const bf_cstream* y, int64_t B, int64_t T, const bf_carry* carry,
                         const bf_out_desc* out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (out && (out->coll_mean.ptr || out->coll_cov.ptr))
    return bf::set_error(BF_EINVAL, "collapsed streams are produced by bf_gsf_ekf_f32 (with one component they equal means / covs)");
  if (!model || !y || !carry || !out) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0) return bf::set_error(BF_EINVAL, "B and T must be positive (B=%lld, T=%lld)", (long long)B, (long long)T);
  if (model->n <= 0 || model->m <= 0 || model->dq <= 0 || model->dr <= 0)
    return bf::set_error(BF_EINVAL, "non-positive model dimension");
  if (!model->A || !model->H || !model->Q || !model->R) return bf::set_error(BF_EINVAL, "A, H, Q, R are required");
  if (!model->G && model->dq != model->n) return bf::set_error(BF_EINVAL, "G == NULL requires dq == n");
  if (!model->D && model->dr != model->m) return bf::set_error(BF_EINVAL, "D == NULL requires dr == m");
  if (model->Q_steps < 1 || model->R_steps < 1) return bf::set_error(BF_EINVAL, "Q_steps / R_steps must be >= 1");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!carry->m_in || !carry->P_in) return bf::set_error(BF_EINVAL, "carry.m_in and carry.P_in are required");
  hipStream_t hs = static_cast<hipStream_t>(stream);
  auto generic = [&]() { return bf::launch_kf_generic(model, y, B, T, carry, out, hs); };
  if (bf::g_force_generic.load()) return generic();
  // dense products large enough for the matrix cores: (64, 32) itself and, zero-padded into its tiles, every model from
  // n = 24 up (4.9e7 steps/s whatever the size; the run-time-dimension kernel does 3.8e7 at n = 24, 1.6e7 at 32, 1.2e6 at 64)
  // 9 <= n <= 32: one wave per trajectory on single 32 x 32 tiles, 1.8e8 steps/s whatever the size; the run-time-dimension
  // kernel is faster only for the smallest of them (n = 12, m = 4: 1.7e8; n = 16, m = 8: 1.0e8; (32, 32): 3.5e6)
  if (model->n >= 9 && model->n <= 32 && model->m <= 32 && (model->n >= 16 || model->m > 8) && bf::g_kf_small_mode.load() != 0)
    return bf::with_generic_fallback(bf::launch_kf_bf32(model, y, B, T, carry, out, hs, 1, false, 0, nullptr, bf::g_kf_small_mode.load() == 2), generic);
  if (model->n >= 24 && model->n <= 64 && model->m <= 32)
    return bf::with_generic_fallback(bf::launch_kf_mfma(model, y, B, T, carry, out, hs, 1, false, 0, nullptr), generic);
  return bf::with_generic_fallback(
      bf::launch_kf_group(model, y, B, T, carry, out, hs, bf::g_kf_emit_mode.load(), bf::g_kf_lanes.load()), generic);
}

int bf_gsf_ekf_f32(const bf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T, int32_t K,
                   const bf_carry* carry, const bf_out_desc* out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!model || !y || !carry || !out) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0 || K <= 0) return bf::set_error(BF_EINVAL, "B, T and K must be positive");
  if (model->n <= 0 || model->m <= 0 || model->dq <= 0 || model->dr <= 0)
    return bf::set_error(BF_EINVAL, "non-positive model dimension");
  if (!model->Q || !model->R) return bf::set_error(BF_EINVAL, "Q and R are required");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!carry->m_in || !carry->P_in) return bf::set_error(BF_EINVAL, "carry.m_in and carry.P_in are required");
  hipStream_t hs = static_cast<hipStream_t>(stream);
  auto generic = [&]() { return bf::launch_gsf_generic(model, y, u, B, T, K, carry, out, hs); };
  if (model->user && bf::g_force_generic.load() == 0 && bf::gsf_user_regs_eligible(model, K, out))   // from source, n <= 8: state in registers
    return bf::launch_gsf_user_regs_impl(model, y, u, B, T, K, carry, out, hs);
  if (model->user || model->dyn_id == BF_FN_USER || model->emi_id == BF_FN_USER) return generic();  // compiled from source
  if (bf::g_force_generic.load()) return generic();
  // a LINEAR model beyond the register kernels (n >= 9): the Gaussian-sum filter's K components take turns on the matrix-core
  // kernels (bf16 three-term products: one wave per trajectory on single 32 x 32 tiles up to n = 32, four waves on 64 x 64 up
  // to n = 64), per-step Q_t / R_t tables included
  const bool small_tiles = model->n >= 9 && model->n <= 32 && model->m <= 32 && (model->n >= 16 || model->m > 8) && bf::g_kf_small_mode.load() != 0;
  const bool big_tiles = !small_tiles && model->n >= 24 && model->n <= 64 && model->m <= 32;
  const bool lin_emi = model->emi_id == 0 && model->n_emi_theta == model->m * model->n + model->m * model->dr;
  const bool lin_dyn = model->dyn_id == 0 && model->n_dyn_theta == model->n * model->n + model->n * model->dq;
  // registry dynamics with an analytic, sparse Jacobian and identity noise input: extended Kalman chains on the one-wave kernel
  const int dyn_kind = (model->dyn_id == 1 && model->n_dyn_theta == 5 && model->dq == model->n) ? 1
                     : (model->dyn_id == 4 && model->n_dyn_theta == 1 && model->dq == model->n) ? 2 : 0;
  if (lin_emi && model->flags == 0 && K <= 64 && !out->coll_mean.ptr && !out->coll_cov.ptr &&
      (small_tiles || big_tiles) && (lin_dyn || dyn_kind != 0)) {
    bf_lgssm lg;
    std::memset(&lg, 0, sizeof(lg));
    lg.n = model->n; lg.dq = model->dq; lg.m = model->m; lg.dr = model->dr;
    if (lin_dyn) { lg.A = model->dyn_theta; lg.G = model->dyn_theta + model->n * model->n; }
    lg.H = model->emi_theta; lg.D = model->emi_theta + model->m * model->n;
    lg.q0 = model->q0; lg.r0 = model->r0; lg.Q = model->Q; lg.R = model->R;
    lg.Q_steps = model->Q_steps > 0 ? model->Q_steps : 1; lg.R_steps = model->R_steps > 0 ? model->R_steps : 1;
    if (!lin_dyn) {
      float dth[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < model->n_dyn_theta && i < 8; ++i) dth[i] = model->dyn_theta[i];
      if (small_tiles) return bf::with_generic_fallback(bf::launch_kf_bf32(&lg, y, B, T, carry, out, hs, K, true, dyn_kind, dth, bf::g_kf_small_mode.load() == 2), generic);
      return bf::with_generic_fallback(bf::launch_kf_mfma(&lg, y, B, T, carry, out, hs, K, true, dyn_kind, dth), generic);
    }
    if (small_tiles) return bf::with_generic_fallback(bf::launch_kf_bf32(&lg, y, B, T, carry, out, hs, K, K > 1, 0, nullptr, bf::g_kf_small_mode.load() == 2), generic);
    return bf::with_generic_fallback(bf::launch_kf_mfma(&lg, y, B, T, carry, out, hs, K, K > 1, 0, nullptr), generic);   // 33 <= n <= 64: four waves per trajectory
  }
  return bf::with_generic_fallback(
      bf::launch_gsf_ekf(model, y, u, B, T, K, carry, out, hs, bf::g_kf_emit_mode.load(), bf::g_kf_lanes.load()), generic);
}

int bf_ugsf_ukf_f32(const bf_model* model, const bf_ukf_params* uparams, const bf_cstream* y, const bf_cstream* u,
                    int64_t B, int64_t T, int32_t K, const bf_carry* carry, const bf_out_desc* out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!model || !uparams || !y || !carry || !out) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0 || K <= 0) return bf::set_error(BF_EINVAL, "B, T and K must be positive");
  if (model->n <= 0 || model->m <= 0 || model->dq <= 0 || model->dr <= 0)
    return bf::set_error(BF_EINVAL, "non-positive model dimension");
  if (!model->Q || !model->R) return bf::set_error(BF_EINVAL, "Q and R are required");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!carry->m_in || !carry->P_in) return bf::set_error(BF_EINVAL, "carry.m_in and carry.P_in are required");
  if (!(uparams->alpha > 0.f)) return bf::set_error(BF_EINVAL, "ParamsUKF.alpha must be positive");
  return bf::launch_ugsf_ukf(model, uparams, y, u, B, T, K, carry, out, static_cast<hipStream_t>(stream));
}

int bf_agsf_ekf_f32(const bf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T,
                    const int32_t num_components[3], const uint32_t key[2], const float opt_args[2], const bf_carry* carry,
                    const bf_out_desc* out, int32_t* leaf_idx, int32_t variant, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (variant < 0 || variant > 2)
    return bf::set_error(BF_EINVAL, "variant must be 0 (speedy), 1 (container branches) or 2 (container branches + optimal resampling)");
  if (!model || !y || !carry || !out || !num_components || !key || !opt_args) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0) return bf::set_error(BF_EINVAL, "B and T must be positive");
  if (num_components[0] <= 0 || num_components[1] <= 0 || num_components[2] <= 0)
    return bf::set_error(BF_EINVAL, "num_components must be three positive counts");
  if (model->n <= 0 || model->m <= 0 || model->dq <= 0 || model->dr <= 0)
    return bf::set_error(BF_EINVAL, "non-positive model dimension");
  if (!model->Q || !model->R) return bf::set_error(BF_EINVAL, "Q and R are required");
  if (model->Q_steps < 1 || model->R_steps < 1) return bf::set_error(BF_EINVAL, "Q_steps / R_steps must be >= 1");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!carry->m_in || !carry->P_in) return bf::set_error(BF_EINVAL, "carry.m_in and carry.P_in are required");
  return bf::launch_agsf_ekf(model, y, u, B, T, num_components, key, opt_args, carry, out, leaf_idx, variant,
                             static_cast<hipStream_t>(stream));
}

int bf_agsf_ukf_f32(const bf_model* model, const bf_ukf_params* uparams, const bf_cstream* y, const bf_cstream* u, int64_t B,
                    int64_t T, const int32_t num_components[3], const uint32_t key[2], const float opt_args[2],
                    const bf_carry* carry, const bf_out_desc* out, int32_t* leaf_idx, int32_t variant, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (variant != 0 && variant != 1) return bf::set_error(BF_EINVAL, "variant must be 0 (speedy) or 1 (container branches)");
  if (!model || !uparams || !y || !carry || !out || !num_components || !key || !opt_args) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0) return bf::set_error(BF_EINVAL, "B and T must be positive");
  if (num_components[0] <= 0 || num_components[1] <= 0 || num_components[2] <= 0)
    return bf::set_error(BF_EINVAL, "num_components must be three positive counts");
  if (model->n <= 0 || model->m <= 0 || model->dq <= 0 || model->dr <= 0)
    return bf::set_error(BF_EINVAL, "non-positive model dimension");
  if (!model->Q || !model->R) return bf::set_error(BF_EINVAL, "Q and R are required");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!carry->m_in || !carry->P_in) return bf::set_error(BF_EINVAL, "carry.m_in and carry.P_in are required");
  if (!(uparams->alpha > 0.f)) return bf::set_error(BF_EINVAL, "ParamsUKF.alpha must be positive");
  return bf::launch_agsf_ukf(model, uparams, y, u, B, T, num_components, key, opt_args, carry, out, leaf_idx, variant,
                             static_cast<hipStream_t>(stream));
}

int bf_optimal_resample_f32(const float* d_weights, const uint32_t key[2], int64_t B, int32_t M, int32_t N, int32_t* d_idx,
                            float* d_weights_out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!d_weights || !key || !d_idx || !d_weights_out) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || M <= 0) return bf::set_error(BF_EINVAL, "B and M must be positive");
  return bf::launch_optimal_resample(d_weights, key, B, M, N, d_idx, d_weights_out, static_cast<hipStream_t>(stream));
}

int bf_collapse_f32(const bf_stream* weights, const bf_stream* means, const bf_stream* covs, int64_t B, int64_t T,
                    int32_t K, int32_t n, float* mean_out, float* cov_out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!weights || !means || !weights->ptr || !means->ptr) return bf::set_error(BF_EINVAL, "weights and means are required");
  if (cov_out && (!covs || !covs->ptr)) return bf::set_error(BF_EINVAL, "cov_out needs the covariance stream");
  if (B <= 0 || T <= 0 || K <= 0 || n <= 0) return bf::set_error(BF_EINVAL, "non-positive size");
  if (!mean_out && !cov_out) return bf::set_error(BF_EINVAL, "nothing to compute");
  return bf::launch_collapse(weights, means, covs, B, T, K, n, mean_out, cov_out, static_cast<hipStream_t>(stream));
}

int bf_bpf_f32(const bf_bpf_model* model, const bf_cstream* y, const bf_cstream* u, int64_t B, int64_t T, int32_t N,
               const uint32_t key[2], float ess_threshold, int32_t resampler, const bf_bpf_carry* carry,
               const bf_bpf_out* out, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!model || !y || !out || !key) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0 || N <= 0) return bf::set_error(BF_EINVAL, "B, T and N must be positive");
  if (!y->ptr) return bf::set_error(BF_EINVAL, "observations pointer is NULL");
  if (!model->ssm.Q || !model->m0 || !model->P0 || !model->lp_cov) return bf::set_error(BF_EINVAL, "Q, m0, P0, lp_cov are required");
  if (resampler != 0 && resampler != 1) return bf::set_error(BF_EINVAL, "resampler must be 0 (multinomial) or 1 (systematic)");
  if (carry && carry->x_in && !carry->w_in) return bf::set_error(BF_EINVAL, "carry.x_in needs carry.w_in");
  return bf::launch_bpf(model, y, u, B, T, N, ess_threshold, resampler, key, carry, out, static_cast<hipStream_t>(stream));
}

int bf_sample_ssm_f32(const bf_bpf_model* model, const uint32_t* d_keys, const bf_cstream* u, int64_t B, int64_t T,
                      float* d_states, float* d_emissions, void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!model || !d_keys || (!d_states && !d_emissions)) return bf::set_error(BF_EINVAL, "NULL argument");
  if (B <= 0 || T <= 0) return bf::set_error(BF_EINVAL, "B and T must be positive");
  if (!model->ssm.Q || !model->ssm.R || !model->m0 || !model->P0) return bf::set_error(BF_EINVAL, "Q, R, m0, P0 are required");
  return bf::launch_sample_ssm(model, d_keys, u, B, T, d_states, d_emissions, static_cast<hipStream_t>(stream));
}

int bf_resample_f32(const float* d_w, const uint32_t* d_keys, int64_t B, int32_t N, int32_t resampler, int32_t* d_idx,
                    void* stream) {
  bf::CallOptionScope call_option_scope;
  if (!d_w || !d_keys || !d_idx || B <= 0 || N <= 0) return bf::set_error(BF_EINVAL, "bad argument");
  return bf::launch_resample(d_w, d_keys, B, N, resampler, d_idx, static_cast<hipStream_t>(stream));
}

int bf_canon_eval_f32(int32_t op, const float* in, int64_t n, float* out, int32_t on_device, void* stream) {
  if (op < 0 || op > 5 || !in || !out || n < 0) return bf::set_error(BF_EINVAL, "bad argument");
  if (n == 0) return BF_OK;
  if (!on_device) {
    for (int64_t i = 0; i < n; ++i) out[i] = bf::canon_eval_one(op, in, n, i);
    return BF_OK;
  }
  hipLaunchKernelGGL(bf::canon_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), op, in, n, out);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

int bf_random_normal_f32(const uint32_t key[2], int64_t count, float* host_out) {
  if (!key || !host_out || count < 0 || count > 0x7fffffff) return bf::set_error(BF_EINVAL, "bad argument");
  for (int64_t i = 0; i < count; ++i)
    host_out[i] = bf::bits_to_normal(bf::threefry_bits(key[0], key[1], (uint32_t)i, (uint32_t)count));
  return BF_OK;
}

int bf_random_split(const uint32_t key[2], int64_t num, uint32_t* host_out) {
  if (!key || !host_out || num <= 0 || num > 0x3fffffff) return bf::set_error(BF_EINVAL, "bad argument");
  for (int64_t i = 0; i < num; ++i) {
    const bf::U32x2 k = bf::threefry_split(key[0], key[1], (uint32_t)i, (uint32_t)num);
    host_out[2 * i] = k.x;
    host_out[2 * i + 1] = k.y;
  }
  return BF_OK;
}

}  // extern "C"
