// generic_scan: the Kalman / extended-Kalman / Gaussian-sum scan for ANY dimensions (run-time n, m, dq, dr, K).
//
// Same recursion as kf_scan_group.hpp / gsf_scan.hpp -- the lax.scan body of gaussian_sum_filter
// (gaussfiltax/inference.py:333-371): vmap(_condition_on) (:345 -> :72-105), reweight (:347-350), vmap(_predict)
// (:353 -> :51-70) -- for the shapes the compile-time-dimension kernels do not cover: state_dim 9 ... ~96, obs_dim
// > 4 or > state_dim, more components than one workgroup's lanes hold.  The reference's functions are
// dimension-generic (jnp on arbitrary shapes); this is the engine's counterpart, slower than the register kernels but
// never BF_EUNSUPPORTED.
//
// Mapping (gfx950).  One workgroup per trajectory.  The state of the component being advanced lives in LDS: P (n x n),
// the linearisations H_x / F_x, the products H P, (H P) H^T, X = solve(S + 1e-6, H P), K S, F P and their transposes --
// "state vectors and covariance tiles staged in LDS" as north_star puts it.  Every product is an LDS-to-LDS matrix
// multiply spread over the workgroup's lanes, one lane per 1 x 4 output block (ds_read_b128 of the right operand, a
// broadcast read of the left one; k ascending, first term a plain multiply: the oracle's and kf_math.hpp's summation
// order).  The m x m solve is an LU factorization with partial pivoting in LAPACK's getrf order (row swaps, multipliers
// left in place), cooperative over the trailing block, followed by a column-per-lane getrs (swaps, forward, backward
// substitution) -- the same per-entry operation sequence as kf_math.hpp's psd_solve.  The log-likelihood factor is a
// left-looking Cholesky, row per lane.  With K > 1 the components take turns in the LDS tile (their carried means /
// covariances live in an HBM scratch that stays L2-resident) and the weight update runs once per step over all K in the
// oracle's adjacent-pair tree order.  NT = 64 threads (one wave: the barriers are free) for n <= 16, 256 above.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "scan_common.hpp"
#include "models.hpp"
#include "bf_rng.hpp"
#include "generic_device.hpp"

namespace bf {

// ---------------------------------------------------------------------------------------------------------------
static size_t gen_lds_floats(int n, int m, int KP) {
  const int ldn = ((n + 3) & ~3) + 4, ldm = ((m + 3) & ~3) + 4, nv = (n + 3) & ~3, mvv = (m + 3) & ~3;
  const size_t upd = 3 * (size_t)m * ldn + 3 * (size_t)n * ldm + 4 * (size_t)m * ldm;
  const size_t prd = 3 * (size_t)n * ldn;
  return (size_t)n * ldn + 2 * nv + 5 * mvv + 3 * (size_t)((KP + 3) & ~3) + (upd > prd ? upd : prd);
}

// Host side: registry model -> one constant block {A, Hm, Gq0, Dr0, R, r0, GQG[steps], DRD[steps]} on the device.
static int gen_fill(const bf_model* p, long long T, GenModel& g, std::vector<float>& blk) {
  const int n = p->n, m = p->m, dq = p->dq, dr = p->dr;
  g.dyn_id = p->dyn_id; g.emi_id = p->emi_id; g.n = n; g.dq = dq; g.m = m; g.dr = dr;
  for (int i = 0; i < 8; ++i) g.dth[i] = g.eth[i] = 0.f;
  if (p->flags & (BF_MODEL_PREDICT_FIRST | BF_MODEL_LEGACY_GSF_COV))
    return set_error(BF_EUNSUPPORTED, "the legacy-class step order / covariance quirk run on the compiled (n <= 8, m <= 4) instances only");
  g.jitter = (p->flags & BF_MODEL_NO_JITTER) ? 0.0f : 1e-6f;
  std::vector<float> G((size_t)n * dq, 0.f), D((size_t)m * dr, 0.f), A((size_t)n * n, 0.f), Hm((size_t)m * n, 0.f);
  for (int i = 0; i < n && i < dq; ++i) G[(size_t)i * dq + i] = 1.f;
  for (int i = 0; i < m && i < dr; ++i) D[(size_t)i * dr + i] = 1.f;
  const float* th = p->dyn_theta;
  switch (p->dyn_id) {
    case DYN_LINEAR:
      if (p->n_dyn_theta != n * n + n * dq) return set_error(BF_EINVAL, "linear dynamics: theta must hold A and G");
      for (int i = 0; i < n * n; ++i) A[i] = th[i];
      for (int i = 0; i < n * dq; ++i) G[i] = th[n * n + i];
      break;
    case DYN_LORENZ96:
      if (p->n_dyn_theta != 5 || dq != n || n < 4) return set_error(BF_EINVAL, "lorenz96: theta = (alpha, beta, gamma, dt, mode), dq = n >= 4");
      for (int i = 0; i < 5; ++i) g.dth[i] = th[i];
      break;
    case DYN_LORENZ63:
      if (n != 3 || p->n_dyn_theta != 4 || dq != 3) return set_error(BF_EINVAL, "lorenz63: n = dq = 3, theta = (sigma, rho, beta, dt)");
      for (int i = 0; i < 4; ++i) g.dth[i] = th[i];
      break;
    case DYN_MANEUVER_BOT: {
      if (n != 4 || p->n_dyn_theta != 2 || dq != 2) return set_error(BF_EINVAL, "maneuver_bot: n = 4, dq = 2, theta = (dt, acc)");
      g.dth[0] = th[0];
      g.dth[1] = th[1];
      const float Gb[8] = {0.5f, 0, 1, 0, 0, 0.5f, 0, 1};
      for (int i = 0; i < 8; ++i) G[i] = Gb[i];
    } break;
    case DYN_SINE:
      if (p->n_dyn_theta != 1 || dq != n) return set_error(BF_EINVAL, "sine: theta = (w0), dq = n");
      g.dth[0] = th[0];
      break;
    case DYN_GROWTH:
      if (n != 1 || dq != 1) return set_error(BF_EINVAL, "growth: n = dq = 1");
      break;
    case DYN_USER:
      if (!p->user) return set_error(BF_EINVAL, "dyn_id = BF_FN_USER needs bf_model.user (bf_user_model_create)");
      for (size_t i = 0; i < G.size(); ++i) G[i] = 0.f;   // the noise bias enters through the user's function itself
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown dynamics function id %d", p->dyn_id);
  }
  th = p->emi_theta;
  switch (p->emi_id) {
    case EMI_LINEAR:
      if (p->n_emi_theta != m * n + m * dr) return set_error(BF_EINVAL, "linear emission: theta must hold H and D");
      for (int i = 0; i < m * n; ++i) Hm[i] = th[i];
      for (int i = 0; i < m * dr; ++i) D[i] = th[m * n + i];
      break;
    case EMI_BEARING_RANGE:
      if (n != 4 || m != 2 || dr != 2) return set_error(BF_EINVAL, "bearing_range: n = 4, m = dr = 2");
      break;
    case EMI_BEARING:
      if (n != 4 || m != 1 || dr != 1) return set_error(BF_EINVAL, "bearing: n = 4, m = dr = 1");
      break;
    case EMI_QUADRATIC:
      if (m != 1 || dr != 1 || p->n_emi_theta != 1) return set_error(BF_EINVAL, "quadratic: m = dr = 1, theta = (c)");
      g.eth[0] = th[0];
      break;
    case EMI_STOCH_VOL:
      if (m != n || dr != n || p->n_emi_theta != 3) return set_error(BF_EINVAL, "stoch_vol: m = dr = n, theta = (sigma, beta, c)");
      for (int i = 0; i < 3; ++i) g.eth[i] = th[i];
      break;
    case EMI_USER:
      if (!p->user) return set_error(BF_EINVAL, "emi_id = BF_FN_USER needs bf_model.user (bf_user_model_create)");
      for (size_t i = 0; i < D.size(); ++i) D[i] = 0.f;
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown emission function id %d", p->emi_id);
  }
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  if (p->R_steps > 1 && p->emi_id == EMI_STOCH_VOL)
    return set_error(BF_EUNSUPPORTED, "time-varying R needs an emission with a constant noise Jacobian H_r");
  g.q_tv = p->Q_steps > 1;
  g.r_tv = p->R_steps > 1;
  const int qs = g.q_tv ? p->Q_steps : 1, rs = g.r_tv ? p->R_steps : 1;
  // block layout (floats): A | Hm | Gq0 | Dr0 | R[rs] | r0 | GQG[qs] | DRD[rs] | q0 | Q[qs] | dyn_theta | emi_theta
  const bool udyn = p->dyn_id == DYN_USER, uemi = p->emi_id == EMI_USER;
  const size_t nR = (size_t)(uemi ? rs : 1) * dr * dr, nth_d = udyn ? (size_t)(p->n_dyn_theta > 0 ? p->n_dyn_theta : 0) : 0,
               nth_e = uemi ? (size_t)(p->n_emi_theta > 0 ? p->n_emi_theta : 0) : 0;
  const size_t oA = 0, oH = oA + (size_t)n * n, oGq = oH + (size_t)m * n, oDr = oGq + n, oR = oDr + m, or0 = oR + nR,
               oGQG = or0 + dr, oDRD = oGQG + (size_t)qs * n * n, oq0 = oDRD + (size_t)rs * m * m, oQ = oq0 + dq,
               oThd = oQ + (size_t)qs * dq * dq, oThe = oThd + nth_d + 1, total = oThe + nth_e + 1;
  blk.assign(total, 0.f);
  for (int i = 0; i < dq; ++i) blk[oq0 + i] = p->q0 ? p->q0[i] : 0.f;
  for (size_t i = 0; i < (size_t)qs * dq * dq; ++i) blk[oQ + i] = p->Q[i];
  for (size_t i = 0; i < nth_d; ++i) blk[oThd + i] = p->dyn_theta[i];
  for (size_t i = 0; i < nth_e; ++i) blk[oThe + i] = p->emi_theta[i];
  for (int i = 0; i < n * n; ++i) blk[oA + i] = A[i];
  for (int i = 0; i < m * n; ++i) blk[oH + i] = Hm[i];
  for (int i = 0; i < n; ++i) {
    float s = 0.f;
    for (int kq = 0; kq < dq; ++kq) s = fmaf(G[(size_t)i * dq + kq], p->q0 ? p->q0[kq] : 0.f, s);
    blk[oGq + i] = s;
  }
  for (int i = 0; i < m; ++i) {
    float s = 0.f;
    for (int kr = 0; kr < dr; ++kr) s = fmaf(D[(size_t)i * dr + kr], p->r0 ? p->r0[kr] : 0.f, s);
    blk[oDr + i] = s;
  }
  for (size_t i = 0; i < nR; ++i) blk[oR + i] = p->R[i];
  for (int i = 0; i < dr; ++i) blk[or0 + i] = p->r0 ? p->r0[i] : 0.f;
  // (F_q Q) F_q^T and (H_r R) H_r^T in fp32 with the association of inference.py:69, :100
  std::vector<float> tmp((size_t)(n > m ? n : m) * (dq > dr ? dq : dr));
  for (int s = 0; s < qs; ++s) {
    const float* Q = p->Q + (size_t)s * dq * dq;
    for (int i = 0; i < n; ++i)
      for (int l = 0; l < dq; ++l) {
        float v = 0.f;
        for (int kq = 0; kq < dq; ++kq) v = fmaf(G[(size_t)i * dq + kq], Q[kq * dq + l], v);
        tmp[(size_t)i * dq + l] = v;
      }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        float v = 0.f;
        for (int l = 0; l < dq; ++l) v = fmaf(tmp[(size_t)i * dq + l], G[(size_t)j * dq + l], v);
        blk[oGQG + (size_t)s * n * n + (size_t)i * n + j] = v;
      }
  }
  for (int s = 0; s < rs; ++s) {
    const float* R = p->R + (size_t)s * dr * dr;
    for (int i = 0; i < m; ++i)
      for (int l = 0; l < dr; ++l) {
        float v = 0.f;
        for (int kr = 0; kr < dr; ++kr) v = fmaf(D[(size_t)i * dr + kr], R[kr * dr + l], v);
        tmp[(size_t)i * dr + l] = v;
      }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) {
        float v = 0.f;
        for (int l = 0; l < dr; ++l) v = fmaf(tmp[(size_t)i * dr + l], D[(size_t)j * dr + l], v);
        blk[oDRD + (size_t)s * m * m + (size_t)i * m + j] = v;
      }
  }
  // offsets -> stored as pointers once the block is on the device (the caller adds the base)
  g.A = reinterpret_cast<const float*>(oA); g.Hm = reinterpret_cast<const float*>(oH); g.Gq0 = reinterpret_cast<const float*>(oGq);
  g.Dr0 = reinterpret_cast<const float*>(oDr); g.R = reinterpret_cast<const float*>(oR); g.r0 = reinterpret_cast<const float*>(or0);
  g.GQG = reinterpret_cast<const float*>(oGQG); g.DRD = reinterpret_cast<const float*>(oDRD);
  g.q0 = reinterpret_cast<const float*>(oq0); g.Q = reinterpret_cast<const float*>(oQ);
  g.dyn_theta = reinterpret_cast<const float*>(oThd); g.emi_theta = reinterpret_cast<const float*>(oThe);
  return BF_OK;
}

// one wave per workgroup (its barriers are free) up to this state / observation dimension, four waves above
static int gen_nt64_max() {
  static const int v = [] {
    const char* e = std::getenv("BAYESFILT_GENERIC_NT64_MAX");
    const int x = e ? std::atoi(e) : 32;
    return x > 0 ? x : 32;
  }();
  return v;
}

int launch_user_kernel(const bf_user_model* um, int nt, unsigned grid, size_t lds_bytes, hipStream_t stream, void** args);
int check_user_model(const bf_user_model* um, const bf_model* p);

int launch_gsf_generic(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                       const bf_carry* carry, const bf_out_desc* out, hipStream_t stream) {
  if (out->coll_mean.ptr || out->coll_cov.ptr)
    return set_error(BF_EUNSUPPORTED, "collapsed streams inside the scan need a compiled instance (n <= 8, m <= 4, K x lanes <= 256); "
                                      "use bf_collapse_f32 on the emitted streams");
  GenModel g;
  std::vector<float> blk;
  int rc = p->user ? check_user_model(p->user, p) : BF_OK;
  if (rc != BF_OK) return rc;
  rc = gen_fill(p, T, g, blk);
  if (rc != BF_OK) return rc;
  int KP = 1;
  while (KP < K) KP <<= 1;
  size_t lds_floats = gen_lds_floats(p->n, p->m, KP);
  if (p->user) {  // scratch of the dual-number linearisations (generic_device.hpp: user_dyn_linearize / user_emi_linearize)
    auto r4 = [](int v) { return (size_t)(((v + 3) & ~3) + 4); };
    const size_t N = p->n, M = p->m, DQ = p->dq, DR = p->dr;
    const size_t dynf = N * r4(p->n) + 2 * N * r4(p->dq) + DQ * r4(p->n) + DQ * r4(p->dq);
    const size_t emif = N * r4(p->n) + 2 * M * r4(p->dr) + DR * r4(p->m) + DR * r4(p->dr);
    lds_floats += dynf > emif ? dynf : emif;
  }
  const size_t lds_bytes = sizeof(float) * lds_floats;
  if (lds_bytes > 160 * 1024)
    return set_error(BF_EUNSUPPORTED, "generic scan: n = %d, m = %d, K = %d need %zu bytes of LDS (160 KiB per workgroup)", p->n, p->m, K, lds_bytes);
  const void* dv = nullptr;
  rc = device_constants(blk.data(), sizeof(float) * blk.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float* base = static_cast<const float*>(dv);
  auto fix = [&](const float*& q) { q = base + reinterpret_cast<size_t>(q); };
  fix(g.A); fix(g.Hm); fix(g.Gq0); fix(g.Dr0); fix(g.R); fix(g.r0); fix(g.GQG); fix(g.DRD);
  fix(g.q0); fix(g.Q); fix(g.dyn_theta); fix(g.emi_theta);

  // K > 1: carried means / covariances of the components that are not in the LDS tile (the caller's carry buffers
  // when given, else a stream-ordered scratch)
  float* gm = nullptr;
  float* gP = nullptr;
  float* scratch = nullptr;
  if (K > 1) {
    gm = carry->m_out;
    gP = carry->P_out;
    if (!gm || !gP) {
      const size_t fl = (size_t)B * K * ((size_t)p->n + (size_t)p->n * p->n);
      BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&scratch), sizeof(float) * fl, stream));
      if (!gm) gm = scratch;
      if (!gP) gP = scratch + (size_t)B * K * p->n;
    }
    // (m_out / P_out may alias m_in / P_in: the kernel reads the inputs at t = 0 only, component by component,
    // before it writes that component's slot)
  }
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  UViewG uv{u && u->ptr ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  hipError_t le;
  if (p->user) {  // the run-time build of the same kernel with the caller's functions compiled in (user_model.hip)
    int kp = KP;
    void* args[] = {&g, &yv, &uv, &cv, &ov, &gm, &gP, &B, &T, &K, &kp};
    const int nt = (p->n <= gen_nt64_max() && p->m <= gen_nt64_max()) ? 64 : 256;
    rc = launch_user_kernel(p->user, nt, (unsigned)B, lds_bytes, stream, args);
    if (scratch) (void)hipFreeAsync(scratch, stream);
    return rc;
  }
  if (p->n <= gen_nt64_max() && p->m <= gen_nt64_max()) {
    auto kern = gsf_generic_kernel<64>;
    if (lds_bytes > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds_bytes, stream, g, yv, uv, cv, ov, gm, gP, B, T, K, KP);
    le = hipGetLastError();
  } else {
    auto kern = gsf_generic_kernel<256>;
    if (lds_bytes > 64 * 1024) BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(256), lds_bytes, stream, g, yv, uv, cv, ov, gm, gP, B, T, K, KP);
    le = hipGetLastError();
  }
  if (scratch) (void)hipFreeAsync(scratch, stream);
  BF_HIP_CHECK(le);
  return BF_OK;
}

// The linear model of bf_kalman_filter_f32 as a registry model (DYN_LINEAR / EMI_LINEAR), K = 1.
int launch_kf_generic(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                      const bf_out_desc* out, hipStream_t stream) {
  const int n = p->n, m = p->m, dq = p->dq, dr = p->dr;
  std::vector<float> dth((size_t)n * n + (size_t)n * dq), eth((size_t)m * n + (size_t)m * dr);
  for (int i = 0; i < n * n; ++i) dth[i] = p->A[i];
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < dq; ++k) dth[(size_t)n * n + (size_t)i * dq + k] = p->G ? p->G[i * dq + k] : (i == k ? 1.f : 0.f);
  for (int i = 0; i < m * n; ++i) eth[i] = p->H[i];
  for (int i = 0; i < m; ++i)
    for (int k = 0; k < dr; ++k) eth[(size_t)m * n + (size_t)i * dr + k] = p->D ? p->D[i * dr + k] : (i == k ? 1.f : 0.f);
  bf_model mdl;
  std::memset(&mdl, 0, sizeof(mdl));
  mdl.dyn_id = DYN_LINEAR; mdl.emi_id = EMI_LINEAR; mdl.n = n; mdl.dq = dq; mdl.m = m; mdl.dr = dr;
  mdl.dyn_theta = dth.data(); mdl.n_dyn_theta = (int)dth.size(); mdl.emi_theta = eth.data(); mdl.n_emi_theta = (int)eth.size();
  mdl.q0 = p->q0; mdl.r0 = p->r0; mdl.Q = p->Q; mdl.R = p->R; mdl.flags = 0; mdl.Q_steps = p->Q_steps; mdl.R_steps = p->R_steps;
  return launch_gsf_generic(&mdl, y, nullptr, B, T, 1, carry, out, stream);
}


// ---------------------------------------------------------------------------------------------------------------
// NonlinearSSM.sample (gaussfiltax/models.py:240-289) for any dimensions: one wave per trajectory, the state in LDS,
// lane i owns entry i of every vector.  Same key schedule and the same per-entry operation order as the
// compile-time-dimension kernel (sample_ssm.hip): split three ways, z_1 ~ N(m0, P0), then (q_t, r_t) from the two
// halves of split(next_keys[t-1]); Gaussian draws as loc + chol(cov) normal(key, (d,)), k ascending.
struct GenSampleModel {
  int dyn_id, emi_id, n, dq, m, dr, g_identity, d_identity;
  float dth[8], eth[8];
  const float *A, *Gm, *Hm, *Dm, *q0, *r0, *LQ, *LR, *m0, *L0;
};

__device__ __forceinline__ void gen_mvn_draw(uint32_t k0, uint32_t k1, const float* loc, const float* L, int D, float* z, float* out,
                                             int lane) {
  for (int j = lane; j < D; j += 64) z[j] = bits_to_normal(threefry_bits(k0, k1, (uint32_t)j, (uint32_t)D));
  wave_lds_sync();
  for (int d = lane; d < D; d += 64) {
    float s = 0.f;
    for (int c = 0; c <= d; ++c) s = fmaf(L[d * D + c], z[c], s);
    out[d] = loc[d] + s;
  }
  wave_lds_sync();
}

__global__ void __launch_bounds__(64)
sample_generic_kernel(GenSampleModel p, const uint32_t* __restrict__ keys, const float* __restrict__ uptr, long long u_sB,
                      long long u_sT, float* __restrict__ states, float* __restrict__ emis, long long B, long long T) {
  const long long b = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = p.n, m = p.m, dq = p.dq, dr = p.dr;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* x = lds;
  float* xn = x + n;
  float* q = xn + n;
  float* r = q + dq;
  float* z = r + dr;  // max(n, dq, dr)
  const uint32_t k0 = keys[b * 2], k1 = keys[b * 2 + 1];
  const U32x2 key1 = threefry_split(k0, k1, 0u, 3u), key2 = threefry_split(k0, k1, 1u, 3u), key3 = threefry_split(k0, k1, 2u, 3u);
  gen_mvn_draw(key1.x, key1.y, p.m0, p.L0, n, z, x, lane);
  gen_mvn_draw(key2.x, key2.y, p.r0, p.LR, dr, z, r, lane);
  for (long long t = 0; t < T; ++t) {
    const float u0 = uptr ? uptr[b * u_sB + t * u_sT] : 0.f;
    if (t > 0) {
      const U32x2 kt = threefry_split(key3.x, key3.y, (uint32_t)(t - 1), (uint32_t)(T - 1));
      const U32x2 ka = threefry_split(kt.x, kt.y, 0u, 2u), kb = threefry_split(kt.x, kt.y, 1u, 2u);
      gen_mvn_draw(ka.x, ka.y, p.q0, p.LQ, dq, z, q, lane);
      gen_mvn_draw(kb.x, kb.y, p.r0, p.LR, dr, z, r, lane);
      for (int i = lane; i < n; i += 64) {
        float o;
        if (p.dyn_id == DYN_LINEAR) {
          o = p.A[i * n] * x[0];
          for (int k = 1; k < n; ++k) o = fmaf(p.A[i * n + k], x[k], o);
        } else if (p.dyn_id == DYN_LORENZ96) {
          const float alpha = p.dth[0], beta = p.dth[1], gamma = p.dth[2], dt = p.dth[3];
          const float ax = x[(i + n - 1) % n];
          const float bx = (p.dth[4] != 0.f) ? (x[(i + 1) % n] - x[(i + 2 * n - 2) % n]) : 0.f;
          o = x[i] + dt * (alpha * (ax * bx) - beta * x[i] + gamma);
        } else {  // DYN_SINE
          o = sinf(p.dth[0] * x[i]);
        }
        if (p.g_identity) {
          o += q[i];
        } else {
          float s = 0.f;
          for (int k = 0; k < dq; ++k) s = fmaf(p.Gm[i * dq + k], q[k], s);
          o += s;
        }
        xn[i] = o;
      }
      wave_lds_sync();
      for (int i = lane; i < n; i += 64) x[i] = xn[i];
      wave_lds_sync();
    }
    for (int a = lane; a < m; a += 64) {
      float hx;
      if (p.emi_id == EMI_STOCH_VOL) {
        const float sigma = p.eth[0], beta = p.eth[1], c = p.eth[2];
        hx = u0 * beta * expf(x[a] / sigma) * r[a] + (1.f - u0) * (c * x[a] + r[a]);
      } else {
        if (p.emi_id == EMI_LINEAR) {
          hx = p.Hm[a * n] * x[0];
          for (int k = 1; k < n; ++k) hx = fmaf(p.Hm[a * n + k], x[k], hx);
        } else {  // EMI_QUADRATIC (m = 1)
          float s = 0.f;
          for (int i = 0; i < n; ++i) s = fmaf(x[i], x[i], s);
          hx = p.eth[0] * s;
        }
        hx += 0.f;  // the emission bias at r_eval = 0 (the compiled kernel adds hb = H_r 0)
        if (p.d_identity) {
          hx += r[a];
        } else {
          float s = 0.f;
          for (int c = 0; c < dr; ++c) s = fmaf(p.Dm[a * dr + c], r[c], s);
          hx += s;
        }
      }
      if (emis) emis[(b * T + t) * m + a] = hx;
    }
    if (states) for (int i = lane; i < n; i += 64) states[(b * T + t) * n + i] = x[i];
  }
}

int launch_sample_generic(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T,
                          float* d_states, float* d_emis, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
  const int n = p->n, m = p->m, dq = p->dq, dr = p->dr;
  if (p->Q_steps > 1 || p->R_steps > 1)
    return set_error(BF_EUNSUPPORTED, "time-varying Q/R are not supported by the data generator");
  if (p->dyn_id != DYN_LINEAR && p->dyn_id != DYN_LORENZ96 && p->dyn_id != DYN_SINE)
    return set_error(BF_EUNSUPPORTED, "sample_ssm: dynamics id %d at (n=%d, dq=%d, m=%d) is not compiled in", p->dyn_id, n, dq, m);
  if (p->emi_id != EMI_LINEAR && p->emi_id != EMI_QUADRATIC && p->emi_id != EMI_STOCH_VOL)
    return set_error(BF_EUNSUPPORTED, "sample_ssm: emission id %d at (n=%d, dq=%d, m=%d) is not compiled in", p->emi_id, n, dq, m);
  GenSampleModel g;
  std::memset(&g, 0, sizeof(g));
  g.dyn_id = p->dyn_id; g.emi_id = p->emi_id; g.n = n; g.dq = dq; g.m = m; g.dr = dr;
  g.g_identity = 1;
  g.d_identity = 1;
  // block: A | Gm | Hm | Dm | q0 | r0 | LQ | LR | m0 | L0
  const size_t oA = 0, oG = oA + (size_t)n * n, oH = oG + (size_t)n * dq, oD = oH + (size_t)m * n, oq = oD + (size_t)m * dr, or_ = oq + dq,
               oLQ = or_ + dr, oLR = oLQ + (size_t)dq * dq, om0 = oLR + (size_t)dr * dr, oL0 = om0 + n, total = oL0 + (size_t)n * n;
  std::vector<float> blk(total, 0.f);
  const float* th = p->dyn_theta;
  if (p->dyn_id == DYN_LINEAR) {
    if (p->n_dyn_theta != n * n + n * dq) return set_error(BF_EINVAL, "linear dynamics: theta must hold A and G");
    for (int i = 0; i < n * n; ++i) blk[oA + i] = th[i];
    for (int i = 0; i < n * dq; ++i) blk[oG + i] = th[n * n + i];
    g.g_identity = 0;
  } else if (p->dyn_id == DYN_LORENZ96) {
    if (p->n_dyn_theta != 5 || dq != n || n < 4) return set_error(BF_EINVAL, "lorenz96: theta = (alpha, beta, gamma, dt, mode), dq = n >= 4");
    for (int i = 0; i < 5; ++i) g.dth[i] = th[i];
  } else {
    if (p->n_dyn_theta != 1 || dq != n) return set_error(BF_EINVAL, "sine: theta = (w0), dq = n");
    g.dth[0] = th[0];
  }
  th = p->emi_theta;
  if (p->emi_id == EMI_LINEAR) {
    if (p->n_emi_theta != m * n + m * dr) return set_error(BF_EINVAL, "linear emission: theta must hold H and D");
    for (int i = 0; i < m * n; ++i) blk[oH + i] = th[i];
    for (int i = 0; i < m * dr; ++i) blk[oD + i] = th[m * n + i];
    g.d_identity = 0;
  } else if (p->emi_id == EMI_QUADRATIC) {
    if (m != 1 || dr != 1 || p->n_emi_theta != 1) return set_error(BF_EINVAL, "quadratic: m = dr = 1, theta = (c)");
    g.eth[0] = th[0];
  } else {
    if (m != n || dr != n || p->n_emi_theta != 3) return set_error(BF_EINVAL, "stoch_vol: m = dr = n, theta = (sigma, beta, c)");
    for (int i = 0; i < 3; ++i) g.eth[i] = th[i];
  }
  for (int i = 0; i < dq; ++i) blk[oq + i] = p->q0 ? p->q0[i] : 0.f;
  for (int i = 0; i < dr; ++i) blk[or_ + i] = p->r0 ? p->r0[i] : 0.f;
  auto chol = [](const float* A, int d, float* L) {  // fp32, row-major, lower; same loop as ssm_device.hpp: cholesky_lower
    for (int j = 0; j < d; ++j) {
      float dd = A[j * d + j];
      for (int k = 0; k < j; ++k) dd -= L[j * d + k] * L[j * d + k];
      if (!(dd > 0.f)) return -1;
      dd = sqrtf(dd);
      L[j * d + j] = dd;
      for (int i = j + 1; i < d; ++i) {
        float s = A[i * d + j];
        for (int k = 0; k < j; ++k) s -= L[i * d + k] * L[j * d + k];
        L[i * d + j] = s / dd;
      }
    }
    return 0;
  };
  if (chol(p->Q, dq, &blk[oLQ]) != 0) return set_error(BF_EINVAL, "dynamics noise covariance is not positive definite");
  if (chol(p->R, dr, &blk[oLR]) != 0) return set_error(BF_EINVAL, "emission noise covariance is not positive definite");
  for (int i = 0; i < n; ++i) blk[om0 + i] = bp->m0[i];
  if (chol(bp->P0, n, &blk[oL0]) != 0) return set_error(BF_EINVAL, "initial covariance is not positive definite");
  const void* dv = nullptr;
  const int rc = device_constants(blk.data(), sizeof(float) * blk.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float* base = static_cast<const float*>(dv);
  g.A = base + oA; g.Gm = base + oG; g.Hm = base + oH; g.Dm = base + oD; g.q0 = base + oq; g.r0 = base + or_;
  g.LQ = base + oLQ; g.LR = base + oLR; g.m0 = base + om0; g.L0 = base + oL0;
  int zmax = n > dq ? n : dq;
  zmax = zmax > dr ? zmax : dr;
  const size_t lds_bytes = sizeof(float) * (size_t)(2 * n + dq + dr + zmax);
  if (lds_bytes > 64 * 1024) return set_error(BF_EUNSUPPORTED, "sample_ssm: state too large");
  hipLaunchKernelGGL(sample_generic_kernel, dim3((unsigned)B), dim3(64), lds_bytes, stream, g, d_keys, (u && u->ptr) ? u->ptr : nullptr,
                     u ? u->sB : 0, u ? u->sT : 0, d_states, d_emis, B, T);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
