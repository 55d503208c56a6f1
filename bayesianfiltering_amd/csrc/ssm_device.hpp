// Device-side state-space model for the sampling kernels (particle filter, data generator):
// the registry functions' VALUES (no Jacobians), the Cholesky factors the Gaussian draws and the
// emission log-density need, and the host code that fills them from the C-ABI structs.
#pragma once
#ifndef BF_JIT
#include <cstring>
#include <vector>
#include "bf_common.hpp"
#endif
#include <type_traits>
#include "bf_rng.hpp"
#include "kf_math.hpp"
#include "models.hpp"
#include "bf_canon_math.hpp"

namespace bf {

template <int N, int DQ, int M>
struct BpfModel {
  int dyn_id, emi_id, g_identity, lq_diag;  // lq_diag: chol(Q) is diagonal (its zero entries are skipped)
  int lr_diag, h_pick, pad0_, pad1_;        // lr_diag: chol(R_lp) diagonal; h_pick: linear emission whose row a is e_{2a}
  float dth[8], eth[8];
  float A[N * N];     // linear dynamics
  float Gm[N * DQ];   // F_q (noise input matrix); identity when g_identity
  float Hm[M * N];    // linear emission
  float q0[DQ];
  float LQ[DQ * DQ];  // chol(Q), lower
  float LQd[DQ];      // its diagonal, contiguous (one wide scalar load when lq_diag)
  float hb[M];        // emission bias term evaluated at r_eval (H_r r_eval)
  float LR[M * M];    // chol(R_lp), lower, of the log-prob covariance
  float rdLR[M];      // 1 / diag(LR)
  float lp_const;     // -0.5 m log(2 pi) - sum log diag(LR)
  float m0[N];
  float L0[N * N];    // chol(P0), lower
  // functions compiled from the caller's source (user_model.hip): their parameter vectors and the noise value the emission
  // density's mean is evaluated at (h(x, r_eval, u))
  float uth_dyn[64], uth_emi[64], uth_lp[64], r_eval[64];
};
enum { DYN_USER_SRC = 100, EMI_USER_SRC = 100 };   // BF_FN_USER

// What the sampling code reads from a model's STRUCTURE fields (function ids, "this factor is diagonal" flags): at run time
// from the struct (SpecRuntime: one binary serves every registry model), or as compile-time constants (SpecFixed: the
// instance a launch selects when the host-side model matches -- same arithmetic, operation for operation, but no per-particle
// switch, none of the other models' code or registers, and only the fields that model reads).
struct SpecRuntime {
  static constexpr bool fixed = false;
  static constexpr bool user_dyn = false, user_emi = false, user_lp = false;
};
// run-time structure flags plus functions from the caller's source (hiprtc builds only): f(x, q, u), h(x, r, u) and / or the
// emission log-density itself
template <bool UD, bool UE, bool ULP>
struct SpecUser {
  static constexpr bool fixed = false;
  static constexpr bool user_dyn = UD, user_emi = UE, user_lp = ULP;
};
template <int DYN, int EMI, bool G_ID, bool LQ_DIAG, bool LR_DIAG, bool H_PICK, int IMPL = 0>
struct SpecFixed {
  static constexpr bool fixed = true;
  static constexpr bool user_dyn = false, user_emi = false, user_lp = false;
  static constexpr int impl = IMPL;   // 0: Threefry + normals as the hand-scheduled block of bf_rng.hpp; 1: plain C++
  static constexpr int dyn_id = DYN, emi_id = EMI;
  static constexpr bool g_identity = G_ID, lq_diag = LQ_DIAG, lr_diag = LR_DIAG, h_pick = H_PICK;
};
#define BF_SPEC_GET(NAME_, TYPE_)                                              \
  template <class SP, class MDL>                                               \
  __host__ __device__ __forceinline__ TYPE_ spec_##NAME_(const MDL& p) {       \
    if constexpr (SP::fixed) return SP::NAME_;                                 \
    else return p.NAME_;                                                       \
  }
BF_SPEC_GET(dyn_id, int)
BF_SPEC_GET(emi_id, int)
BF_SPEC_GET(g_identity, bool)
BF_SPEC_GET(lq_diag, bool)
BF_SPEC_GET(lr_diag, bool)
BF_SPEC_GET(h_pick, bool)
#undef BF_SPEC_GET

// noise-free part g(x, u) of the registry dynamics f(x, q, u) = g(x, u) + F_q q; MDL is any model struct with the fields
// dyn_id, dth, A
template <int N, int DQ, class MDL, class SP = SpecRuntime>
__device__ __forceinline__ void dyn_base_t(const MDL& p, const float* x, float u0, float* out) {
  // canonical arithmetic (bf_canon_math.hpp): every operation rounded on its own, in the order written -- the order
  // of the test oracle's NumPy expressions -- and fused multiply-adds only where spelled out (mv)
#pragma clang fp contract(off)
  switch (spec_dyn_id<SP>(p)) {
    case DYN_LINEAR: mv<N, N>(p.A, x, out); break;
    case DYN_LORENZ96: {
      const float alpha = p.dth[0], beta = p.dth[1], gamma = p.dth[2], dt = p.dth[3];
      const bool mp = p.dth[4] != 0.f;
      BF_UNROLL for (int i = 0; i < N; ++i) {
        const float ax = x[(i + N - 1) % N];
        const float bx = mp ? (x[(i + 1) % N] - x[(i + 2 * N - 2) % N]) : 0.f;
        out[i] = x[i] + dt * (alpha * (ax * bx) - beta * x[i] + gamma);
      }
    } break;
    case DYN_LORENZ63:
      if constexpr (N == 3) {
        const float s = p.dth[0], r = p.dth[1], b = p.dth[2], dt = p.dth[3];
        out[0] = dt * s * (x[1] - x[0]) + x[0];
        out[1] = dt * (x[0] * r - x[1] - x[0] * x[2]) + x[1];
        out[2] = dt * (x[0] * x[1] - b * x[2]) + x[2];
      }
      break;
    case DYN_MANEUVER_BOT:
      if constexpr (N == 4) {
        const float dt = p.dth[0], acc = p.dth[1];
        const float c0 = 0.5f * (u0 - 1.f) * (u0 - 2.f), c1 = -u0 * (u0 - 2.f), c2 = 0.5f * u0 * (u0 - 1.f);
        float Mx[16] = {c0, c0 * dt, 0, 0, 0, c0, 0, 0, 0, 0, c0, c0 * dt, 0, 0, 0, c0};
        const float nrm = sqrtf(x[1] * x[1] + x[3] * x[3]);
        // the two turn matrices differ in the sign of the angle only: one sincos serves both (sin is odd, cos even)
        float sn0, cs0;
        canon_sincos(dt * (0.1f * acc / nrm), &sn0, &cs0);
        BF_UNROLL for (int sgn = 0; sgn < 2; ++sgn) {
          const float cc = sgn == 0 ? c1 : c2;
          const float om = 0.1f * (sgn == 0 ? acc : -acc) / nrm;
          const float sn = sgn == 0 ? sn0 : -sn0, cs = cs0;
          const float so = sn / om, co = (1.f - cs) / om;
          const float Fm[16] = {1, so, 0, -co, 0, cs, 0, -sn, 0, co, 1, so, 0, sn, 0, cs};
          BF_UNROLL for (int i = 0; i < 16; ++i) Mx[i] += cc * Fm[i];
        }
        mv<4, 4>(Mx, x, out);
      }
      break;
    case DYN_SINE: BF_UNROLL for (int i = 0; i < N; ++i) out[i] = canon_sin(p.dth[0] * x[i]); break;
    case DYN_GROWTH:
      if constexpr (N == 1) out[0] = x[0] / 2.0f + 25.0f * x[0] / (1.f + x[0] * x[0]) + u0;
      break;
    default: BF_UNROLL for (int i = 0; i < N; ++i) out[i] = x[i]; break;
  }
}

// f(x, q, u) = g(x, u) + F_q q: the noise-free part above, then the noise through F_q (identity for most registry functions)
template <int N, int DQ, class MDL, class SP = SpecRuntime>
__device__ __forceinline__ void dyn_value_t(const MDL& p, const float* x, const float* q, float u0, float* out) {
#pragma clang fp contract(off)
#ifdef BF_USER_DYN
  if constexpr (SP::user_dyn) {   // the caller's f(x, q, u): the noise enters however the function says (models.py:82-84)
    bfu::dynamics<float>(x, q, u0, p.uth_dyn, out);
    return;
  }
#endif
  dyn_base_t<N, DQ, MDL, SP>(p, x, u0, out);
  if (spec_g_identity<SP>(p)) {
    if constexpr (DQ == N) BF_UNROLL for (int i = 0; i < N; ++i) out[i] += q[i];
  } else {
    BF_UNROLL for (int i = 0; i < N; ++i) {
      float s = 0.f;
      BF_UNROLL for (int k = 0; k < DQ; ++k) s = fmaf(p.Gm[i * DQ + k], q[k], s);
      out[i] += s;
    }
  }
}

template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ void dyn_value(const BpfModel<N, DQ, M>& p, const float* x, const float* q, float u0, float* out) {
  dyn_value_t<N, DQ, BpfModel<N, DQ, M>, SP>(p, x, q, u0, out);
}

// noise-free part g(x, u) of the registry emissions h(x, r, u) = g(x, u) + H_r r (constant H_r); MDL is any
// model struct with the fields emi_id, eth, Hm
template <int N, int M, class MDL, class SP = SpecRuntime>
__device__ __forceinline__ void emi_mean_t(const MDL& p, const float* x, float u0, float* hx) {
#pragma clang fp contract(off)
  switch (spec_emi_id<SP>(p)) {
    case EMI_LINEAR: mv<M, N>(p.Hm, x, hx); break;
    case EMI_BEARING_RANGE:
      if constexpr (N == 4 && M == 2) {
        hx[0] = canon_atan2(x[2], x[0]);
        hx[1] = sqrtf(x[0] * x[0] + x[2] * x[2]);
      }
      break;
    case EMI_BEARING:
      if constexpr (N == 4 && M == 1) hx[0] = canon_atan2(x[2], x[0]);
      break;
    case EMI_QUADRATIC:
      if constexpr (M == 1) {
        float s = 0.f;
        BF_UNROLL for (int i = 0; i < N; ++i) s = fmaf(x[i], x[i], s);
        hx[0] = p.eth[0] * s;
      }
      break;
    default: BF_UNROLL for (int a = 0; a < M; ++a) hx[a] = 0.f; break;
  }
}

// The stochastic-volatility emission h(x, r, u) = M(x, u) r + (1 - u) c x with M = diag(u beta exp(x / sigma) + (1 - u))
// (glmsv, docs/experiments/adaptive_experiment.py:54) has the state-dependent log-density
// MVN(h(x, r_eval, u), M R M^T).log_prob(y) (lmsvlp, :55-57).  chol(M R M^T) = M chol(R) for the positive diagonal M:
// the residual is divided by the diagonal of M before the constant factor's forward substitution, and sum log M_ii
// joins the log-determinant.
template <class MDL>
__device__ __forceinline__ float sv_scale(const MDL& p, float xa, float u0) {
#pragma clang fp contract(off)
  return u0 * p.eth[1] * canon_exp(xa / p.eth[0]) + (1.f - u0);
}

// mean of the emission density: h(x, r_eval, u) for the registry emissions with constant H_r (hb = H_r r_eval), and
// for the stochastic-volatility emission (hb = r_eval)
template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ void emi_value(const BpfModel<N, DQ, M>& p, const float* x, float u0, float* hx) {
#pragma clang fp contract(off)
  if constexpr (N == M) {
    if (spec_emi_id<SP>(p) == EMI_STOCH_VOL) {
      BF_UNROLL for (int a = 0; a < M; ++a)
        hx[a] = u0 * p.eth[1] * canon_exp(x[a] / p.eth[0]) * p.hb[a] + (1.f - u0) * (p.eth[2] * x[a] + p.hb[a]);
      return;
    }
  }
  emi_mean_t<N, M, BpfModel<N, DQ, M>, SP>(p, x, u0, hx);
  BF_UNROLL for (int a = 0; a < M; ++a) hx[a] += p.hb[a];
}

template <int J, int H, class F>
__device__ __forceinline__ void ssm_static_for(F&& f) {
  if constexpr (J < H) {
    f(std::integral_constant<int, J>{});
    ssm_static_for<J + 1, H>(f);
  }
}

// q = q0 + chol(Q) normal(key_i, (dq,)) (gaussfiltax/models.py:82-83): accumulated column by column as the normals
// arrive (per entry the same fma chain as the row-wise product, c ascending from 0), so no z vector stays live.
// normal(key, (dq,)): Threefry block j yields entries j and h + j.
template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ void draw_dynamics_noise(const BpfModel<N, DQ, M>& mdl, U32x2 ki, float* q) {
  constexpr int h = (DQ + 1) / 2;
  const bool lq_diag = spec_lq_diag<SP>(mdl);
  float zhi[h];
  BF_UNROLL for (int d = 0; d < DQ; ++d) q[d] = 0.f;
  const uint32_t ks2 = ki.x ^ ki.y ^ 0x1BD11BDAu;
  ssm_static_for<0, h>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    // the hand-scheduled block of bf_rng.hpp: threefry2x32(ki, j, h + j) and bits_to_normal of both words, same bits
    float zj, zh;
    threefry_two_normals_gfx950<j, (h + j < DQ) ? h + j : 0>(ki.x, ki.y, ks2, zj, zh);
    zhi[j] = (h + j < DQ) ? zh : 0.f;
    if (lq_diag) q[j] = mdl.LQ[j * DQ + j] * zj;
    else BF_UNROLL for (int d = j; d < DQ; ++d) q[d] = __builtin_fmaf(mdl.LQ[d * DQ + j], zj, q[d]);
  });
  BF_UNROLL for (int j = 0; h + j < DQ; ++j) {
    if (lq_diag) q[h + j] = mdl.LQ[(h + j) * DQ + h + j] * zhi[j];
    else BF_UNROLL for (int d = h + j; d < DQ; ++d) q[d] = __builtin_fmaf(mdl.LQ[d * DQ + h + j], zhi[j], q[d]);
  }
  BF_UNROLL for (int d = 0; d < DQ; ++d) q[d] = mdl.q0[d] + q[d];
}

// MVN(h(x, r_eval, u), R_lp).log_prob(y) (the `*lp` functions of the reference's scripts, e.g. nonlinearities.py:51-52)
// through the Cholesky factor: forward substitution by fma, multiplication by the reciprocal diagonal, the quadratic
// form by fma, -0.5 quad + const as ONE fma.  The oracle restates exactly this sequence (gaussfilt_oracle.py,
// arith = "canonical").
template <int N, int DQ, int M, class SP = SpecRuntime>
__device__ __forceinline__ float emission_loglik(const BpfModel<N, DQ, M>& mdl, const float* xn, float u0, const float* yv) {
#pragma clang fp contract(off)
#ifdef BF_USER_LP
  if constexpr (SP::user_lp) return bfu::log_prob<float>(xn, yv, u0, mdl.uth_lp);   // the caller's emission_distribution_log_prob
#endif
  float hx[M], zz[M];
#ifdef BF_USER_EMI
  if constexpr (SP::user_emi) {
    bfu::emission<float>(xn, mdl.r_eval, u0, mdl.uth_emi, hx);   // mean of the density: h(x, r_eval, u)
  } else
#endif
  if (spec_h_pick<SP>(mdl)) {  // selection emission (e.g. the even states of Lorenz-96): the exact-zero terms of H x are skipped
    BF_UNROLL for (int a = 0; a < M; ++a) hx[a] = xn[(2 * a) % N] + mdl.hb[a];
  } else {
    emi_value<N, DQ, M, SP>(mdl, xn, u0, hx);
  }
  float quad = 0.f, lsc = 0.f;
  BF_UNROLL for (int a = 0; a < M; ++a) {
    float s = yv[a] - hx[a];
    if constexpr (N == M) {
      if (spec_emi_id<SP>(mdl) == EMI_STOCH_VOL) {  // state-dependent covariance M R M^T: chol(M R M^T) = M chol(R)
        const float d = sv_scale(mdl, xn[a], u0);
        s = s / d;
        lsc = lsc + canon_log(d);
      }
    }
    if (!spec_lr_diag<SP>(mdl)) BF_UNROLL for (int c = 0; c < a; ++c) s = __builtin_fmaf(-mdl.LR[a * M + c], zz[c], s);
    zz[a] = s * mdl.rdLR[a];
    quad = __builtin_fmaf(zz[a], zz[a], quad);
  }
  return __builtin_fmaf(-0.5f, quad, mdl.lp_const) - lsc;
}

#ifndef BF_JIT   // host side: the C-ABI structs -> BpfModel
static inline int cholesky_lower(const float* A, int n, float* L) {  // fp32, row-major; returns 0 or -1 (not PD)
#pragma clang fp contract(off)  // every operation rounded on its own: what the test oracle's NumPy loop does
  for (int i = 0; i < n * n; ++i) L[i] = 0.f;
  for (int j = 0; j < n; ++j) {
    float d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.f)) return -1;
    d = sqrtf(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      float s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  return 0;
}

// The fields of a BpfModel<N, DQ, M> by address, for dimensions known at run time: the struct holds 4-byte members only, in
// declaration order and without padding, so the model of a kernel compiled at run time (user_model.hip) is the same words laid
// out one after the other (bpf_model_view_flat) -- and ONE fill routine serves both.
struct BpfModelView {
  int N, DQ, M;
  int *dyn_id, *emi_id, *g_identity, *lq_diag, *lr_diag, *h_pick;
  float *dth, *eth, *A, *Gm, *Hm, *q0, *LQ, *LQd, *hb, *LR, *rdLR, *lp_const, *m0, *L0, *uth_dyn, *uth_emi, *uth_lp, *r_eval;
};
constexpr size_t bpf_model_words(int n, int dq, int m) {
  return 8 + 16 + (size_t)n * n + (size_t)n * dq + (size_t)m * n + dq + (size_t)dq * dq + dq + m + (size_t)m * m + m + 1 + n + (size_t)n * n + 4 * 64;
}
static inline BpfModelView bpf_model_view_flat(uint32_t* w, int n, int dq, int m) {
  BpfModelView v;
  v.N = n; v.DQ = dq; v.M = m;
  int* iw = reinterpret_cast<int*>(w);
  v.dyn_id = iw; v.emi_id = iw + 1; v.g_identity = iw + 2; v.lq_diag = iw + 3; v.lr_diag = iw + 4; v.h_pick = iw + 5;
  float* f = reinterpret_cast<float*>(w) + 8;
  auto take = [&](size_t k) { float* r = f; f += k; return r; };
  v.dth = take(8); v.eth = take(8); v.A = take((size_t)n * n); v.Gm = take((size_t)n * dq); v.Hm = take((size_t)m * n); v.q0 = take(dq);
  v.LQ = take((size_t)dq * dq); v.LQd = take(dq); v.hb = take(m); v.LR = take((size_t)m * m); v.rdLR = take(m); v.lp_const = take(1);
  v.m0 = take(n); v.L0 = take((size_t)n * n); v.uth_dyn = take(64); v.uth_emi = take(64); v.uth_lp = take(64); v.r_eval = take(64);
  return v;
}

// user_flags: bit 0 = dynamics, bit 1 = emission, bit 2 = log-density come from the caller's source (their parameter vectors:
// p->dyn_theta, p->emi_theta, bp->lp_theta; at most 64 entries each)
static inline int fill_bpf_model_view(const bf_bpf_model* bp, BpfModelView e, int user_flags, const float* lp_theta, int n_lp_theta) {
#pragma clang fp contract(off)
  const bf_model* p = &bp->ssm;
  const int N = e.N, DQ = e.DQ, M = e.M;
  if (p->Q_steps > 1 || p->R_steps > 1)
    return set_error(BF_EUNSUPPORTED, "time-varying Q/R are not supported by the sampling kernels (particle filter, data generator)");
  *e.dyn_id = p->dyn_id;
  *e.emi_id = p->emi_id;
  *e.g_identity = 1;
  const float* th = p->dyn_theta;
  if (user_flags & 1) {
    if (p->n_dyn_theta > 64) return set_error(BF_EUNSUPPORTED, "a dynamics function from source takes at most 64 parameters here");
    for (int i = 0; i < p->n_dyn_theta; ++i) e.uth_dyn[i] = th[i];
  } else switch (p->dyn_id) {
    case DYN_LINEAR:
      if (p->n_dyn_theta != N * N + N * DQ) return set_error(BF_EINVAL, "linear dynamics: theta must hold A and G");
      for (int i = 0; i < N * N; ++i) e.A[i] = th[i];
      for (int i = 0; i < N * DQ; ++i) e.Gm[i] = th[N * N + i];
      *e.g_identity = 0;
      break;
    case DYN_LORENZ96:
      if (p->n_dyn_theta != 5 || DQ != N) return set_error(BF_EINVAL, "lorenz96: theta = (alpha, beta, gamma, dt, mode), dq = n");
      for (int i = 0; i < 5; ++i) e.dth[i] = th[i];
      break;
    case DYN_LORENZ63:
      if (N != 3 || p->n_dyn_theta != 4 || DQ != 3) return set_error(BF_EINVAL, "lorenz63: n = dq = 3");
      for (int i = 0; i < 4; ++i) e.dth[i] = th[i];
      break;
    case DYN_MANEUVER_BOT: {
      if (N != 4 || p->n_dyn_theta != 2 || DQ != 2) return set_error(BF_EINVAL, "maneuver_bot: n = 4, dq = 2");
      e.dth[0] = th[0];
      e.dth[1] = th[1];
      const float Gb[8] = {0.5f, 0, 1, 0, 0, 0.5f, 0, 1};
      for (int i = 0; i < 8 && i < N * DQ; ++i) e.Gm[i] = Gb[i];
      *e.g_identity = 0;
    } break;
    case DYN_SINE:
      if (p->n_dyn_theta != 1 || DQ != N) return set_error(BF_EINVAL, "sine: theta = (w0), dq = n");
      e.dth[0] = th[0];
      break;
    case DYN_GROWTH:
      if (N != 1 || DQ != 1) return set_error(BF_EINVAL, "growth: n = dq = 1");
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown dynamics function id %d", p->dyn_id);
  }
  th = p->emi_theta;
  const int dr = p->dr;
  if (dr > 64) return set_error(BF_EUNSUPPORTED, "emission noise dimension > 64");
  std::vector<float> D((size_t)M * dr, 0.f);
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < dr; ++k) D[i * dr + k] = (i == k) ? 1.f : 0.f;
  for (int k = 0; k < dr; ++k) e.r_eval[k] = bp->r_eval ? bp->r_eval[k] : 0.f;
  if (user_flags & 4) {
    if (n_lp_theta > 64) return set_error(BF_EUNSUPPORTED, "a log-density function from source takes at most 64 parameters here");
    for (int i = 0; i < n_lp_theta; ++i) e.uth_lp[i] = lp_theta[i];
  }
  if (user_flags & 2) {
    if (p->n_emi_theta > 64) return set_error(BF_EUNSUPPORTED, "an emission function from source takes at most 64 parameters here");
    for (int i = 0; i < p->n_emi_theta; ++i) e.uth_emi[i] = th[i];
  } else if (user_flags & 4) {
    // the density is the caller's own function: the emission function plays no part in the particle filter
  } else switch (p->emi_id) {
    case EMI_LINEAR:
      if (p->n_emi_theta != M * N + M * dr) return set_error(BF_EINVAL, "linear emission: theta must hold H and D");
      for (int i = 0; i < M * N; ++i) e.Hm[i] = th[i];
      for (int i = 0; i < M * dr; ++i) D[i] = th[M * N + i];
      break;
    case EMI_BEARING_RANGE:
      if (N != 4 || M != 2 || dr != 2) return set_error(BF_EINVAL, "bearing_range: n = 4, m = dr = 2");
      break;
    case EMI_BEARING:
      if (N != 4 || M != 1 || dr != 1) return set_error(BF_EINVAL, "bearing: n = 4, m = dr = 1");
      break;
    case EMI_QUADRATIC:
      if (M != 1 || dr != 1 || p->n_emi_theta != 1) return set_error(BF_EINVAL, "quadratic: m = dr = 1");
      e.eth[0] = th[0];
      break;
    case EMI_STOCH_VOL:
      if (N != M || dr != M || p->n_emi_theta != 3) return set_error(BF_EINVAL, "stoch_vol: n = m = dr, theta = (sigma, beta, c)");
      for (int i = 0; i < 3; ++i) e.eth[i] = th[i];
      for (int i = 0; i < M * dr; ++i) D[i] = (i / dr == i % dr) ? 1.f : 0.f;  // hb = r_eval itself
      break;
    default:
      return set_error(BF_EUNSUPPORTED, "emission function id %d has no Gaussian log-density on the device", p->emi_id);
  }
  for (int i = 0; i < M; ++i) {
    float s = 0.f;
    for (int k = 0; k < dr; ++k) s = fmaf(D[i * dr + k], bp->r_eval ? bp->r_eval[k] : 0.f, s);
    e.hb[i] = s;
  }
  for (int i = 0; i < DQ; ++i) e.q0[i] = p->q0 ? p->q0[i] : 0.f;
  if (cholesky_lower(p->Q, DQ, e.LQ) != 0) return set_error(BF_EINVAL, "dynamics noise covariance is not positive definite");
  *e.lq_diag = 1;
  for (int i = 0; i < DQ; ++i) {
    e.LQd[i] = e.LQ[i * DQ + i];
    for (int k = 0; k < i; ++k)
      if (e.LQ[i * DQ + k] != 0.f) *e.lq_diag = 0;
  }
  *e.lr_diag = 1;
  *e.h_pick = 0;
  if (!(user_flags & 4)) {   // Gaussian density MVN(h(x, r_eval, u), lp_cov): its Cholesky factor and constant
    if (cholesky_lower(bp->lp_cov, M, e.LR) != 0) return set_error(BF_EINVAL, "log-prob covariance is not positive definite");
    for (int i = 0; i < M; ++i)
      for (int k = 0; k < i; ++k)
        if (e.LR[i * M + k] != 0.f) *e.lr_diag = 0;
    *e.h_pick = (!(user_flags & 2) && p->emi_id == EMI_LINEAR && 2 * M <= N + 1) ? 1 : 0;
    if (*e.h_pick)
      for (int a = 0; a < M; ++a)
        for (int i = 0; i < N; ++i)
          if (e.Hm[a * N + i] != ((i == 2 * a) ? 1.0f : 0.0f)) *e.h_pick = 0;
    float logdet = 0.f;
    for (int i = 0; i < M; ++i) {
      e.rdLR[i] = 1.0f / e.LR[i * M + i];
      logdet += canon_log(e.LR[i * M + i]);
    }
    *e.lp_const = -0.5f * (float)M * 1.8378770664093453f - logdet;
  }
  for (int i = 0; i < N; ++i) e.m0[i] = bp->m0[i];
  if (cholesky_lower(bp->P0, N, e.L0) != 0) return set_error(BF_EINVAL, "initial covariance is not positive definite");
  return BF_OK;
}

template <int N, int DQ, int M>
static inline int fill_bpf_model(const bf_bpf_model* bp, BpfModel<N, DQ, M>& e) {
  static_assert(sizeof(BpfModel<N, DQ, M>) == 4 * bpf_model_words(N, DQ, M), "BpfModel: 4-byte members in declaration order, no padding");
  std::memset(&e, 0, sizeof(e));
  return fill_bpf_model_view(bp, bpf_model_view_flat(reinterpret_cast<uint32_t*>(&e), N, DQ, M), 0, nullptr, 0);
}

#endif  // BF_JIT

}  // namespace bf
