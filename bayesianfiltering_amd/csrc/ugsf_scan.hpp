// ugsf_scan: batched unscented Gaussian-sum filter = bank of K unscented Kalman filters (non-additive
// noise, augmented sigma points) + weight update.
//
// Replaces the lax.scan body of unscented_gaussian_sum_filter (gaussfiltax/inference.py:379-456):
//   vmap(_ukf_condition_on_nonadditive) over components   :421 -> :198-224
//   reweight                                              :424-427
//   vmap(_ukf_predict_nonadditive)                        :430 -> :146-174
// with the sigma points of utils._get_sigma_points (utils.py:247-254): the 2 L rows
//   mA +- sqrt(L + lambda) * sqrtm(PA)[j, :],  mA = (m, noise bias),  PA = blockdiag(P, noise covariance),
// L = n + noise_dim, lambda = alpha^2 (L + kappa) - L.  sqrtm of the block-diagonal PA is
// blockdiag(sqrtm(P), sqrtm(noise covariance)): the state block is the symmetric square root of the
// carried covariance, recomputed twice per step on the device (cyclic Jacobi eigen-decomposition,
// V diag(sqrt(max(lambda_i, 0))) V^T == Re sqrtm for a symmetric matrix); the noise block is constant
// and comes from the host.
//
// Mapping (gfx950).  One lane per (trajectory, component) chain: the whole n x n covariance, its
// square root and the eigenvector matrix live in that lane's VGPRs; the sigma points are generated,
// pushed through f / h (values only, csrc/ssm_device.hpp) and folded into the moment sums one at a
// time, in two passes (mean, then covariances), so no point set is ever stored.  The components of a
// trajectory occupy KP consecutive lanes (K rounded up to a power of two, <= 256): the reweight is the
// same segmented xor-butterfly (adjacent-pair tree) as in gsf_scan.hpp.  Outputs go out as strided
// dword stores: a step is ~10^4 VALU operations, the stores are not what bounds it.
#pragma once
#include <type_traits>
#ifndef BF_JIT
#include <cstring>
#include <cmath>
#include <vector>
#include "bf_common.hpp"
#endif
#include "kf_math.hpp"
#include "scan_common.hpp"
#include "models.hpp"
#include "ssm_device.hpp"

namespace bf {

template <int N, int DQ, int M, int DR>
struct UkfModel {
  int dyn_id, emi_id, g_identity, d_identity;
  float dth[8], eth[8];
  float A[N * N];      // linear dynamics
  float Gm[N * DQ];    // F_q
  float Hm[M * N];     // linear emission
  float Dm[M * DR];    // H_r (constant-H_r emissions)
  float q0[DQ], r0[DR];
  float sQ[DQ * DQ];   // sqrtm(Q), symmetric
  float sR[DR * DR];   // sqrtm(R), symmetric
  // unscented-transform constants for the update (L = n + dr) and the prediction (L = n + dq)
  float c_u, ws_u, w0_u, wc_u;  // sqrt(L + lambda), 1 / (2 (L + lambda)), lambda / (L + lambda), w0 + 1 - alpha^2 + beta
  float c_p, ws_p, w0_p, wc_p;
  float uth_dyn[64], uth_emi[64];   // parameters of functions compiled from the caller's source (user_model.hip)
};

// Symmetric square root of a symmetric positive semi-definite matrix, in place (row-major N x N).
// Cyclic Jacobi: rotations annihilate a[p][q] in a fixed order, sweep after sweep, until the
// off-diagonal mass is below fp32 resolution; then V diag(sqrt(max(d, 0))) V^T.
template <int N>
__device__ __forceinline__ void sym_sqrt(float* a) {
#pragma clang fp contract(fast)   // (stated, not inherited: a build compiled at run time sets contraction off for the caller's functions)
  if constexpr (N == 1) {
    a[0] = sqrtf(fmaxf(a[0], 0.f));
    return;
  } else {
    float v[N * N];
    BF_UNROLL for (int i = 0; i < N * N; ++i) v[i] = (i / N == i % N) ? 1.f : 0.f;
    for (int sweep = 0; sweep < 12; ++sweep) {
      float off = 0.f, diag = 0.f;
      BF_UNROLL for (int p = 0; p < N; ++p) {
        diag = fmaf(a[p * N + p], a[p * N + p], diag);
        BF_UNROLL for (int q = p + 1; q < N; ++q) off = fmaf(a[p * N + q], a[p * N + q], off);
      }
      if (!(off > 1e-14f * diag)) break;  // also leaves on NaN
      BF_UNROLL for (int p = 0; p < N - 1; ++p) BF_UNROLL for (int q = p + 1; q < N; ++q) {
        const float apq = a[p * N + q];
        const float app = a[p * N + p], aqq = a[q * N + q];
        // rotation angle: tan(2 phi) = 2 a_pq / (a_qq - a_pp), the smaller root t = tan(phi)
        // (single-instruction reciprocal / square roots, 1 ulp: the rotation only has to shrink a_pq -- what an ulp in t, c
        // leaves behind goes in the next sweep, and the sweeps stop on the measured off-diagonal mass; IEEE division and
        // square root cost ten instructions each, three times per rotation)
#if defined(__HIP_DEVICE_COMPILE__)
        const float theta = (aqq - app) * __builtin_amdgcn_rcpf(2.f * apq);
        float t = __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.f)));
#else
        const float theta = (aqq - app) / (2.f * apq);
        float t = 1.f / (fabsf(theta) + sqrtf(fmaf(theta, theta, 1.f)));
#endif
        t = theta < 0.f ? -t : t;
        const bool skip = !(fabsf(apq) > 1e-30f);  // already zero (or NaN): identity rotation
        t = skip ? 0.f : t;
#if defined(__HIP_DEVICE_COMPILE__)
        const float c = __builtin_amdgcn_rsqf(fmaf(t, t, 1.f)), s = t * c;
#else
        const float c = 1.f / sqrtf(fmaf(t, t, 1.f)), s = t * c;
#endif
        a[p * N + p] = app - t * apq;
        a[q * N + q] = aqq + t * apq;
        a[p * N + q] = 0.f;
        a[q * N + p] = 0.f;
        BF_UNROLL for (int k = 0; k < N; ++k) {
          if (k != p && k != q) {
            const float akp = a[k * N + p], akq = a[k * N + q];
            const float np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
            a[k * N + p] = np_; a[p * N + k] = np_;
            a[k * N + q] = nq_; a[q * N + k] = nq_;
          }
          const float vkp = v[k * N + p], vkq = v[k * N + q];
          v[k * N + p] = c * vkp - s * vkq;
          v[k * N + q] = s * vkp + c * vkq;
        }
      }
    }
    float d[N];
    BF_UNROLL for (int i = 0; i < N; ++i) d[i] = sqrtf(fmaxf(a[i * N + i], 0.f));
    BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int j = i; j < N; ++j) {
      float s = 0.f;
      BF_UNROLL for (int k = 0; k < N; ++k) s = fmaf(v[i * N + k] * d[k], v[j * N + k], s);
      a[i * N + j] = s;
      a[j * N + i] = s;
    }
  }
}

// f(x, q, u) of the registry dynamics (noise through the constant F_q)
template <class SP = SpecRuntime, int N, int DQ, int M, int DR>
__device__ __forceinline__ void ukf_dyn(const UkfModel<N, DQ, M, DR>& p, const float* x, const float* q, float u0, float* out) {
  dyn_value_t<N, DQ, UkfModel<N, DQ, M, DR>, SP>(p, x, q, u0, out);   // (SP::user_dyn: the caller's f(x, q, u) from source)
}

// h(x, r, u) of the registry emissions: g(x, u) + H_r r, or the stochastic-volatility form
// u beta exp(x / sigma) r + (1 - u)(c x + r)  (docs/experiments/adaptive_experiment.py:51-54)
template <class SP = SpecRuntime, int N, int DQ, int M, int DR>
__device__ __forceinline__ void ukf_emi(const UkfModel<N, DQ, M, DR>& p, const float* x, const float* r, float u0, float* out) {
#pragma clang fp contract(fast)
#ifdef BF_USER_EMI
  if constexpr (SP::user_emi) {   // the caller's h(x, r, u) from source
    bfu::emission<float>(x, r, u0, p.uth_emi, out);
    return;
  }
#endif
  if (p.emi_id == EMI_STOCH_VOL) {
    if constexpr (M == N && DR == N) {
      const float sigma = p.eth[0], beta = p.eth[1], c = p.eth[2];
      BF_UNROLL for (int i = 0; i < N; ++i) out[i] = u0 * beta * expf(x[i] / sigma) * r[i] + (1.f - u0) * (c * x[i] + r[i]);
    }
    return;
  }
  emi_mean_t<N, M>(p, x, u0, out);
  if (p.d_identity) {
    if constexpr (DR == M) BF_UNROLL for (int a = 0; a < M; ++a) out[a] += r[a];
  } else {
    BF_UNROLL for (int a = 0; a < M; ++a) {
      float s = 0.f;
      BF_UNROLL for (int k = 0; k < DR; ++k) s = fmaf(p.Dm[a * DR + k], r[k], s);
      out[a] += s;
    }
  }
}

// _ukf_condition_on_nonadditive (inference.py:198-224): m, P <- posterior; returns the log-likelihood.
// sR: sqrtm of this step's emission noise covariance (mdl.sR, or row t of the per-step table when R is (T, dr, dr):
// inference.py:416 picks R_t before the step, :206-209 take the square root of blockdiag(P, R_t))
template <class SP = SpecRuntime, int N, int DQ, int M, int DR>
__device__ __forceinline__ float ukf_condition_on(const UkfModel<N, DQ, M, DR>& mdl, float* m, float* P, const float* yv, float u0,
                                                  const float* sR) {
#pragma clang fp contract(fast)
  constexpr int EP = N * N;
  float ll;

  float sP[EP];
  BF_UNROLL for (int i = 0; i < EP; ++i) sP[i] = P[i];
  sym_sqrt<N>(sP);
  float h0[M], mu[M];
  ukf_emi<SP>(mdl, m, mdl.r0, u0, h0);
  // visits the 2 L sigma points in the order of utils.py:251-253: the plus rows, then the minus rows.  The points are
  // needed twice (mean, then covariances about the mean): small problems keep the 2 L images h(x) in registers, large
  // ones push the points through h again (2 L M floats would not fit)
  constexpr int NPTS = 2 * (N + DR);
  constexpr bool STORE = NPTS * M <= 96;
  float img[STORE ? NPTS * M : 1];
  auto for_points = [&](auto Eval, auto&& fn) __attribute__((always_inline)) {
    constexpr bool eval = decltype(Eval)::value || !STORE;
    int pt = 0;
    BF_UNROLL for (int sg = 0; sg < 2; ++sg) {
      const float cs = sg == 0 ? mdl.c_u : -mdl.c_u;
      BF_UNROLL for (int j = 0; j < N; ++j) {
        float x[N], dx[N], yy[M];
        BF_UNROLL for (int i = 0; i < N; ++i) {
          dx[i] = cs * sP[j * N + i];
          x[i] = m[i] + dx[i];
        }
        if constexpr (eval) {
          ukf_emi<SP>(mdl, x, mdl.r0, u0, yy);
          if constexpr (STORE) BF_UNROLL for (int a = 0; a < M; ++a) img[pt * M + a] = yy[a];
        } else {
          BF_UNROLL for (int a = 0; a < M; ++a) yy[a] = img[pt * M + a];
        }
        fn(yy, dx, true);
        ++pt;
      }
      BF_UNROLL for (int j = 0; j < DR; ++j) {
        float r[DR], dx[N], yy[M];
        BF_UNROLL for (int i = 0; i < DR; ++i) r[i] = mdl.r0[i] + cs * sR[j * DR + i];
        BF_UNROLL for (int i = 0; i < N; ++i) dx[i] = 0.f;
        if constexpr (eval) {
          ukf_emi<SP>(mdl, m, r, u0, yy);
          if constexpr (STORE) BF_UNROLL for (int a = 0; a < M; ++a) img[pt * M + a] = yy[a];
        } else {
          BF_UNROLL for (int a = 0; a < M; ++a) yy[a] = img[pt * M + a];
        }
        fn(yy, dx, false);
        ++pt;
      }
    }
  };
  BF_UNROLL for (int a = 0; a < M; ++a) mu[a] = 0.f;
  for_points(std::true_type{}, [&](const float* yy, const float*, bool) { BF_UNROLL for (int a = 0; a < M; ++a) mu[a] += yy[a]; });
  BF_UNROLL for (int a = 0; a < M; ++a) mu[a] = mu[a] * mdl.ws_u + h0[a] * mdl.w0_u;
  float S[M * M], C[M * N];
  BF_UNROLL for (int i = 0; i < M * M; ++i) S[i] = 0.f;
  BF_UNROLL for (int i = 0; i < M * N; ++i) C[i] = 0.f;
  for_points(std::false_type{}, [&](const float* yy, const float* dx, bool moved) {
    float dy[M];
    BF_UNROLL for (int a = 0; a < M; ++a) dy[a] = yy[a] - mu[a];
    BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int c2 = 0; c2 < M; ++c2) S[a * M + c2] = fmaf(dy[a], dy[c2], S[a * M + c2]);
    if (moved) BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int i = 0; i < N; ++i) C[a * N + i] = fmaf(dy[a], dx[i], C[a * N + i]);
  });
  float d0[M];
  BF_UNROLL for (int a = 0; a < M; ++a) d0[a] = h0[a] - mu[a];
  BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int c2 = 0; c2 < M; ++c2)
      S[a * M + c2] = S[a * M + c2] * mdl.ws_u + mdl.wc_u * (d0[a] * d0[c2]);
  BF_UNROLL for (int i = 0; i < M * N; ++i) C[i] *= mdl.ws_u;
  // K = psd_solve(S, C)^T;  P+ = P - K S K^T;  m+ = m + K (y - mu);  ll = MVN(mu, S).log_prob(y)
  psd_solve<M, N>(S, C);  // C <- (S + 1e-6)^-1 C = K^T
  float KS[N * M];
  BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int c2 = 0; c2 < M; ++c2) {
    float s = C[i] * S[c2];
    BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(C[a * N + i], S[a * M + c2], s);
    KS[i * M + c2] = s;
  }
  BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int j = 0; j < N; ++j) {
    float s = KS[i * M] * C[j];
    BF_UNROLL for (int c2 = 1; c2 < M; ++c2) s = fmaf(KS[i * M + c2], C[c2 * N + j], s);
    P[i * N + j] -= s;
  }
  float v[M];
  BF_UNROLL for (int a = 0; a < M; ++a) v[a] = yv[a] - mu[a];
  BF_UNROLL for (int i = 0; i < N; ++i) {
    float s = C[i] * v[0];
    BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(C[a * N + i], v[a], s);
    m[i] += s;
  }
  ll = mvn_logpdf_chol<M>(S, v);
  return ll;
}

// _ukf_predict_nonadditive (inference.py:146-174): m, P <- predicted mean and covariance.  sQ: sqrtm(Q_t), as sR above
template <class SP = SpecRuntime, int N, int DQ, int M, int DR>
__device__ __forceinline__ void ukf_predict(const UkfModel<N, DQ, M, DR>& mdl, float* m, float* P, float u0, const float* sQ) {
#pragma clang fp contract(fast)
  constexpr int EP = N * N;

  float sP[EP];
  BF_UNROLL for (int i = 0; i < EP; ++i) sP[i] = P[i];
  sym_sqrt<N>(sP);
  float f0[N], mu[N];
  ukf_dyn<SP>(mdl, m, mdl.q0, u0, f0);
  constexpr int NPTS = 2 * (N + DQ);
  constexpr bool STORE = NPTS * N <= 96;  // keep the 2 L images f(x) in registers instead of evaluating f twice
  float img[STORE ? NPTS * N : 1];
  auto for_points = [&](auto Eval, auto&& fn) __attribute__((always_inline)) {
    constexpr bool eval = decltype(Eval)::value || !STORE;
    int pt = 0;
    BF_UNROLL for (int sg = 0; sg < 2; ++sg) {
      const float cs = sg == 0 ? mdl.c_p : -mdl.c_p;
      BF_UNROLL for (int j = 0; j < N; ++j) {
        float x[N], xx[N];
        if constexpr (eval) {
          BF_UNROLL for (int i = 0; i < N; ++i) x[i] = m[i] + cs * sP[j * N + i];
          ukf_dyn<SP>(mdl, x, mdl.q0, u0, xx);
          if constexpr (STORE) BF_UNROLL for (int i = 0; i < N; ++i) img[pt * N + i] = xx[i];
        } else {
          BF_UNROLL for (int i = 0; i < N; ++i) xx[i] = img[pt * N + i];
        }
        fn(xx);
        ++pt;
      }
      BF_UNROLL for (int j = 0; j < DQ; ++j) {
        float q[DQ], xx[N];
        if constexpr (eval) {
          BF_UNROLL for (int i = 0; i < DQ; ++i) q[i] = mdl.q0[i] + cs * sQ[j * DQ + i];
          ukf_dyn<SP>(mdl, m, q, u0, xx);
          if constexpr (STORE) BF_UNROLL for (int i = 0; i < N; ++i) img[pt * N + i] = xx[i];
        } else {
          BF_UNROLL for (int i = 0; i < N; ++i) xx[i] = img[pt * N + i];
        }
        fn(xx);
        ++pt;
      }
    }
  };
  BF_UNROLL for (int i = 0; i < N; ++i) mu[i] = 0.f;
  for_points(std::true_type{}, [&](const float* xx) { BF_UNROLL for (int i = 0; i < N; ++i) mu[i] += xx[i]; });
  BF_UNROLL for (int i = 0; i < N; ++i) mu[i] = mu[i] * mdl.ws_p + f0[i] * mdl.w0_p;
  BF_UNROLL for (int i = 0; i < EP; ++i) P[i] = 0.f;
  for_points(std::false_type{}, [&](const float* xx) {
    float d[N];
    BF_UNROLL for (int i = 0; i < N; ++i) d[i] = xx[i] - mu[i];
    BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int j = 0; j < N; ++j) P[i * N + j] = fmaf(d[i], d[j], P[i * N + j]);
  });
  float d0[N];
  BF_UNROLL for (int i = 0; i < N; ++i) d0[i] = f0[i] - mu[i];
  BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int j = 0; j < N; ++j)
      P[i * N + j] = P[i * N + j] * mdl.ws_p + mdl.wc_p * (d0[i] * d0[j]);
  BF_UNROLL for (int i = 0; i < N; ++i) m[i] = mu[i];
}

#ifdef BF_USER_EKF_NODES
// Extended-Kalman node operations around functions compiled from the caller's source (user_model.hip): _predict /
// _condition_on (inference.py:51-105) with the Jacobians of :58-61 / :82-86 -- jacfwd w.r.t. the state AND w.r.t. the noise, at
// the noise bias -- by forward-mode dual numbers, one seed direction after the other on the chain's own lane; F_q Q F_q^T and
// H_r R H_r^T are formed here every step.  A function that is not from source must be the registry's linear one (A x + G q,
// H x + D r: its Jacobians are the matrices).  The model block is a UkfModel whose sQ / sR (and per-step tables) hold Q / R
// THEMSELVES (fill_ukf_model_view, raw-covariance flag).  Used by the augmented filter's tree nodes (agsf_scan.hpp) and, one
// lane per (trajectory, component), by the Gaussian-sum scan below.
template <int N, int DQ, int M, int DR, class SP>
struct UserEkfNodes {
  using Arg = const UkfModel<N, DQ, M, DR>*;
  static constexpr int TVQ = DQ * DQ, TVR = DR * DR;
  template <int R_, int C_, int D_>   // J (R_ x C_) S (C_ x C_) J^T: (J S) first, then times J^T -- the reference's association
  static __device__ __forceinline__ void congruence(const float* J, const float* S, float* out) {
    float JS[R_ * C_];
    BF_UNROLL for (int i = 0; i < R_; ++i) BF_UNROLL for (int j = 0; j < C_; ++j) {
      float s = 0.f;
      BF_UNROLL for (int k = 0; k < C_; ++k) s = fmaf(J[i * C_ + k], S[k * C_ + j], s);
      JS[i * C_ + j] = s;
    }
    BF_UNROLL for (int i = 0; i < R_; ++i) BF_UNROLL for (int j = 0; j < R_; ++j) {
      float s = 0.f;
      BF_UNROLL for (int k = 0; k < C_; ++k) s = fmaf(JS[i * C_ + k], J[j * C_ + k], s);
      out[i * D_ + j] = s;
    }
  }
  static __device__ __forceinline__ void predict(Arg mdl, float* m, float* P, float u0, const float* tq) {
    float F[N * N], Fq[N * DQ], fx[N], FqQFq[N * N];
#ifdef BF_USER_DYN
    if constexpr (SP::user_dyn) {
      bfu::Dual xd[N], qd[DQ], od[N];
      BF_UNROLL for (int i = 0; i < N; ++i) xd[i] = bfu::Dual(m[i]);
      BF_UNROLL for (int i = 0; i < DQ; ++i) qd[i] = bfu::Dual(mdl->q0[i]);
      BF_UNROLL for (int s = 0; s < N + DQ; ++s) {
        if (s < N) xd[s < N ? s : 0].d = 1.f; else qd[s >= N ? s - N : 0].d = 1.f;
        bfu::dynamics<bfu::Dual>(xd, qd, bfu::Dual(u0), mdl->uth_dyn, od);
        if (s < N) xd[s < N ? s : 0].d = 0.f; else qd[s >= N ? s - N : 0].d = 0.f;
        BF_UNROLL for (int i = 0; i < N; ++i) {
          if (s < N) F[i * N + (s < N ? s : 0)] = od[i].d; else Fq[i * DQ + (s >= N ? s - N : 0)] = od[i].d;
          fx[i] = od[i].v;
        }
      }
    } else
#endif
    {  // the registry's linear dynamics A x + G q
      BF_UNROLL for (int i = 0; i < N * N; ++i) F[i] = mdl->A[i];
      BF_UNROLL for (int i = 0; i < N * DQ; ++i) Fq[i] = mdl->Gm[i];
      BF_UNROLL for (int i = 0; i < N; ++i) {
        float s = 0.f, g = 0.f;
        BF_UNROLL for (int k = 0; k < N; ++k) s = fmaf(mdl->A[i * N + k], m[k], s);
        BF_UNROLL for (int k = 0; k < DQ; ++k) g = fmaf(mdl->Gm[i * DQ + k], mdl->q0[k], g);
        fx[i] = s + g;
      }
    }
    congruence<N, DQ, N>(Fq, tq ? tq : mdl->sQ, FqQFq);
    predict_cov<N>(F, FqQFq, P);  // F P F^T + F_q Q F_q^T
    BF_UNROLL for (int i = 0; i < N; ++i) m[i] = fx[i];
  }
  static __device__ __forceinline__ float condition(Arg mdl, float* m, float* P, const float* yv, float u0, const float* tr) {
    float H[M * N], Hr[M * DR], hx[M], HrRHr[M * M], v[M];
#ifdef BF_USER_EMI
    if constexpr (SP::user_emi) {
      bfu::Dual xd[N], rd[DR], od[M];
      BF_UNROLL for (int i = 0; i < N; ++i) xd[i] = bfu::Dual(m[i]);
      BF_UNROLL for (int i = 0; i < DR; ++i) rd[i] = bfu::Dual(mdl->r0[i]);
      BF_UNROLL for (int s = 0; s < N + DR; ++s) {
        if (s < N) xd[s < N ? s : 0].d = 1.f; else rd[s >= N ? s - N : 0].d = 1.f;
        bfu::emission<bfu::Dual>(xd, rd, bfu::Dual(u0), mdl->uth_emi, od);
        if (s < N) xd[s < N ? s : 0].d = 0.f; else rd[s >= N ? s - N : 0].d = 0.f;
        BF_UNROLL for (int a = 0; a < M; ++a) {
          if (s < N) H[a * N + (s < N ? s : 0)] = od[a].d; else Hr[a * DR + (s >= N ? s - N : 0)] = od[a].d;
          hx[a] = od[a].v;
        }
      }
    } else
#endif
    {  // the registry's linear emission H x + D r
      BF_UNROLL for (int i = 0; i < M * N; ++i) H[i] = mdl->Hm[i];
      BF_UNROLL for (int i = 0; i < M * DR; ++i) Hr[i] = mdl->Dm[i];
      BF_UNROLL for (int a = 0; a < M; ++a) {
        float s = 0.f, g = 0.f;
        BF_UNROLL for (int k = 0; k < N; ++k) s = fmaf(mdl->Hm[a * N + k], m[k], s);
        BF_UNROLL for (int k = 0; k < DR; ++k) g = fmaf(mdl->Dm[a * DR + k], mdl->r0[k], g);
        hx[a] = s + g;
      }
    }
    congruence<M, DR, M>(Hr, tr ? tr : mdl->sR, HrRHr);
    BF_UNROLL for (int a = 0; a < M; ++a) v[a] = yv[a] - hx[a];
    return condition_on<N, M>(H, HrRHr, v, m, P);
  }
};
#endif  // BF_USER_EKF_NODES

// NODES = void: the unscented operations above (sQ / sR: square roots).  Any other NODES (UserEkfNodes): its condition / predict
// pair on the same lane-per-(trajectory, component) scan -- the Gaussian-sum filter of inference.py:333-371 with the caller's
// functions in registers (sQ / sR: the covariances themselves).
template <int N, int DQ, int M, int DR, class SP = SpecRuntime, class NODES = void>
__device__ __forceinline__ void ugsf_scan_body(const UkfModel<N, DQ, M, DR>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB,
                 long long u_sT, CarryView carry, OutViews out, long long B, long long T, int K, int KP,
                 const float* __restrict__ tvsq, const float* __restrict__ tvsr) {
#pragma clang fp contract(fast)
  const UkfModel<N, DQ, M, DR>& mdl = *mdlp;
  constexpr int EP = N * N;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpb = 256 / KP;  // trajectories per workgroup
  const int k = tid % KP;
  const long long b_raw = (long long)blockIdx.x * tpb + tid / KP;
  const bool traj_ok = b_raw < B;
  const bool comp_ok = k < K;
  const bool chain_ok = traj_ok && comp_ok;
  const long long b = traj_ok ? b_raw : B - 1;
  const long long chain = b * K + (comp_ok ? k : 0);  // padding components shadow component 0 (never stored)

  __shared__ float red[8];
  auto reduce_k = [&](float v, auto op) {  // over the KP lanes of a trajectory, adjacent-pair tree
    const int lim = KP < 64 ? KP : 64;
    for (int off = 1; off < lim; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    if (KP > 64) {
      lds_barrier();
      if (lane == 0) red[wave] = v;
      lds_barrier();
      const int wpt = KP / 64;
      const int w0 = (wave / wpt) * wpt;
      if (wpt == 2) v = op(red[w0], red[w0 + 1]);
      else v = op(op(red[w0], red[w0 + 1]), op(red[w0 + 2], red[w0 + 3]));
    }
    return v;
  };

  float P[EP], m[N], w;
  BF_UNROLL for (int i = 0; i < EP; ++i) P[i] = carry.P_in[chain * EP + i];
  BF_UNROLL for (int i = 0; i < N; ++i) m[i] = carry.m_in[chain * N + i];
  w = comp_ok ? (carry.w_in ? carry.w_in[chain] : 1.0f / (float)K) : 0.f;

  // observation and input of step t + 1 are fetched while step t runs (a load issued at the top of its own step would put
  // an HBM round trip on every step's critical path)
  float ynext[M], unext;
  BF_UNROLL for (int a = 0; a < M; ++a) ynext[a] = y.p[b * y.sB + a * y.sE];
  unext = uptr ? uptr[b * u_sB + 0 * u_sT] : 0.f;
  for (long long t = 0; t < T; ++t) {
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = ynext[a];
    const float u0 = unext;
    {
      const long long tn = t + 1 < T ? t + 1 : t;
      BF_UNROLL for (int a = 0; a < M; ++a) ynext[a] = y.p[b * y.sB + tn * y.sT + a * y.sE];
      unext = uptr ? uptr[b * u_sB + tn * u_sT] : 0.f;
    }
    float ll;

    // ================= _ukf_condition_on_nonadditive (inference.py:198-224)
    if constexpr (std::is_void<NODES>::value) ll = ukf_condition_on<SP>(mdl, m, P, yv, u0, tvsr ? tvsr + t * (DR * DR) : mdl.sR);
    else ll = NODES::condition(mdlp, m, P, yv, u0, tvsr ? tvsr + t * (DR * DR) : nullptr);

    // ================= reweight (inference.py:424-427)
    {
      const float llm = comp_ok ? ll : -__builtin_inff();
      const float mx = reduce_k(llm, [](float a, float b2) { return (a != a || b2 != b2) ? __builtin_nanf("") : fmaxf(a, b2); });
      const float e = comp_ok ? expf(ll - mx) * w : 0.f;
      const float tot = reduce_k(e, [](float a, float b2) { return a + b2; });
      w = comp_ok ? e / tot : 0.f;
    }
    if (chain_ok) {
      if (out.m.p) BF_UNROLL for (int i = 0; i < N; ++i) out.m.p[b * out.m.sB + k * out.m.sK + t * out.m.sT + i * out.m.sE] = m[i];
      if (out.P.p) BF_UNROLL for (int i = 0; i < EP; ++i) out.P.p[b * out.P.sB + k * out.P.sK + t * out.P.sT + i * out.P.sE] = P[i];
      if (out.w.p) out.w.p[b * out.w.sB + k * out.w.sK + t * out.w.sT] = w;
      if (out.ll.p) out.ll.p[b * out.ll.sB + k * out.ll.sK + t * out.ll.sT] = ll;
    }

    // ================= _ukf_predict_nonadditive (inference.py:146-174)
    if constexpr (std::is_void<NODES>::value) ukf_predict<SP>(mdl, m, P, u0, tvsq ? tvsq + t * (DQ * DQ) : mdl.sQ);
    else NODES::predict(mdlp, m, P, u0, tvsq ? tvsq + t * (DQ * DQ) : nullptr);
    if (chain_ok) {
      if (out.pm.p) BF_UNROLL for (int i = 0; i < N; ++i) out.pm.p[b * out.pm.sB + k * out.pm.sK + t * out.pm.sT + i * out.pm.sE] = m[i];
      if (out.pP.p) BF_UNROLL for (int i = 0; i < EP; ++i) out.pP.p[b * out.pP.sB + k * out.pP.sK + t * out.pP.sT + i * out.pP.sE] = P[i];
    }
  }

  if (chain_ok) {
    if (carry.m_out) BF_UNROLL for (int i = 0; i < N; ++i) carry.m_out[chain * N + i] = m[i];
    if (carry.P_out) BF_UNROLL for (int i = 0; i < EP; ++i) carry.P_out[chain * EP + i] = P[i];
    if (carry.w_out) carry.w_out[chain] = w;
  }
}

template <int N, int DQ, int M, int DR>
__global__ void __launch_bounds__(256)
ugsf_scan_kernel(const UkfModel<N, DQ, M, DR>* __restrict__ mdlp, CView y, const float* __restrict__ uptr, long long u_sB,
                 long long u_sT, CarryView carry, OutViews out, long long B, long long T, int K, int KP,
                 const float* __restrict__ tvsq, const float* __restrict__ tvsr) {
  ugsf_scan_body<N, DQ, M, DR, SpecRuntime>(mdlp, y, uptr, u_sB, u_sT, carry, out, B, T, K, KP, tvsq, tvsr);
}

#ifndef BF_JIT   // host side
// ---------------------------------------------------------------------------------------
// host: symmetric square root in double precision (cyclic Jacobi), rounded to fp32
static inline void host_sym_sqrt(const float* A, int n, float* out) {
  std::vector<double> a(n * n), v(n * n, 0.0);
  for (int i = 0; i < n * n; ++i) a[i] = A[i];
  for (int i = 0; i < n; ++i) v[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += a[p * n + p] * a[p * n + p];
      for (int q = p + 1; q < n; ++q) off += a[p * n + q] * a[p * n + q];
    }
    if (!(off > 1e-30 * diag)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[p * n + q];
        if (!(std::fabs(apq) > 1e-300)) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = 1.0 / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k2 = 0; k2 < n; ++k2) {
          const double akp = a[k2 * n + p], akq = a[k2 * n + q];
          a[k2 * n + p] = c * akp - s * akq;
          a[k2 * n + q] = s * akp + c * akq;
        }
        for (int k2 = 0; k2 < n; ++k2) {
          const double apk = a[p * n + k2], aqk = a[q * n + k2];
          a[p * n + k2] = c * apk - s * aqk;
          a[q * n + k2] = s * apk + c * aqk;
        }
        for (int k2 = 0; k2 < n; ++k2) {
          const double vkp = v[k2 * n + p], vkq = v[k2 * n + q];
          v[k2 * n + p] = c * vkp - s * vkq;
          v[k2 * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k2 = 0; k2 < n; ++k2) s += v[i * n + k2] * std::sqrt(a[k2 * n + k2] > 0.0 ? a[k2 * n + k2] : 0.0) * v[j * n + k2];
      out[i * n + j] = (float)s;
    }
}

// per-step table -> device (stream-ordered upload through the constant cache, const_cache.hip); empty = NULL
static inline int upload_table(const std::vector<float>& v, hipStream_t stream, const float** d_out) {
  *d_out = nullptr;
  if (v.empty()) return BF_OK;
  const void* dv = nullptr;
  const int rc = device_constants(v.data(), sizeof(float) * v.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  *d_out = static_cast<const float*>(dv);
  return BF_OK;
}

// tvsq / tvsr: filled with sqrtm(Q_t) / sqrtm(R_t), one matrix per step, when p->Q_steps / p->R_steps > 1
// (inference.py:414-417: the step's covariances are picked by _get_params before the unscented transforms)
// The fields of a UkfModel<N, DQ, M, DR> by address for dimensions known at run time (4-byte members, declaration order, no
// padding: see BpfModelView in ssm_device.hpp).  user_flags: bit 0 = dynamics, bit 1 = emission from the caller's source.
struct UkfModelView {
  int N, DQ, M, DR;
  int *dyn_id, *emi_id, *g_identity, *d_identity;
  float *dth, *eth, *A, *Gm, *Hm, *Dm, *q0, *r0, *sQ, *sR, *cu, *cp, *uth_dyn, *uth_emi;
};
constexpr size_t ukf_model_words(int n, int dq, int m, int dr) {
  return 4 + 16 + (size_t)n * n + (size_t)n * dq + (size_t)m * n + (size_t)m * dr + dq + dr + (size_t)dq * dq + (size_t)dr * dr + 8 + 128;
}
static inline UkfModelView ukf_model_view_flat(uint32_t* w, int n, int dq, int m, int dr) {
  UkfModelView v;
  v.N = n; v.DQ = dq; v.M = m; v.DR = dr;
  int* iw = reinterpret_cast<int*>(w);
  v.dyn_id = iw; v.emi_id = iw + 1; v.g_identity = iw + 2; v.d_identity = iw + 3;
  float* f = reinterpret_cast<float*>(w) + 4;
  auto take = [&](size_t k) { float* r = f; f += k; return r; };
  v.dth = take(8); v.eth = take(8); v.A = take((size_t)n * n); v.Gm = take((size_t)n * dq); v.Hm = take((size_t)m * n); v.Dm = take((size_t)m * dr);
  v.q0 = take(dq); v.r0 = take(dr); v.sQ = take((size_t)dq * dq); v.sR = take((size_t)dr * dr); v.cu = take(4); v.cp = take(4);
  v.uth_dyn = take(64); v.uth_emi = take(64);
  return v;
}

static inline int fill_ukf_model_view(const bf_model* p, const bf_ukf_params* up, UkfModelView e, int user_flags,
                                      std::vector<float>* tvsq = nullptr, std::vector<float>* tvsr = nullptr) {
  const int N = e.N, DQ = e.DQ, M = e.M, DR = e.DR;
  if ((p->Q_steps > 1 && !tvsq) || (p->R_steps > 1 && !tvsr))
    return set_error(BF_EUNSUPPORTED, "time-varying Q/R are not supported on this path");
  if (p->flags != 0) return set_error(BF_EUNSUPPORTED, "legacy-class flags do not apply to the unscented filter");
  *e.dyn_id = p->dyn_id;
  *e.emi_id = p->emi_id;
  *e.g_identity = 1;
  *e.d_identity = 1;
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
  if (user_flags & 2) {
    if (p->n_emi_theta > 64) return set_error(BF_EUNSUPPORTED, "an emission function from source takes at most 64 parameters here");
    for (int i = 0; i < p->n_emi_theta; ++i) e.uth_emi[i] = th[i];
  } else switch (p->emi_id) {
    case EMI_LINEAR:
      if (p->n_emi_theta != M * N + M * DR) return set_error(BF_EINVAL, "linear emission: theta must hold H and D");
      for (int i = 0; i < M * N; ++i) e.Hm[i] = th[i];
      for (int i = 0; i < M * DR; ++i) e.Dm[i] = th[M * N + i];
      *e.d_identity = 0;
      break;
    case EMI_BEARING_RANGE:
      if (N != 4 || M != 2 || DR != 2) return set_error(BF_EINVAL, "bearing_range: n = 4, m = dr = 2");
      break;
    case EMI_BEARING:
      if (N != 4 || M != 1 || DR != 1) return set_error(BF_EINVAL, "bearing: n = 4, m = dr = 1");
      break;
    case EMI_QUADRATIC:
      if (M != 1 || DR != 1 || p->n_emi_theta != 1) return set_error(BF_EINVAL, "quadratic: m = dr = 1");
      e.eth[0] = th[0];
      break;
    case EMI_STOCH_VOL:
      if (M != N || DR != N || p->n_emi_theta != 3) return set_error(BF_EINVAL, "stoch_vol: m = dr = n, theta = (sigma, beta, c)");
      for (int i = 0; i < 3; ++i) e.eth[i] = th[i];
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown emission function id %d", p->emi_id);
  }
  for (int i = 0; i < DQ; ++i) e.q0[i] = p->q0 ? p->q0[i] : 0.f;
  for (int i = 0; i < DR; ++i) e.r0[i] = p->r0 ? p->r0[i] : 0.f;
  // user_flags & 4: the covariances themselves instead of their square roots (extended-Kalman nodes of the augmented filter)
  const bool raw = (user_flags & 4) != 0;
  auto root = [&](const float* src, int d, float* dst) {
    if (raw) std::memcpy(dst, src, sizeof(float) * (size_t)d * d); else host_sym_sqrt(src, d, dst);
  };
  root(p->Q, DQ, e.sQ);
  root(p->R, DR, e.sR);
  if (p->Q_steps > 1) {
    tvsq->resize((size_t)p->Q_steps * DQ * DQ);
    for (int t = 0; t < p->Q_steps; ++t) root(p->Q + (size_t)t * DQ * DQ, DQ, tvsq->data() + (size_t)t * DQ * DQ);
  }
  if (p->R_steps > 1) {
    tvsr->resize((size_t)p->R_steps * DR * DR);
    for (int t = 0; t < p->R_steps; ++t) root(p->R + (size_t)t * DR * DR, DR, tvsr->data() + (size_t)t * DR * DR);
  }
  auto consts = [&](int L, float& c, float& ws, float& w0, float& wc) {
    const float a2 = up->alpha * up->alpha;
    const float lam = a2 * ((float)L + up->kappa) - (float)L;  // inference.py:163, :206
    c = sqrtf((float)L + lam);                                  // utils.py:251
    ws = 1.0f / (2.0f * (lam + (float)L));
    w0 = lam / (lam + (float)L);
    wc = w0 + 1.0f - a2 + up->beta;
  };
  consts(N + DR, e.cu[0], e.cu[1], e.cu[2], e.cu[3]);
  consts(N + DQ, e.cp[0], e.cp[1], e.cp[2], e.cp[3]);
  return BF_OK;
}

template <int N, int DQ, int M, int DR>
static inline int fill_ukf_model(const bf_model* p, const bf_ukf_params* up, UkfModel<N, DQ, M, DR>& e,
                                 std::vector<float>* tvsq = nullptr, std::vector<float>* tvsr = nullptr) {
  static_assert(sizeof(UkfModel<N, DQ, M, DR>) == 4 * ukf_model_words(N, DQ, M, DR), "UkfModel: 4-byte members in declaration order, no padding");
  std::memset(&e, 0, sizeof(e));
  return fill_ukf_model_view(p, up, ukf_model_view_flat(reinterpret_cast<uint32_t*>(&e), N, DQ, M, DR), 0, tvsq, tvsr);
}

template <int N, int DQ, int M, int DR>
static inline int launch_ugsf(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B,
                              long long T, int K, const bf_carry* carry, const bf_out_desc* out, hipStream_t stream) {
  UkfModel<N, DQ, M, DR> h;
  std::memset(&h, 0, sizeof(h));  // the constant cache compares contents
  std::vector<float> tvsq, tvsr;
  int rc = fill_ukf_model<N, DQ, M, DR>(p, up, h, &tvsq, &tvsr);
  if (rc != BF_OK) return rc;
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  int KP = 1;
  while (KP < K) KP <<= 1;
  if (KP > 256) return set_error(BF_EUNSUPPORTED, "unscented Gaussian-sum filter: %d components exceed one workgroup (256 lanes)", K);
  if (out->coll_mean.ptr || out->coll_cov.ptr)
    return set_error(BF_EUNSUPPORTED, "collapsed streams are produced by bf_gsf_ekf_f32 only");
  const void* dv = nullptr;
  rc = device_constants(&h, sizeof(h), stream, &dv);
  if (rc != BF_OK) return rc;
  const UkfModel<N, DQ, M, DR>* d_mdl = static_cast<const UkfModel<N, DQ, M, DR>*>(dv);
  const float *d_tvsq = nullptr, *d_tvsr = nullptr;
  if ((rc = upload_table(tvsq, stream, &d_tvsq)) != BF_OK || (rc = upload_table(tvsr, stream, &d_tvsr)) != BF_OK) return rc;
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  const int tpb = 256 / KP;
  hipLaunchKernelGGL((ugsf_scan_kernel<N, DQ, M, DR>), dim3((unsigned)((B + tpb - 1) / tpb)), dim3(256), 0, stream, d_mdl, yv,
                     (u && u->ptr) ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0, cv, ov, B, T, K, KP, d_tvsq, d_tvsr);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

#endif  // BF_JIT

}  // namespace bf
