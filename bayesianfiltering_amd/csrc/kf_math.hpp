// Register-resident small-matrix math for the per-lane filter recursions (gfx950).
// Every loop has compile-time bounds and is fully unrolled so that the arrays live in VGPRs
// (runtime-indexed arrays would go to scratch).  Row-major everywhere.
#pragma once
#include <hip/hip_runtime.h>

namespace bf {

#define BF_UNROLL _Pragma("unroll")

// Single-instruction transcendental forms (v_rcp_f32 / v_sqrt_f32 / v_log_f32 / v_exp_f32, each
// accurate to ~1 ulp).  The IEEE-exact library forms cost ~10 VALU instructions apiece and the
// recursions here are instruction-issue bound; the parity budget (1e-5 relative, fp32) is two
// orders of magnitude above what these cost.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// The reweight of gaussfiltax/inference.py:347-350 for ONE component: lls -= max(lls) gives 0
// (NaN if ll is not finite), w <- exp(0) * w, w <- w / sum(w) = w / w: exactly 1.0 unless
// something is non-finite or w == 0, in which case NaN -- evaluated without exp or division.
__device__ __forceinline__ float reweight_single(float ll, float w) {
  const float l0 = ll - ll;          // 0, or NaN for +-inf / NaN
  const float wn = (l0 + 1.0f) * w;  // exp(0) * w
  const bool ok = (wn == wn) && (wn != 0.0f) && (fabsf(wn) != __builtin_inff());
  return ok ? 1.0f : __builtin_nanf("");
}

// c[R x C] = a[R x K] * b[K x C]
template <int R, int K, int C>
__device__ __forceinline__ void mm(const float* a, const float* b, float* c) {
  BF_UNROLL for (int i = 0; i < R; ++i) BF_UNROLL for (int j = 0; j < C; ++j) {
    float s = a[i * K] * b[j];
    BF_UNROLL for (int k = 1; k < K; ++k) s = fmaf(a[i * K + k], b[k * C + j], s);
    c[i * C + j] = s;
  }
}

// c[R x C] = a[R x K] * b[C x K]^T
template <int R, int K, int C>
__device__ __forceinline__ void mm_nt(const float* a, const float* b, float* c) {
  BF_UNROLL for (int i = 0; i < R; ++i) BF_UNROLL for (int j = 0; j < C; ++j) {
    float s = a[i * K] * b[j * K];
    BF_UNROLL for (int k = 1; k < K; ++k) s = fmaf(a[i * K + k], b[j * K + k], s);
    c[i * C + j] = s;
  }
}

// c[R] = a[R x K] * x[K]
template <int R, int K>
__device__ __forceinline__ void mv(const float* a, const float* x, float* c) {
  BF_UNROLL for (int i = 0; i < R; ++i) {
    float s = a[i * K] * x[0];
    BF_UNROLL for (int k = 1; k < K; ++k) s = fmaf(a[i * K + k], x[k], s);
    c[i] = s;
  }
}

// Solve (S + 1e-6 on every entry) X = Bm for X, Bm is [M x C]; LU with partial pivoting in
// the order of LAPACK getrf/getrs (what jnp.linalg.solve runs): gaussfiltax/utils.py:256-259.
// Row swaps are done with selects so lanes with different pivots stay convergent.
template <int M, int C>
__device__ __forceinline__ void psd_solve(const float* S, float* X /* in: Bm, out: X */, float jitter = 1e-6f) {
  float a[M * M];
  float rdiag[M];
  BF_UNROLL for (int i = 0; i < M * M; ++i) a[i] = S[i] + jitter;
  BF_UNROLL for (int k = 0; k < M; ++k) {
    // pivot search: first row of maximal |a[i][k]|, i >= k (isamax semantics)
    int p = k;
    float best = fabsf(a[k * M + k]);
    BF_UNROLL for (int i = k + 1; i < M; ++i) {
      float v = fabsf(a[i * M + k]);
      bool gt = v > best;
      best = gt ? v : best;
      p = gt ? i : p;
    }
    BF_UNROLL for (int i = k + 1; i < M; ++i) {
      bool sw = (p == i);
      BF_UNROLL for (int j = 0; j < M; ++j) {
        float u = a[k * M + j], v = a[i * M + j];
        a[k * M + j] = sw ? v : u;
        a[i * M + j] = sw ? u : v;
      }
      BF_UNROLL for (int j = 0; j < C; ++j) {
        float u = X[k * C + j], v = X[i * C + j];
        X[k * C + j] = sw ? v : u;
        X[i * C + j] = sw ? u : v;
      }
    }
    const float rpiv = fast_rcp(a[k * M + k]);
    rdiag[k] = rpiv;
    BF_UNROLL for (int i = k + 1; i < M; ++i) {
      float l = a[i * M + k] * rpiv;
      BF_UNROLL for (int j = k + 1; j < M; ++j) a[i * M + j] = fmaf(-l, a[k * M + j], a[i * M + j]);
      BF_UNROLL for (int j = 0; j < C; ++j) X[i * C + j] = fmaf(-l, X[k * C + j], X[i * C + j]);
    }
  }
  BF_UNROLL for (int i = M - 1; i >= 0; --i) {
    const float inv = rdiag[i];
    BF_UNROLL for (int j = 0; j < C; ++j) {
      float s = X[i * C + j];
      BF_UNROLL for (int q = i + 1; q < M; ++q) s = fmaf(-a[i * M + q], X[q * C + j], s);
      X[i * C + j] = s * inv;
    }
  }
}

// log N(y; mu, S) via the Cholesky factor of S, as tfp's MultivariateNormalFullCovariance
// does for gaussfiltax/inference.py:24; v = y - mu.  Non-PD S gives NaN (sqrt of a negative).
template <int M>
__device__ __forceinline__ float mvn_logpdf_chol(const float* S, const float* v) {
  float L[M * M];
  float rd[M];
  float dprod = 1.0f;
  BF_UNROLL for (int j = 0; j < M; ++j) {
    float d = S[j * M + j];
    BF_UNROLL for (int k = 0; k < j; ++k) d = fmaf(-L[j * M + k], L[j * M + k], d);
    d = fast_sqrt(d);  // NaN for a non-PD S, as the reference's Cholesky
    L[j * M + j] = d;
    dprod *= d;
    const float inv = fast_rcp(d);
    rd[j] = inv;
    BF_UNROLL for (int i = j + 1; i < M; ++i) {
      float s = S[i * M + j];
      BF_UNROLL for (int k = 0; k < j; ++k) s = fmaf(-L[i * M + k], L[j * M + k], s);
      L[i * M + j] = s * inv;
    }
  }
  float quad = 0.f;
  float z[M];
  BF_UNROLL for (int i = 0; i < M; ++i) {
    float s = v[i];
    BF_UNROLL for (int j = 0; j < i; ++j) s = fmaf(-L[i * M + j], z[j], s);
    z[i] = s * rd[i];
    quad = fmaf(z[i], z[i], quad);
  }
  constexpr float kLog2Pi = 1.8378770664093453f;
  // sum_i log L_ii = log prod_i L_ii (one v_log_f32; M <= 8 keeps the product in range)
  return -0.5f * quad - 0.5f * float(M) * kLog2Pi - fast_log(dprod);
}

// _condition_on (gaussfiltax/inference.py:72-105) given the linearisation at the prior mean:
//   Hx [M x N], HrRHr = H_r R H_r^T [M x M], v = y - h(m).
// Updates m, P in place; returns the log-likelihood.
template <int N, int M>
__device__ __forceinline__ float condition_on(const float* Hx, const float* HrRHr, const float* v,
                                              float* m, float* P) {
  float HP[M * N];
  mm<M, N, N>(Hx, P, HP);               // H_x @ P
  float S[M * M];
  mm_nt<M, N, M>(HP, Hx, S);            // (H_x P) H_x^T
  BF_UNROLL for (int i = 0; i < M * M; ++i) S[i] = HrRHr[i] + S[i];
  float X[M * N];
  BF_UNROLL for (int i = 0; i < M * N; ++i) X[i] = HP[i];
  psd_solve<M, N>(S, X);                // X = (S + 1e-6)^-1 (H_x P);  K = X^T
  float KS[N * M];                      // K @ S,  K[i][a] = X[a][i]
  BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int b = 0; b < M; ++b) {
    float s = X[i] * S[b];
    BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * N + i], S[a * M + b], s);
    KS[i * M + b] = s;
  }
  BF_UNROLL for (int i = 0; i < N; ++i) BF_UNROLL for (int j = 0; j < N; ++j) {
    float s = KS[i * M] * X[j];
    BF_UNROLL for (int b = 1; b < M; ++b) s = fmaf(KS[i * M + b], X[b * N + j], s);
    P[i * N + j] -= s;                  // P - (K S) K^T
  }
  BF_UNROLL for (int i = 0; i < N; ++i) {
    float s = X[i] * v[0];
    BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * N + i], v[a], s);
    m[i] += s;                          // m + K (y - h(m))
  }
  return mvn_logpdf_chol<M>(S, v);
}

// Covariance part of _predict (inference.py:69): P <- F_x P F_x^T + FqQFq.
template <int N>
__device__ __forceinline__ void predict_cov(const float* Fx, const float* FqQFq, float* P) {
  float FP[N * N];
  mm<N, N, N>(Fx, P, FP);
  float t[N * N];
  mm_nt<N, N, N>(FP, Fx, t);
  BF_UNROLL for (int i = 0; i < N * N; ++i) P[i] = t[i] + FqQFq[i];
}

}  // namespace bf
