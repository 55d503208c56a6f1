/* Oracle (TEST INFRASTRUCTURE, not product): plain-C fp32 port of the NumPy oracle's Kalman
 * path (oracle/gaussfilt_oracle.py: gaussian_sum_filter with K = 1 and linear f, h), i.e. of
 *   _condition_on   gaussfiltax/inference.py:72-105
 *   reweight        gaussfiltax/inference.py:347-350
 *   _predict        gaussfiltax/inference.py:51-70
 *   psd_solve       gaussfiltax/utils.py:256-259   (S + 1e-6 everywhere, LU partial pivoting)
 *   _MVN_log_prob   gaussfiltax/inference.py:24    (Cholesky of the un-jittered S)
 * It exists to (a) check the HIP kernels at sizes the NumPy oracle is too slow for and (b) be
 * the timed CPU baseline of bench.py ("cpu_baseline", kind "port").  Runtime dimensions,
 * reference layout [B][1][T][E]; OpenMP over the batch axis.  PARITY UNPINNED (see
 * oracle/__init__.py); validated against the NumPy oracle in tests/test_oracle_c.py.
 *
 * Build:  gcc -O2 -fopenmp -shared -fPIC -o liboracle_kf.so kf_oracle.c -lm
 * (-O2 without -ffast-math so the summation order below is what runs; fp contraction off.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#pragma STDC FP_CONTRACT OFF

#define MAXN 128

static void mm(const float* a, const float* b, float* c, int R, int K, int C) { /* c = a b */
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += a[i * K + k] * b[k * C + j];
      c[i * C + j] = s;
    }
}
static void mm_nt(const float* a, const float* b, float* c, int R, int K, int C) { /* c = a b^T */
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += a[i * K + k] * b[j * K + k];
      c[i * C + j] = s;
    }
}

/* solve (S + 1e-6) X = Bm, Bm [M x C] overwritten */
static void psd_solve(const float* S, float* X, int M, int C, float* a /* M*M scratch */) {
  for (int i = 0; i < M * M; ++i) a[i] = S[i] + 1e-6f;
  for (int k = 0; k < M; ++k) {
    int p = k;
    float best = fabsf(a[k * M + k]);
    for (int i = k + 1; i < M; ++i)
      if (fabsf(a[i * M + k]) > best) { best = fabsf(a[i * M + k]); p = i; }
    if (p != k) {
      for (int j = 0; j < M; ++j) { float t = a[k * M + j]; a[k * M + j] = a[p * M + j]; a[p * M + j] = t; }
      for (int j = 0; j < C; ++j) { float t = X[k * C + j]; X[k * C + j] = X[p * C + j]; X[p * C + j] = t; }
    }
    for (int i = k + 1; i < M; ++i) {
      float l = a[i * M + k] / a[k * M + k];
      for (int j = k + 1; j < M; ++j) a[i * M + j] -= l * a[k * M + j];
      for (int j = 0; j < C; ++j) X[i * C + j] -= l * X[k * C + j];
    }
  }
  for (int i = M - 1; i >= 0; --i)
    for (int j = 0; j < C; ++j) {
      float s = X[i * C + j];
      for (int q = i + 1; q < M; ++q) s -= a[i * M + q] * X[q * C + j];
      X[i * C + j] = s / a[i * M + i];
    }
}

static float mvn_logpdf(const float* S, const float* v, int M, float* L, float* z) {
  for (int j = 0; j < M; ++j) {
    float d = S[j * M + j];
    for (int k = 0; k < j; ++k) d -= L[j * M + k] * L[j * M + k];
    d = sqrtf(d);
    L[j * M + j] = d;
    for (int i = j + 1; i < M; ++i) {
      float s = S[i * M + j];
      for (int k = 0; k < j; ++k) s -= L[i * M + k] * L[j * M + k];
      L[i * M + j] = s / d;
    }
  }
  float quad = 0.f, logdet = 0.f;
  for (int i = 0; i < M; ++i) {
    float s = v[i];
    for (int j = 0; j < i; ++j) s -= L[i * M + j] * z[j];
    z[i] = s / L[i * M + i];
    quad += z[i] * z[i];
    logdet += logf(L[i * M + i]);
  }
  return -0.5f * quad - 0.5f * (float)M * 1.8378770664093453f - logdet;
}

/* All arrays host, row-major fp32.  GQG = (G Q) G^T [n,n], DRD = (D R) D^T [m,m], Gq0 [n],
 * Dr0 [m] are precomputed by the caller (time-invariant case).  Outputs in the reference
 * layout [B][T][E]; any output pointer may be NULL.  Returns 0, or -1 on bad dims. */
int oracle_kalman_filter_f32(int n, int m, const float* A, const float* H, const float* GQG, const float* DRD,
                             const float* Gq0, const float* Dr0, const float* y /* [B][T][m] */, int64_t B,
                             int64_t T, const float* m_in /* [B][n] */, const float* P_in /* [B][n][n] */,
                             float* w_out, float* means, float* covs, float* pmeans, float* pcovs, float* ll_out,
                             int nthreads) {
  if (n <= 0 || m <= 0 || n > MAXN || m > MAXN) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    float mu[MAXN], mp[MAXN], v[MAXN], z[MAXN];
    float* P = (float*)malloc(sizeof(float) * (size_t)(4 * n * n + 2 * m * n + n * m + 3 * m * m));
    float* FP = P + n * n;
    float* T2 = FP + n * n;
    float* HP = T2 + n * n;      /* m*n */
    float* X = HP + m * n;       /* m*n */
    float* KS = X + m * n;       /* n*m */
    float* S = KS + n * m;       /* m*m */
    float* a = S + m * m;        /* m*m */
    float* L = a + m * m;        /* m*m */
    float* Pn = L + m * m;       /* n*n (unused tail kept for alignment of the carve) */
    (void)Pn;
    memcpy(mu, m_in + b * n, sizeof(float) * n);
    memcpy(P, P_in + b * n * n, sizeof(float) * n * n);
    float w = 1.0f;
    for (int64_t t = 0; t < T; ++t) {
      const float* yt = y + (b * T + t) * m;
      /* condition_on */
      for (int i = 0; i < m; ++i) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += H[i * n + k] * mu[k];
        v[i] = yt[i] - (s + Dr0[i]);
      }
      mm(H, P, HP, m, n, n);
      mm_nt(HP, H, S, m, n, m);
      for (int i = 0; i < m * m; ++i) S[i] = DRD[i] + S[i];
      memcpy(X, HP, sizeof(float) * m * n);
      psd_solve(S, X, m, n, a);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
          float s = 0.f;
          for (int q = 0; q < m; ++q) s += X[q * n + i] * S[q * m + j];
          KS[i * m + j] = s;
        }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          float s = 0.f;
          for (int q = 0; q < m; ++q) s += KS[i * m + q] * X[q * n + j];
          P[i * n + j] -= s;
        }
      for (int i = 0; i < n; ++i) {
        float s = 0.f;
        for (int q = 0; q < m; ++q) s += X[q * n + i] * v[q];
        mu[i] += s;
      }
      float ll = mvn_logpdf(S, v, m, L, z);
      /* reweight, K = 1 */
      float l0 = ll - ll;
      float wn = expf(l0) * w;
      w = wn / wn;
      int64_t o = b * T + t;
      if (means) memcpy(means + o * n, mu, sizeof(float) * n);
      if (covs) memcpy(covs + o * n * n, P, sizeof(float) * n * n);
      if (w_out) w_out[o] = w;
      if (ll_out) ll_out[o] = ll;
      /* predict */
      for (int i = 0; i < n; ++i) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += A[i * n + k] * mu[k];
        mp[i] = s + Gq0[i];
      }
      memcpy(mu, mp, sizeof(float) * n);
      mm(A, P, FP, n, n, n);
      mm_nt(FP, A, T2, n, n, n);
      for (int i = 0; i < n * n; ++i) P[i] = T2[i] + GQG[i];
      if (pmeans) memcpy(pmeans + o * n, mu, sizeof(float) * n);
      if (pcovs) memcpy(pcovs + o * n * n, P, sizeof(float) * n * n);
    }
    free(P);
  }
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
