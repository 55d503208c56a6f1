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

/* ---------------------------------------------------------------------------------------------------------------------
 * Gaussian-sum filter (bank of K extended Kalman filters + weight update) for the Lorenz-96 dynamics of
 * gaussfiltax/nonlinearities.py:37-50 with identity noise input and a LINEAR emission H (the even-state emission g96 as
 * a matrix): the scan body of gaussian_sum_filter (inference.py:333-371) as restated by oracle/gaussfilt_oracle.py
 * (_condition_on per component, reweight, _predict), fp32, libm.  BASELINE configs[2]'s CPU baseline ("port").
 * theta = (alpha, beta, gamma, dt, mode); Q [n,n], R [m,m], q0 [n], r0 [m]; init means [B][K][n], P0 [n,n];
 * outputs (any may be NULL): weights [B][K][T], means [B][K][T][n], covs [B][K][T][n][n]. */
static void l96_value_jac(const float* th, const float* x, int n, float* fx, float* F) {
  const float alpha = th[0], beta = th[1], gamma = th[2], dt = th[3];
  const int mp = th[4] != 0.f;
  for (int i = 0; i < n * n; ++i) F[i] = 0.f;
  for (int i = 0; i < n; ++i) {
    const int im1 = (i + n - 1) % n, ip1 = (i + 1) % n, im2 = (i + 2 * n - 2) % n;
    const float ax = x[im1];
    const float bx = mp ? (x[ip1] - x[im2]) : 0.f;
    fx[i] = x[i] + dt * (alpha * (ax * bx) - beta * x[i] + gamma);
    F[i * n + i] += 1.0f - dt * beta;
    if (mp) {
      F[i * n + im1] += dt * alpha * bx;
      F[i * n + ip1] += dt * alpha * ax;
      F[i * n + im2] -= dt * alpha * ax;
    }
  }
}

int oracle_gsf_lorenz96_f32(int n, int m, int K, const float* theta, const float* H, const float* Q, const float* R, const float* q0,
                            const float* r0, const float* y /* [B][T][m] */, int64_t B, int64_t T, const float* init_means,
                            const float* P0, float* w_out, float* means, float* covs, int nthreads) {
  if (n <= 0 || m <= 0 || K <= 0 || n > MAXN || m > MAXN) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    const size_t per = (size_t)n + (size_t)n * n;
    float* st = (float*)malloc(sizeof(float) * ((size_t)K * per + 5 * (size_t)n * n + 3 * (size_t)m * n + 3 * (size_t)m * m + 4 * MAXN + 2 * (size_t)K));
    float* mus = st;                       /* [K][n]   predicted means */
    float* Ps = mus + (size_t)K * n;       /* [K][n][n] predicted covariances */
    float* F = Ps + (size_t)K * n * n;
    float* FP = F + n * n;
    float* T2 = FP + n * n;
    float* Pf = T2 + n * n;
    float* HP = Pf + n * n;                /* m*n */
    float* X = HP + m * n;
    float* KS = X + m * n;                 /* n*m */
    float* S = KS + n * m;
    float* a = S + m * m;
    float* L = a + m * m;
    float* v = L + m * m;
    float* z = v + MAXN;
    float* mf = z + MAXN;
    float* fx = mf + MAXN;
    float* lls = fx + MAXN;
    float* w = lls + K;
    for (int k = 0; k < K; ++k) {
      memcpy(mus + (size_t)k * n, init_means + ((size_t)b * K + k) * n, sizeof(float) * n);
      memcpy(Ps + (size_t)k * n * n, P0, sizeof(float) * n * n);
      w[k] = 1.0f / (float)K;
    }
    for (int64_t t = 0; t < T; ++t) {
      const float* yt = y + (b * T + t) * m;
      for (int k = 0; k < K; ++k) {
        float* mu = mus + (size_t)k * n;
        float* P = Ps + (size_t)k * n * n;
        /* _condition_on (inference.py:72-105), h(x, r) = H x + r */
        for (int i = 0; i < m; ++i) {
          float s = 0.f;
          for (int q = 0; q < n; ++q) s += H[i * n + q] * mu[q];
          v[i] = yt[i] - (s + r0[i]);
        }
        mm(H, P, HP, m, n, n);
        mm_nt(HP, H, S, m, n, m);
        for (int i = 0; i < m * m; ++i) S[i] = R[i] + S[i];
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
            Pf[i * n + j] = P[i * n + j] - s;
          }
        for (int i = 0; i < n; ++i) {
          float s = 0.f;
          for (int q = 0; q < m; ++q) s += X[q * n + i] * v[q];
          mf[i] = mu[i] + s;
        }
        lls[k] = mvn_logpdf(S, v, m, L, z);
        const int64_t o = ((b * K + k) * T + t);
        if (means) memcpy(means + o * n, mf, sizeof(float) * n);
        if (covs) memcpy(covs + o * n * n, Pf, sizeof(float) * n * n);
        /* _predict (inference.py:51-70): F_x at the filtered mean, identity noise input */
        l96_value_jac(theta, mf, n, fx, F);
        for (int i = 0; i < n; ++i) mu[i] = fx[i] + q0[i];
        mm(F, Pf, FP, n, n, n);
        mm_nt(FP, F, T2, n, n, n);
        for (int i = 0; i < n * n; ++i) P[i] = T2[i] + Q[i];
      }
      /* reweight (inference.py:347-350): linear-domain weights, max-subtracted within the step */
      float mx = lls[0];
      for (int k = 1; k < K; ++k) mx = (lls[k] > mx || lls[k] != lls[k]) ? lls[k] : mx;
      float tot = 0.f;
      for (int k = 0; k < K; ++k) { w[k] = expf(lls[k] - mx) * w[k]; tot += w[k]; }
      for (int k = 0; k < K; ++k) {
        w[k] = w[k] / tot;
        if (w_out) w_out[(b * K + k) * T + t] = w[k];
      }
    }
    free(st);
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Bootstrap particle filter (inference.py:1302-1380 as restated by oracle/gaussfilt_oracle.py::bootstrap_particle_filter,
 * arith = "libm") for the same Lorenz-96 dynamics, identity noise input with DIAGONAL Q, the even-state emission and a
 * diagonal Gaussian log-density: Threefry-2x32 keys in JAX's split / bits layout (oracle/threefry.py), normals by XLA's
 * erf_inv polynomial, linear-domain weights, ESS rule, multinomial resampling through a cumulative sum and a left binary
 * search.  BASELINE configs[3]'s CPU baseline ("port"): summaries only (weighted mean [B][T][n], resampled flags [B][T]). */
static uint32_t rotl32_(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
static void tf2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t* o0, uint32_t* o1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  static const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  x0 += ks[0];
  x1 += ks[1];
  for (int i = 0; i < 5; ++i) {
    for (int r = 0; r < 4; ++r) { x0 += x1; x1 = rotl32_(x1, rot[i % 2][r]); x1 ^= x0; }
    x0 += ks[(i + 1) % 3];
    x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
  }
  *o0 = x0;
  *o1 = x1;
}
static uint32_t tf_bits(uint32_t k0, uint32_t k1, uint32_t i, uint32_t count) { /* element i of threefry_2x32(key, iota(count)) */
  const uint32_t h = (count + 1u) >> 1, j = i < h ? i : i - h, c1 = (h + j < count) ? h + j : 0u;
  uint32_t a, b2;
  tf2x32(k0, k1, j, c1, &a, &b2);
  return i < h ? a : b2;
}
static float bits_unit(uint32_t bits) {
  union { uint32_t u; float f; } c;
  c.u = (bits >> 9) | 0x3F800000u;
  return c.f - 1.0f;
}
static float erfinv_xla(float x) {
  float w = -log1pf(-x * x), p;
  static const float lt[9] = {2.81022636e-08f, 3.43273939e-07f, -3.5233877e-06f, -4.39150654e-06f, 0.00021858087f, -0.00125372503f,
                              -0.00417768164f, 0.246640727f, 1.50140941f};
  static const float ge[9] = {-0.000200214257f, 0.000100950558f, 0.00134934322f, -0.00367342844f, 0.00573950773f, -0.0076224613f,
                              0.00943887047f, 1.00167406f, 2.83297682f};
  const float* c = w < 5.0f ? lt : ge;
  w = w < 5.0f ? w - 2.5f : sqrtf(w) - 3.0f;
  p = c[0];
  for (int i = 1; i < 9; ++i) p = c[i] + p * w;
  return p * x;
}
static float bits_normal(uint32_t bits) {
  const float lo = -0.99999994f;
  float u = bits_unit(bits) * (1.0f - lo) + lo;
  u = u > lo ? u : lo;
  return 1.41421356237309515f * erfinv_xla(u);
}

int oracle_bpf_lorenz96_f32(int n, int m, int N, const float* theta, const float* q0, const float* sdq /* sqrt(diag Q) */,
                            const float* lp_sd /* sqrt(diag R_lp) */, const float* m0, const float* sd0 /* sqrt(diag P0) */,
                            const float* y /* [B][T][m] */, int64_t B, int64_t T, const uint32_t key[2], float ess_threshold,
                            float* mean_out /* [B][T][n] */, float* resampled /* [B][T] */, int nthreads) {
  if (n <= 0 || m <= 0 || N <= 0 || n > MAXN || 2 * m > n + 1) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  float lpc = -0.5f * (float)m * 1.8378770664093453f;
  for (int i = 0; i < m; ++i) lpc -= logf(lp_sd[i]);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t b = 0; b < B; ++b) {
    float* x = (float*)malloc(sizeof(float) * ((size_t)2 * N * n + 4 * (size_t)N));
    float* xn = x + (size_t)N * n;
    float* w = xn + (size_t)N * n;
    float* wn = w + N;
    float* cdf = wn + N;
    float* ll = cdf + N;
    int* idx = (int*)malloc(sizeof(int) * (size_t)N);
    uint32_t k0 = key[0], k1 = key[1];
    /* keys = split(key, N + 1); next = keys[0]; x_i ~ N(m0, diag) with keys[1 + i] (:1369-1373) */
    for (int i = 0; i < N; ++i) {
      const uint32_t ka = tf_bits(k0, k1, 2u * (uint32_t)(i + 1), 2u * (uint32_t)(N + 1)), kb = tf_bits(k0, k1, 2u * (uint32_t)(i + 1) + 1u, 2u * (uint32_t)(N + 1));
      for (int d = 0; d < n; ++d) x[(size_t)i * n + d] = m0[d] + sd0[d] * bits_normal(tf_bits(ka, kb, (uint32_t)d, (uint32_t)n));
      w[i] = 1.0f / (float)N;
    }
    { const uint32_t a = tf_bits(k0, k1, 0u, 2u * (uint32_t)(N + 1)), c = tf_bits(k0, k1, 1u, 2u * (uint32_t)(N + 1)); k0 = a; k1 = c; }
    for (int64_t t = 0; t < T; ++t) {
      const float* yt = y + (b * T + t) * m;
      const uint32_t nk0 = tf_bits(k0, k1, 0u, 2u * (uint32_t)(N + 1)), nk1 = tf_bits(k0, k1, 1u, 2u * (uint32_t)(N + 1));
      float mx = -INFINITY;
      for (int i = 0; i < N; ++i) {
        const uint32_t ka = tf_bits(k0, k1, 2u * (uint32_t)(i + 1), 2u * (uint32_t)(N + 1)), kb = tf_bits(k0, k1, 2u * (uint32_t)(i + 1) + 1u, 2u * (uint32_t)(N + 1));
        const float* xi = x + (size_t)i * n;
        float* xo = xn + (size_t)i * n;
        const float alpha = theta[0], beta = theta[1], gamma = theta[2], dt = theta[3];
        const int mp = theta[4] != 0.f;
        for (int d = 0; d < n; ++d) {
          const float ax = xi[(d + n - 1) % n];
          const float bx = mp ? (xi[(d + 1) % n] - xi[(d + 2 * n - 2) % n]) : 0.f;
          const float q = q0[d] + sdq[d] * bits_normal(tf_bits(ka, kb, (uint32_t)d, (uint32_t)n));
          xo[d] = (xi[d] + dt * (alpha * (ax * bx) - beta * xi[d] + gamma)) + q;
        }
        float quad = 0.f;
        for (int a2 = 0; a2 < m; ++a2) {
          const float zz = (yt[a2] - xo[2 * a2]) / lp_sd[a2];
          quad += zz * zz;
        }
        ll[i] = -0.5f * quad + lpc;
        mx = (ll[i] > mx || ll[i] != ll[i]) ? ll[i] : mx;
      }
      float tot = 0.f;
      for (int i = 0; i < N; ++i) { wn[i] = expf(ll[i] - mx) * w[i]; tot += wn[i]; }
      float s2 = 0.f;
      for (int i = 0; i < N; ++i) { wn[i] /= tot; s2 += wn[i] * wn[i]; }
      const int rs = (1.0f / s2) < ess_threshold * (float)N;
      if (rs) { /* utils.py:207-214: keys = split(key, 2); idx = choice(keys[0], N, (N,), p = w); key = keys[1] */
        const uint32_t c0 = tf_bits(nk0, nk1, 0u, 4u), c1 = tf_bits(nk0, nk1, 1u, 4u);
        float acc = 0.f;
        for (int i = 0; i < N; ++i) { acc += wn[i]; cdf[i] = acc; }
        for (int i = 0; i < N; ++i) {
          const float r = cdf[N - 1] * (1.0f - bits_unit(tf_bits(c0, c1, (uint32_t)i, (uint32_t)N)));
          int lo = 0, hi = N;
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] < r) lo = mid + 1; else hi = mid; }
          idx[i] = lo < N ? lo : N - 1;
        }
        for (int i = 0; i < N; ++i) { memcpy(x + (size_t)i * n, xn + (size_t)idx[i] * n, sizeof(float) * n); w[i] = 1.0f / (float)N; }
        k0 = tf_bits(nk0, nk1, 2u, 4u);
        k1 = tf_bits(nk0, nk1, 3u, 4u);
      } else {
        memcpy(x, xn, sizeof(float) * (size_t)N * n);
        memcpy(w, wn, sizeof(float) * N);
        k0 = nk0;
        k1 = nk1;
      }
      if (resampled) resampled[b * T + t] = rs ? 1.0f : 0.0f;
      if (mean_out)
        for (int d = 0; d < n; ++d) {
          float s = 0.f;
          for (int i = 0; i < N; ++i) s += w[i] * x[(size_t)i * n + d];
          mean_out[(b * T + t) * n + d] = s;
        }
    }
    free(x);
    free(idx);
  }
  return 0;
}
