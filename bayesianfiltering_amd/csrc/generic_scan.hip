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
#include <cstring>
#include <vector>
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "scan_common.hpp"
#include "models.hpp"
#include "bf_rng.hpp"

namespace bf {

struct UViewG {
  const float* p;
  long long sB, sT;
};

struct GenModel {  // pointers are DEVICE pointers into one constant block (const_cache.hip)
  int dyn_id, emi_id, n, dq, m, dr;
  float dth[8], eth[8];
  const float *A, *Hm, *GQG, *DRD, *Gq0, *Dr0, *R, *r0;
  int q_tv, r_tv;  // GQG / DRD hold one matrix per step (the (T, d, d) rule of inference.py:21, :337-340)
  float jitter;
};

template <int NT>
__device__ __forceinline__ void gsync() {
  if constexpr (NT == 64) wave_lds_sync();
  else lds_barrier();
}

// C = (MODE 0) A B | (MODE 1) I + A B | (MODE 2) I - A B,  A [R x Kd] (pitch lda), B [Kd x Cn] (pitch ldb), all in LDS;
// pitches are multiples of 4 floats and rows are padded to them, so the b128 reads of B stay inside its rows.
template <int NT, int MODE>
__device__ __forceinline__ void mm_lds(float* C, int ldc, const float* A, int lda, const float* Bm, int ldb, const float* I,
                                       int ldi, int R, int Kd, int Cn, int tid) {
  const int c4 = (Cn + 3) >> 2;
  for (int e = tid; e < R * c4; e += NT) {
    const int i = e / c4, j = (e - i * c4) * 4;
    const float* ar = A + i * lda;
    float4 b = *reinterpret_cast<const float4*>(Bm + j);
    float a = ar[0];
    float s0 = a * b.x, s1 = a * b.y, s2 = a * b.z, s3 = a * b.w;
    for (int k = 1; k < Kd; ++k) {
      a = ar[k];
      b = *reinterpret_cast<const float4*>(Bm + k * ldb + j);
      s0 = fmaf(a, b.x, s0);
      s1 = fmaf(a, b.y, s1);
      s2 = fmaf(a, b.z, s2);
      s3 = fmaf(a, b.w, s3);
    }
    float s[4] = {s0, s1, s2, s3};
    BF_UNROLL for (int q = 0; q < 4; ++q) if (j + q < Cn) {
      if constexpr (MODE == 0) C[i * ldc + j + q] = s[q];
      else if constexpr (MODE == 1) C[i * ldc + j + q] = I[i * ldi + j + q] + s[q];
      else C[i * ldc + j + q] = I[i * ldi + j + q] - s[q];
    }
  }
}

template <int NT>
__device__ __forceinline__ void transpose_lds(float* D, int ldd, const float* S, int lds_, int R, int Cn, int tid) {
  for (int e = tid; e < R * Cn; e += NT) {
    const int i = e / Cn, j = e - i * Cn;
    D[j * ldd + i] = S[i * lds_ + j];
  }
}

// f(x, q0, u), F_x at x -> LDS (F pitch ld).  Value and Jacobian formulas: csrc/models.hpp (same sources).
template <int NT>
__device__ void gen_dyn_linearize(const GenModel& p, const float* x, float u0, float* F, int ld, float* fx, int tid) {
  const int n = p.n;
  for (int e = tid; e < n * n; e += NT) F[(e / n) * ld + (e % n)] = (p.dyn_id == DYN_LINEAR) ? p.A[e] : 0.f;
  gsync<NT>();
  switch (p.dyn_id) {
    case DYN_LINEAR:
      for (int i = tid; i < n; i += NT) {
        float s = p.A[i * n] * x[0];
        for (int k = 1; k < n; ++k) s = fmaf(p.A[i * n + k], x[k], s);
        fx[i] = s;
      }
      break;
    case DYN_LORENZ96: {
      const float alpha = p.dth[0], beta = p.dth[1], gamma = p.dth[2], dt = p.dth[3];
      const bool mp = p.dth[4] != 0.f;
      for (int i = tid; i < n; i += NT) {
        const int im1 = (i + n - 1) % n, ip1 = (i + 1) % n, im2 = (i + 2 * n - 2) % n;
        const float ax = x[im1];
        const float bx = mp ? (x[ip1] - x[im2]) : 0.f;
        fx[i] = x[i] + dt * (alpha * (ax * bx) - beta * x[i] + gamma);
        // (row i is this lane's alone; the += keep the accumulation order of models.hpp when indices coincide at small n)
        F[i * ld + i] += 1.0f - dt * beta;
        if (mp) {
          F[i * ld + im1] += dt * alpha * bx;
          F[i * ld + ip1] += dt * alpha * ax;
          F[i * ld + im2] += -dt * alpha * ax;
        }
      }
    } break;
    case DYN_LORENZ63:
      if (tid == 0) {
        const float s = p.dth[0], r = p.dth[1], b = p.dth[2], dt = p.dth[3];
        fx[0] = dt * s * (x[1] - x[0]) + x[0];
        fx[1] = dt * (x[0] * r - x[1] - x[0] * x[2]) + x[1];
        fx[2] = dt * (x[0] * x[1] - b * x[2]) + x[2];
        F[0] = 1.f - dt * s;              F[1] = dt * s;            F[2] = 0.f;
        F[ld] = dt * (r - x[2]);          F[ld + 1] = 1.f - dt;     F[ld + 2] = -dt * x[0];
        F[2 * ld] = dt * x[1];            F[2 * ld + 1] = dt * x[0]; F[2 * ld + 2] = 1.f - dt * b;
      }
      break;
    case DYN_MANEUVER_BOT:
      if (tid == 0) {
        const float dt = p.dth[0], acc = p.dth[1];
        const float c0 = 0.5f * (u0 - 1.f) * (u0 - 2.f), c1 = -u0 * (u0 - 2.f), c2 = 0.5f * u0 * (u0 - 1.f);
        float Mx[16] = {c0, c0 * dt, 0, 0, 0, c0, 0, 0, 0, 0, c0, c0 * dt, 0, 0, 0, c0};
        float J[16];
        for (int i = 0; i < 16; ++i) J[i] = 0.f;
        const float xs[4] = {x[0], x[1], x[2], x[3]};
        const float s2 = xs[1] * xs[1] + xs[3] * xs[3];
        const float nrm = sqrtf(s2);
        float sn0, cs0;
        sincosf(dt * (0.1f * acc / nrm), &sn0, &cs0);
        for (int sgn = 0; sgn < 2; ++sgn) {
          const float cc = sgn == 0 ? c1 : c2;
          const float a = sgn == 0 ? acc : -acc;
          const float om = 0.1f * a / nrm;
          const float sn = sgn == 0 ? sn0 : -sn0, cs = cs0;
          const float so = sn / om, co = (1.f - cs) / om;
          const float Fm[16] = {1, so, 0, -co, 0, cs, 0, -sn, 0, co, 1, so, 0, sn, 0, cs};
          const float dso = (dt * cs * om - sn) / (om * om);
          const float dco = (dt * sn * om - (1.f - cs)) / (om * om);
          const float dF[16] = {0, dso, 0, -dco, 0, -dt * sn, 0, -dt * cs, 0, dco, 0, dso, 0, dt * cs, 0, -dt * sn};
          const float dom1 = -om * xs[1] / s2, dom3 = -om * xs[3] / s2;
          for (int i = 0; i < 4; ++i) {
            float dfx = 0.f;
            for (int k = 0; k < 4; ++k) {
              Mx[i * 4 + k] += cc * Fm[i * 4 + k];
              dfx = fmaf(dF[i * 4 + k], xs[k], dfx);
            }
            J[i * 4 + 1] += cc * dfx * dom1;
            J[i * 4 + 3] += cc * dfx * dom3;
          }
        }
        for (int i = 0; i < 4; ++i) {
          float s = 0.f;
          for (int k = 0; k < 4; ++k) {
            s = fmaf(Mx[i * 4 + k], xs[k], s);
            F[i * ld + k] = Mx[i * 4 + k] + J[i * 4 + k];
          }
          fx[i] = s;
        }
      }
      break;
    case DYN_SINE: {
      const float w0 = p.dth[0];
      for (int i = tid; i < n; i += NT) {
        fx[i] = sinf(w0 * x[i]);
        F[i * ld + i] = w0 * cosf(w0 * x[i]);
      }
    } break;
    case DYN_GROWTH:
      if (tid == 0) {
        const float d = 1.f + x[0] * x[0];
        fx[0] = x[0] / 2.0f + 25.0f * x[0] / d + u0;
        F[0] = 0.5f + 25.0f * (1.f - x[0] * x[0]) / (d * d);
      }
      break;
    default: break;
  }
  gsync<NT>();
  for (int i = tid; i < n; i += NT) fx[i] += p.Gq0[i];
}

// h(x, r0, u), H_x at x, H_r R H_r^T -> LDS (H pitch ldh, HrRHr pitch ldr).
template <int NT>
__device__ void gen_emi_linearize(const GenModel& p, const float* x, float u0, long long t, float* H, int ldh, float* hx,
                                  float* HrRHr, int ldr, int tid) {
  const int n = p.n, m = p.m;
  const float* DRD = p.DRD + (p.r_tv ? t * m * m : 0);
  for (int e = tid; e < m * n; e += NT) H[(e / n) * ldh + (e % n)] = (p.emi_id == EMI_LINEAR) ? p.Hm[e] : 0.f;
  for (int e = tid; e < m * m; e += NT) HrRHr[(e / m) * ldr + (e % m)] = DRD[e];
  gsync<NT>();
  switch (p.emi_id) {
    case EMI_LINEAR:
      for (int a = tid; a < m; a += NT) {
        float s = p.Hm[a * n] * x[0];
        for (int k = 1; k < n; ++k) s = fmaf(p.Hm[a * n + k], x[k], s);
        hx[a] = s + p.Dr0[a];
      }
      break;
    case EMI_BEARING_RANGE:
      if (tid == 0) {
        const float d2 = x[0] * x[0] + x[2] * x[2];
        const float d = sqrtf(d2);
        hx[0] = atan2f(x[2], x[0]) + p.Dr0[0];
        hx[1] = d + p.Dr0[1];
        H[0] = -x[2] / d2;   H[2] = x[0] / d2;
        H[ldh] = x[0] / d;   H[ldh + 2] = x[2] / d;
      }
      break;
    case EMI_BEARING:
      if (tid == 0) {
        const float d2 = x[0] * x[0] + x[2] * x[2];
        hx[0] = atan2f(x[2], x[0]) + p.Dr0[0];
        H[0] = -x[2] / d2;
        H[2] = x[0] / d2;
      }
      break;
    case EMI_QUADRATIC:
      if (tid == 0) {
        const float c = p.eth[0];
        float s = 0.f;
        for (int i = 0; i < n; ++i) {
          s = fmaf(x[i], x[i], s);
          H[i] = 2.0f * c * x[i];
        }
        hx[0] = c * s + p.Dr0[0];
      }
      break;
    case EMI_STOCH_VOL: {
      const float sigma = p.eth[0], beta = p.eth[1], c = p.eth[2];
      for (int i = tid; i < n; i += NT) {
        const float e = u0 * beta * expf(x[i] / sigma);
        hx[i] = e * p.r0[i] + (1.f - u0) * (c * x[i] + p.r0[i]);
        H[i * ldh + i] = e * p.r0[i] / sigma + (1.f - u0) * c;
      }
      for (int e2 = tid; e2 < m * m; e2 += NT) {
        const int a = e2 / m, b = e2 % m;
        const float ha = u0 * beta * expf(x[a] / sigma) + (1.f - u0), hb = u0 * beta * expf(x[b] / sigma) + (1.f - u0);
        HrRHr[a * ldr + b] = (ha * p.R[a * m + b]) * hb;
      }
    } break;
    default: break;
  }
  gsync<NT>();
}

template <int NT>
__global__ void __launch_bounds__(NT)
gsf_generic_kernel(GenModel p, CView y, UViewG u, CarryView carry, OutViews out, float* __restrict__ gm, float* __restrict__ gP,
                   long long B, long long T, int K, int KP) {
  const int tid = threadIdx.x;
  const long long b = blockIdx.x;
  const int n = p.n, m = p.m;
  const int ldn = ((n + 3) & ~3) + 4, ldm = ((m + 3) & ~3) + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // ---- carve: P | vectors | weights | region shared by the update scratch and the predict scratch
  float* sP = lds;                   // [n][ldn]
  float* smean = sP + n * ldn;       // [n]   (rounded up to 4)
  const int nv = (n + 3) & ~3, mv_ = (m + 3) & ~3;
  float* sfx = smean + nv;           // [n]
  float* shx = sfx + nv;             // [m]
  float* sv = shx + mv_;             // [m]
  float* sr = sv + mv_;              // [m]  forward-substitution residual / z
  float* srd = sr + mv_;             // [m]  reciprocal LU pivots
  int* sperm = reinterpret_cast<int*>(srd + mv_);  // [m]
  const int KPa = (KP + 3) & ~3;     // keeps the matrices behind 16-byte aligned
  float* sll = reinterpret_cast<float*>(sperm + mv_);  // [KP] log-likelihoods
  float* sw = sll + KPa;             // [KP] weights
  float* stree = sw + KPa;           // [KP] reduction trees
  float* reg = stree + KPa;
  // update scratch
  float* sH = reg;                   // [m][ldn]
  float* sHP = sH + m * ldn;         // [m][ldn]
  float* sX = sHP + m * ldn;         // [m][ldn]
  float* sHT = sX + m * ldn;         // [n][ldm]
  float* sXT = sHT + n * ldm;        // [n][ldm]   K = X^T
  float* sKS = sXT + n * ldm;        // [n][ldm]
  float* sS = sKS + n * ldm;         // [m][ldm]
  float* sa = sS + m * ldm;          // [m][ldm]   LU of S + jitter
  float* sL = sa + m * ldm;          // [m][ldm]   chol(S)
  float* sRR = sL + m * ldm;         // [m][ldm]   H_r R H_r^T
  // predict scratch (aliases the update scratch: dead by then)
  float* sF = reg;                   // [n][ldn]
  float* sFT = sF + n * ldn;         // [n][ldn]
  float* sFP = sFT + n * ldn;        // [n][ldn]

  for (int k = tid; k < KP; k += NT) sw[k] = (k < K) ? (carry.w_in ? carry.w_in[b * K + k] : 1.0f / (float)K) : 0.f;
  // K == 1: the state stays in LDS for the whole scan; K > 1: the components take turns (HBM scratch, L2-resident)
  const float* m_src = carry.m_in + b * (long long)K * n;
  const float* P_src = carry.P_in + b * (long long)K * n * n;
  float* gmb = gm ? gm + b * (long long)K * n : nullptr;
  float* gPb = gP ? gP + b * (long long)K * n * n : nullptr;
  if (K == 1) {
    for (int e = tid; e < n * n; e += NT) sP[(e / n) * ldn + (e % n)] = P_src[e];
    for (int i = tid; i < n; i += NT) smean[i] = m_src[i];
  }
  gsync<NT>();

  for (long long t = 0; t < T; ++t) {
    const float u0 = u.p ? u.p[b * u.sB + t * u.sT] : 0.f;
    const float* GQG = p.GQG + (p.q_tv ? t * n * n : 0);
    for (int k = 0; k < K; ++k) {
      if (K > 1) {
        const float* ms = (t == 0) ? m_src + k * n : gmb + k * n;
        const float* Ps = (t == 0) ? P_src + (long long)k * n * n : gPb + (long long)k * n * n;
        for (int e = tid; e < n * n; e += NT) sP[(e / n) * ldn + (e % n)] = Ps[e];
        for (int i = tid; i < n; i += NT) smean[i] = ms[i];
        gsync<NT>();
      }
      // ================= _condition_on (inference.py:72-105)
      gen_emi_linearize<NT>(p, smean, u0, t, sH, ldn, shx, sRR, ldm, tid);
      for (int a = tid; a < m; a += NT) sv[a] = y.p[b * y.sB + t * y.sT + a * y.sE] - shx[a];
      transpose_lds<NT>(sHT, ldm, sH, ldn, m, n, tid);
      mm_lds<NT, 0>(sHP, ldn, sH, ldn, sP, ldn, nullptr, 0, m, n, n, tid);           // H_x P
      gsync<NT>();
      mm_lds<NT, 1>(sS, ldm, sHP, ldn, sHT, ldm, sRR, ldm, m, n, m, tid);            // S = H_r R H_r^T + (H_x P) H_x^T
      for (int e = tid; e < m * n; e += NT) sX[(e / n) * ldn + (e % n)] = sHP[(e / n) * ldn + (e % n)];
      gsync<NT>();
      for (int e = tid; e < m * m; e += NT) sa[(e / m) * ldm + (e % m)] = sS[(e / m) * ldm + (e % m)] + p.jitter;
      gsync<NT>();
      // ---- psd_solve (utils.py:256-259): getrf with partial pivoting ...
      for (int kk = 0; kk < m; ++kk) {
        int pv = kk;
        float best = fabsf(sa[kk * ldm + kk]);
        for (int i = kk + 1; i < m; ++i) {  // every lane scans the column (broadcast reads): no hand-off needed
          const float val = fabsf(sa[i * ldm + kk]);
          if (val > best) { best = val; pv = i; }
        }
        if (tid == 0) sperm[kk] = pv;
        if (pv != kk) {
          gsync<NT>();
          for (int j = tid; j < m; j += NT) {
            const float a0 = sa[kk * ldm + j], a1 = sa[pv * ldm + j];
            sa[kk * ldm + j] = a1;
            sa[pv * ldm + j] = a0;
          }
        }
        gsync<NT>();
        const float rpiv = fast_rcp(sa[kk * ldm + kk]);
        if (tid == 0) srd[kk] = rpiv;
        const int rem = m - 1 - kk;
        for (int e = tid; e < rem * rem; e += NT) {
          const int i = kk + 1 + e / rem, j = kk + 1 + e % rem;
          const float l = sa[i * ldm + kk] * rpiv;
          sa[i * ldm + j] = fmaf(-l, sa[kk * ldm + j], sa[i * ldm + j]);
        }
        gsync<NT>();
      }
      // ... and getrs, one right-hand side (column of H P) per lane
      for (int c = tid; c < n; c += NT) {
        for (int kk = 0; kk < m; ++kk) {
          const int pv = sperm[kk];
          if (pv != kk) {
            const float x0 = sX[kk * ldn + c], x1 = sX[pv * ldn + c];
            sX[kk * ldn + c] = x1;
            sX[pv * ldn + c] = x0;
          }
        }
        for (int kk = 0; kk < m; ++kk) {
          const float xk = sX[kk * ldn + c], rp = srd[kk];
          for (int i = kk + 1; i < m; ++i) sX[i * ldn + c] = fmaf(-(sa[i * ldm + kk] * rp), xk, sX[i * ldn + c]);
        }
        for (int i = m - 1; i >= 0; --i) {
          float s = sX[i * ldn + c];
          for (int q = i + 1; q < m; ++q) s = fmaf(-sa[i * ldm + q], sX[q * ldn + c], s);
          sX[i * ldn + c] = s * srd[i];
        }
      }
      gsync<NT>();
      transpose_lds<NT>(sXT, ldm, sX, ldn, m, n, tid);                                 // K = X^T
      gsync<NT>();
      mm_lds<NT, 0>(sKS, ldm, sXT, ldm, sS, ldm, nullptr, 0, n, m, m, tid);            // K S (un-jittered S)
      for (int i = tid; i < n; i += NT) {                                               // m+ = m + K v
        float s = sXT[i * ldm] * sv[0];
        for (int a = 1; a < m; ++a) s = fmaf(sXT[i * ldm + a], sv[a], s);
        smean[i] += s;
      }
      gsync<NT>();
      mm_lds<NT, 2>(sP, ldn, sKS, ldm, sX, ldn, sP, ldn, n, m, n, tid);               // P+ = P - (K S) K^T
      // ---- log N(y; h(m), S) through chol(S) (inference.py:104, :24), left-looking, row per lane
      for (int j = 0; j < m; ++j) {
        float d = sS[j * ldm + j];
        for (int q = 0; q < j; ++q) d = fmaf(-sL[j * ldm + q], sL[j * ldm + q], d);
        d = fast_sqrt(d);
        const float inv = fast_rcp(d);
        for (int i = j + tid; i < m; i += NT) {
          if (i == j) {
            sL[j * ldm + j] = d;
          } else {
            float s = sS[i * ldm + j];
            for (int q = 0; q < j; ++q) s = fmaf(-sL[i * ldm + q], sL[j * ldm + q], s);
            sL[i * ldm + j] = s * inv;
          }
        }
        gsync<NT>();
      }
      for (int a = tid; a < m; a += NT) sr[a] = sv[a];
      gsync<NT>();
      for (int j = 0; j < m; ++j) {
        const float zj = sr[j] * fast_rcp(sL[j * ldm + j]);
        gsync<NT>();
        for (int i = j + tid; i < m; i += NT) {
          if (i == j) sr[j] = zj;
          else sr[i] = fmaf(-sL[i * ldm + j], zj, sr[i]);
        }
        gsync<NT>();
      }
      if (tid == 0) {
        float quad = 0.f, logdet = 0.f;
        for (int i = 0; i < m; ++i) {
          quad = fmaf(sr[i], sr[i], quad);
          logdet += fast_log(sL[i * ldm + i]);
        }
        const float ll = -0.5f * quad - 0.5f * (float)m * 1.8378770664093453f - logdet;
        sll[k] = ll;
        if (out.ll.p) out.ll.p[b * out.ll.sB + k * out.ll.sK + t * out.ll.sT] = ll;
      }
      gsync<NT>();
      // filtered streams
      if (out.m.p) for (int i = tid; i < n; i += NT) out.m.p[b * out.m.sB + k * out.m.sK + t * out.m.sT + i * out.m.sE] = smean[i];
      if (out.P.p) for (int e = tid; e < n * n; e += NT)
          out.P.p[b * out.P.sB + k * out.P.sK + t * out.P.sT + e * out.P.sE] = sP[(e / n) * ldn + (e % n)];
      // ================= _predict (inference.py:51-70)
      gen_dyn_linearize<NT>(p, smean, u0, sF, ldn, sfx, tid);
      gsync<NT>();
      transpose_lds<NT>(sFT, ldn, sF, ldn, n, n, tid);
      mm_lds<NT, 0>(sFP, ldn, sF, ldn, sP, ldn, nullptr, 0, n, n, n, tid);            // F_x P+
      gsync<NT>();
      for (int e = tid; e < n * n; e += NT) sP[(e / n) * ldn + (e % n)] = GQG[e];     // P- = (F_x P+) F_x^T + F_q Q F_q^T
      for (int i = tid; i < n; i += NT) smean[i] = sfx[i];
      gsync<NT>();
      mm_lds<NT, 1>(sP, ldn, sFP, ldn, sFT, ldn, sP, ldn, n, n, n, tid);
      gsync<NT>();
      if (out.pm.p) for (int i = tid; i < n; i += NT) out.pm.p[b * out.pm.sB + k * out.pm.sK + t * out.pm.sT + i * out.pm.sE] = smean[i];
      if (out.pP.p) for (int e = tid; e < n * n; e += NT)
          out.pP.p[b * out.pP.sB + k * out.pP.sK + t * out.pP.sT + e * out.pP.sE] = sP[(e / n) * ldn + (e % n)];
      if (K > 1) {
        for (int e = tid; e < n * n; e += NT) gPb[(long long)k * n * n + e] = sP[(e / n) * ldn + (e % n)];
        for (int i = tid; i < n; i += NT) gmb[k * n + i] = smean[i];
        __syncthreads();  // global + LDS: the next component reuses the tile, the next step reads this component back
      }
    }
    // ================= reweight (inference.py:347-350): lls -= max; w = exp(lls) * w; w /= sum(w), adjacent-pair trees
    for (int k = tid; k < KP; k += NT) stree[k] = (k < K) ? sll[k] : -__builtin_inff();
    gsync<NT>();
    for (int s = 1; s < KP; s <<= 1) {
      for (int k = tid * 2 * s; k + s < KP; k += NT * 2 * s) {
        const float a = stree[k], c = stree[k + s];
        stree[k] = (a != a || c != c) ? __builtin_nanf("") : fmaxf(a, c);  // jnp.max propagates NaN
      }
      gsync<NT>();
    }
    const float mx = stree[0];
    gsync<NT>();
    for (int k = tid; k < KP; k += NT) {
      const float e = (k < K) ? expf(sll[k] - mx) * sw[k] : 0.f;
      sw[k] = e;
      stree[k] = e;
    }
    gsync<NT>();
    for (int s = 1; s < KP; s <<= 1) {
      for (int k = tid * 2 * s; k + s < KP; k += NT * 2 * s) stree[k] += stree[k + s];
      gsync<NT>();
    }
    const float tot = stree[0];
    gsync<NT>();
    for (int k = tid; k < K; k += NT) {
      const float wn = sw[k] / tot;
      sw[k] = wn;
      if (out.w.p) out.w.p[b * out.w.sB + k * out.w.sK + t * out.w.sT] = wn;
    }
    gsync<NT>();
  }

  // ---- carry out
  if (K == 1) {
    if (carry.P_out) for (int e = tid; e < n * n; e += NT) carry.P_out[b * (long long)n * n + e] = sP[(e / n) * ldn + (e % n)];
    if (carry.m_out) for (int i = tid; i < n; i += NT) carry.m_out[b * (long long)n + i] = smean[i];
  } else {
    // the HBM scratch IS the carry when the caller asked for it; otherwise copy nothing
    if (carry.P_out && carry.P_out != gP)
      for (long long e = tid; e < (long long)K * n * n; e += NT) carry.P_out[b * (long long)K * n * n + e] = gPb[e];
    if (carry.m_out && carry.m_out != gm)
      for (long long e = tid; e < (long long)K * n; e += NT) carry.m_out[b * (long long)K * n + e] = gmb[e];
  }
  if (carry.w_out) for (int k = tid; k < K; k += NT) carry.w_out[b * K + k] = sw[k];
}

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
    default: return set_error(BF_EUNSUPPORTED, "unknown emission function id %d", p->emi_id);
  }
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  if (p->R_steps > 1 && p->emi_id == EMI_STOCH_VOL)
    return set_error(BF_EUNSUPPORTED, "time-varying R needs an emission with a constant noise Jacobian H_r");
  g.q_tv = p->Q_steps > 1;
  g.r_tv = p->R_steps > 1;
  const int qs = g.q_tv ? p->Q_steps : 1, rs = g.r_tv ? p->R_steps : 1;
  // block layout (floats): A | Hm | Gq0 | Dr0 | R | r0 | GQG[qs] | DRD[rs]
  const size_t oA = 0, oH = oA + (size_t)n * n, oGq = oH + (size_t)m * n, oDr = oGq + n, oR = oDr + m, or0 = oR + (size_t)dr * dr,
               oGQG = or0 + dr, oDRD = oGQG + (size_t)qs * n * n, total = oDRD + (size_t)rs * m * m;
  blk.assign(total, 0.f);
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
  for (int i = 0; i < dr * dr; ++i) blk[oR + i] = p->R[i];
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
  return BF_OK;
}

int launch_gsf_generic(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                       const bf_carry* carry, const bf_out_desc* out, hipStream_t stream) {
  if (out->coll_mean.ptr || out->coll_cov.ptr)
    return set_error(BF_EUNSUPPORTED, "collapsed streams inside the scan need a compiled instance (n <= 8, m <= 4, K x lanes <= 256); "
                                      "use bf_collapse_f32 on the emitted streams");
  GenModel g;
  std::vector<float> blk;
  int rc = gen_fill(p, T, g, blk);
  if (rc != BF_OK) return rc;
  int KP = 1;
  while (KP < K) KP <<= 1;
  const size_t lds_bytes = sizeof(float) * gen_lds_floats(p->n, p->m, KP);
  if (lds_bytes > 160 * 1024)
    return set_error(BF_EUNSUPPORTED, "generic scan: n = %d, m = %d, K = %d need %zu bytes of LDS (160 KiB per workgroup)", p->n, p->m, K, lds_bytes);
  const void* dv = nullptr;
  rc = device_constants(blk.data(), sizeof(float) * blk.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float* base = static_cast<const float*>(dv);
  auto fix = [&](const float*& q) { q = base + reinterpret_cast<size_t>(q); };
  fix(g.A); fix(g.Hm); fix(g.Gq0); fix(g.Dr0); fix(g.R); fix(g.r0); fix(g.GQG); fix(g.DRD);

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
  if (p->n <= 16 && p->m <= 16) {
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
