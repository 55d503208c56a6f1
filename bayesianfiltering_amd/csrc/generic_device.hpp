// generic_device.hpp: the device side of the run-time-dimension scan (see generic_scan.hip for the design notes).
// It is compiled twice: ahead of time into libbayesfilt_hip.so (generic_scan.hip), and at RUN time by hiprtc together
// with a user's own dynamics / emission functions (user_model.hip: BF_JIT, BF_USER_DYN / BF_USER_EMI) -- the text of this
// file is embedded in the library for that purpose, so it must stay self-contained under BF_JIT.
#pragma once
#ifdef BF_JIT
// hiprtc build: the few helpers the ahead-of-time build takes from bf_common.hpp / kf_math.hpp / scan_common.hpp / models.hpp
namespace bf {
#define BF_UNROLL _Pragma("unroll")
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
enum { DYN_LINEAR = 0, DYN_LORENZ96 = 1, DYN_LORENZ63 = 2, DYN_MANEUVER_BOT = 3, DYN_SINE = 4, DYN_GROWTH = 5 };
enum { EMI_LINEAR = 0, EMI_BEARING_RANGE = 1, EMI_QUADRATIC = 2, EMI_STOCH_VOL = 3, EMI_BEARING = 4 };
struct SView {
  float* p;
  long long sB, sK, sT, sE;
};
struct CView {
  const float* p;
  long long sB, sT, sE;
};
struct OutViews {
  SView w, m, P, pm, pP, ll;
  SView cm, cP;
};
struct CarryView {
  const float* w_in;
  const float* m_in;
  const float* P_in;
  float* w_out;
  float* m_out;
  float* P_out;
};
}  // namespace bf
#endif

namespace bf {

enum { DYN_USER = 100, EMI_USER = 100 };  // functions compiled at run time from the caller's source (user_model.hip)

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
  // user-defined functions (DYN_USER / EMI_USER): raw noise moments and the caller's parameter vectors
  const float *q0, *Q, *dyn_theta, *emi_theta;
};

#if defined(BF_USER_DYN) || defined(BF_USER_EMI)
// Scratch of the user-function linearisations (LDS, after everything else): their noise Jacobians are not constant, so
// F_q Q F_q^T / H_r R H_r^T are formed on the device at every step (inference.py:69, :100).
struct UserScratch {
  float *Jn, *JnT, *Cov, *T1, *Out;  // noise Jacobian [rows][ldc], its transpose [cols][ldr], covariance [cols][ldc], J Cov, J Cov J^T
};
#endif

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
  // every operation rounded on its own, in the order written: the ahead-of-time build and a run-time build of the same
  // formulas (a user's source twin of a registry function, differentiated by dual numbers) then agree bit for bit
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
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

#if defined(BF_USER_DYN) || defined(BF_USER_EMI)
// J Cov J^T for a noise Jacobian J [rows x cols] just written to sc.Jn (LDS): Out [rows][ldo]
template <int NT>
__device__ void user_noise_cov(const float* cov_g, int rows, int cols, int ldo, float* Jn, float* JnT, float* Cov, float* T1,
                               float* Out, int tid) {
  const int ldc = ((cols + 3) & ~3) + 4, ldr = ((rows + 3) & ~3) + 4;
  for (int e = tid; e < cols * cols; e += NT) Cov[(e / cols) * ldc + (e % cols)] = cov_g[e];
  transpose_lds<NT>(JnT, ldr, Jn, ldc, rows, cols, tid);
  gsync<NT>();
  mm_lds<NT, 0>(T1, ldc, Jn, ldc, Cov, ldc, nullptr, 0, rows, cols, cols, tid);      // (J Cov)
  gsync<NT>();
  mm_lds<NT, 0>(Out, ldo, T1, ldc, JnT, ldr, nullptr, 0, rows, cols, rows, tid);     // (J Cov) J^T
  gsync<NT>();
}
#endif

#ifdef BF_USER_DYN
// f(m, q0, u) and its Jacobians w.r.t. the state and the noise by forward-mode dual numbers (the reference's
// jacfwd(f, 0), jacfwd(f, 1) at (m, q0, u): inference.py:328-329, :66-67): lane d evaluates the user's function once with
// the unit seed in direction d of (x, q) and owns column d of [F_x | F_q].  Then F_q Q_t F_q^T (inference.py:69).
template <int NT>
__device__ void user_dyn_linearize(const GenModel& p, const float* x, float u0, long long t, float* F, int ld, float* fx, float* ureg,
                                   int tid) {
#pragma clang fp contract(off)
  constexpr int N = BF_N, DQ = BF_DQ;
  constexpr int ldc = ((DQ + 3) & ~3) + 4, ldr = ((N + 3) & ~3) + 4;
  float* Out = ureg;                 // [N][ld]   F_q Q F_q^T
  float* Jn = Out + N * ld;          // [N][ldc]  F_q
  float* JnT = Jn + N * ldc;         // [DQ][ldr]
  float* Cov = JnT + DQ * ldr;       // [DQ][ldc]
  float* T1 = Cov + DQ * ldc;        // [N][ldc]
  for (int d = tid; d < N + DQ; d += NT) {
    bfu::Dual xs[N], qs[DQ], out[N];
    BF_UNROLL for (int i = 0; i < N; ++i) xs[i] = bfu::Dual(x[i], i == d ? 1.0f : 0.0f);
    BF_UNROLL for (int k = 0; k < DQ; ++k) qs[k] = bfu::Dual(p.q0[k], (N + k) == d ? 1.0f : 0.0f);
    bfu::dynamics<bfu::Dual>(xs, qs, bfu::Dual(u0), p.dyn_theta, out);
    if (d < N) {
      BF_UNROLL for (int i = 0; i < N; ++i) F[i * ld + d] = out[i].d;
    } else {
      BF_UNROLL for (int i = 0; i < N; ++i) Jn[i * ldc + (d - N)] = out[i].d;
    }
    if (d == 0) BF_UNROLL for (int i = 0; i < N; ++i) fx[i] = out[i].v;
  }
  gsync<NT>();
  user_noise_cov<NT>(p.Q + (p.q_tv ? t * DQ * DQ : 0), N, DQ, ld, Jn, JnT, Cov, T1, Out, tid);
}
#endif

#ifdef BF_USER_EMI
// h(m, r0, u), H_x, H_r by dual numbers (jacfwd(h, 0), jacfwd(h, 1): inference.py:328-329, :98-99) and H_r R_t H_r^T (:100)
template <int NT>
__device__ void user_emi_linearize(const GenModel& p, const float* x, float u0, long long t, float* H, int ldh, float* hx,
                                   float* HrRHr, int ldo, float* ureg, int tid) {
#pragma clang fp contract(off)
  constexpr int N = BF_N, M = BF_M, DR = BF_DR;
  constexpr int ldc = ((DR + 3) & ~3) + 4, ldr = ((M + 3) & ~3) + 4;
  float* Jn = ureg + N * (((N + 3) & ~3) + 4);   // behind the dynamics' output slot (so the two carve-ups never overlap a live F_q Q F_q^T)
  float* JnT = Jn + M * ldc;
  float* Cov = JnT + DR * ldr;
  float* T1 = Cov + DR * ldc;
  for (int d = tid; d < N + DR; d += NT) {
    bfu::Dual xs[N], rs[DR], out[M];
    BF_UNROLL for (int i = 0; i < N; ++i) xs[i] = bfu::Dual(x[i], i == d ? 1.0f : 0.0f);
    BF_UNROLL for (int k = 0; k < DR; ++k) rs[k] = bfu::Dual(p.r0[k], (N + k) == d ? 1.0f : 0.0f);
    bfu::emission<bfu::Dual>(xs, rs, bfu::Dual(u0), p.emi_theta, out);
    if (d < N) {
      BF_UNROLL for (int a = 0; a < M; ++a) H[a * ldh + d] = out[a].d;
    } else {
      BF_UNROLL for (int a = 0; a < M; ++a) Jn[a * ldc + (d - N)] = out[a].d;
    }
    if (d == 0) BF_UNROLL for (int a = 0; a < M; ++a) hx[a] = out[a].v;
  }
  gsync<NT>();
  user_noise_cov<NT>(p.R + (p.r_tv ? t * DR * DR : 0), M, DR, ldo, Jn, JnT, Cov, T1, HrRHr, tid);
}
#endif

template <int NT>
__device__ __forceinline__ void gsf_generic_body(const GenModel& p, CView y, UViewG u, CarryView carry, OutViews out,
                                                 float* __restrict__ gm, float* __restrict__ gP, long long B, long long T, int K,
                                                 int KP) {
  // every fused multiply-add of this body is written out; with contraction off the ahead-of-time build and a run-time
  // (hiprtc) build of the same text produce the same bits whatever the two compilers' fusion heuristics are
#pragma clang fp contract(off)
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
#if defined(BF_USER_DYN) || defined(BF_USER_EMI)
  // user-function scratch behind the larger of the two regions above
  float* ureg;
  {
    const long upd = 3L * m * ldn + 3L * n * ldm + 4L * m * ldm, prd = 3L * n * ldn;
    ureg = reg + (upd > prd ? upd : prd);
  }
#endif

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
#ifdef BF_USER_EMI
      if (p.emi_id == EMI_USER) user_emi_linearize<NT>(p, smean, u0, t, sH, ldn, shx, sRR, ldm, ureg, tid);
      else
#endif
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
      const float* gqg_src = GQG;   // F_q Q F_q^T: a constant from the host, or formed on the device for a user function
      int gqg_ld = n;
#ifdef BF_USER_DYN
      if (p.dyn_id == DYN_USER) {
        user_dyn_linearize<NT>(p, smean, u0, t, sF, ldn, sfx, ureg, tid);
        gqg_src = ureg;             // UserScratch::Out sits first
        gqg_ld = ldn;
      } else
#endif
      gen_dyn_linearize<NT>(p, smean, u0, sF, ldn, sfx, tid);
      gsync<NT>();
      transpose_lds<NT>(sFT, ldn, sF, ldn, n, n, tid);
      mm_lds<NT, 0>(sFP, ldn, sF, ldn, sP, ldn, nullptr, 0, n, n, n, tid);            // F_x P+
      gsync<NT>();
      for (int e = tid; e < n * n; e += NT) sP[(e / n) * ldn + (e % n)] = gqg_src[(e / n) * gqg_ld + (e % n)];  // P- = (F_x P+) F_x^T + F_q Q F_q^T
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

#ifndef BF_JIT
template <int NT>
__global__ void __launch_bounds__(NT)
gsf_generic_kernel(GenModel p, CView y, UViewG u, CarryView carry, OutViews out, float* __restrict__ gm, float* __restrict__ gP,
                   long long B, long long T, int K, int KP) {
  gsf_generic_body<NT>(p, y, u, carry, out, gm, gP, B, T, K, KP);
}
#endif


}  // namespace bf
