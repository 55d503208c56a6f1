// kf_scan_mfma: batched Kalman filter for n = 64, m = 32 on the fp32 matrix cores.
//
// Same recursion as kf_scan_group.hip -- the lax.scan body of gaussian_sum_filter
// (gaussfiltax/inference.py:333-371) for linear f/h and one component: _condition_on (:72-105),
// reweight (:347-350), _predict (:51-70) -- but at state_dim 64 one step is ~2 MFLOP of dense
// 64x64 / 32x64 products (60-120 flop per output byte, above the fp32 ridge),
// so the covariance algebra runs on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fmaf chain).
//
// Mapping (gfx950).  One workgroup of 4 waves per trajectory; wave w owns the 32x32 output tile
// (ti, tj) = (w >> 1, w & 1) of every 64x64 product.  P lives in LDS (65-float pitch: both the
// row-indexed and the column-indexed MFMA operand patterns are bank-conflict free) and, tile by
// tile, in the accumulator registers of its owner wave across steps; the constant operands (the
// wave's row blocks of A and its tile of G Q G^T) stay in VGPRs in MFMA operand layout for the
// whole scan; H and D R D^T are LDS-resident.  Per step:
//   A  H P            (waves 0,1; K = 64)          hm = H m, v = y - hm                 (wave 2)
//   B  S = (H P) H^T + D R D^T                     (wave 3; K = 64)
//   C  chol(S + 1e-6) (wave 0, rows in registers, multipliers broadcast through LDS) with its inverse
//      L^-1 trailing it column block by column block on wave 1
//      chol(S), z = L^-1 v, log-likelihood         (wave 3)
//   E  W = L^-1 (H P),  F  X = L^-T W = (S + 1e-6)^-1 H P   (waves 2,3; K = 32)     K = X^T
//   G  K S = X^T S    (waves 2,3; K = 32)          m+ = m + X^T v                       (wave 1)
//   H  P+ = P - (K S) X                            (all waves; K = 32)
//   I  A P+           (all waves; K = 64)          m- = A m+ + G q0                 (waves 0,3)
//   J  P- = (A P+) A^T + G Q G^T                   (all waves; K = 64)
// The m x m system is solved through a Cholesky factor instead of the reference's LU with
// partial pivoting (utils.py:256-259): S + 1e-6 is symmetric positive definite, the solution is
// the same linear system's, and the two factorizations agree to ~1e-6 relative for the
// conditioned S of a filter (the parity budget is 1e-5); the log-likelihood uses the Cholesky
// factor of the un-jittered S exactly like the reference (inference.py:104, :24).
#include <cstdlib>
#include <cstring>
#include <vector>
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "lane_group.hpp"
#include "scan_common.hpp"

namespace bf {

static int mfma_variant_default() {  // BAYESFILT_MFMA_VARIANT=1..5 overrides the default for A/B runs of unmodified programs
  const char* e = std::getenv("BAYESFILT_MFMA_VARIANT");
  const int v = e ? std::atoi(e) : 5;
  return (v >= 1 && v <= 5) ? v : 5;
}
Option g_kf_mfma_variant{mfma_variant_default(), OPT_KF_MFMA_VARIANT};  // bf_set_option "kf_mfma_variant": 5 = products as three-term bf16 splits on the bf16 matrix pipe (default); 2 = fp32 MFMAs, gain-free update, factorization in VALU registers; 3 = factorization by rank-2 MFMAs; 4 = variant 2 at three workgroups per CU; 1 = round 1's kernel

using f32x16 = __attribute__((ext_vector_type(16))) float;
using lds_f = __attribute__((address_space(3))) float;
using lds_i = __attribute__((address_space(3))) int;
using v4f = __attribute__((ext_vector_type(4))) float;
using lds_f4 = const __attribute__((address_space(3))) v4f;

template <int N, int M>
struct MfmaConst {  // device-resident (too large for kernel arguments)
  float A[N * N], H[M * N], GQG[N * N], DRD[M * M], Gq0[N], Dr0[M];
  float dth[8];   // DYN != 0: the registry dynamics' scalars
  unsigned short A3[3][N * N], H3[3][M * N];  // variant 5: A = A3[0] + A3[1] + A3[2] exactly, three bf16 terms (row-major)
};

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of accumulator register r inside a 32x32 tile (C/D layout of the 32x32 MFMA shapes)
__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// A zero the compiler cannot see through, produced inside the time loop: LDS addresses formed from it
// are loop-variant, so they stay "register + immediate offset" operands instead of being hoisted out
// of the loop as hundreds of loop-invariant address registers (which then spill).
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

__device__ __forceinline__ void wave_lds_order() {  // order one wave's LDS traffic (the hardware runs it in issue order)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// In-wave Cholesky of a 32x32 SPD matrix, right-looking: lane (l & 31) holds row l of the matrix in
// a[0..31]; on return a[k] (k < row) holds L[row][k], lane k of rdv holds 1 / L[k][k], and the LDS
// block Lc holds the factor by columns, Lc[32 j + i] = L[i][j] for i > j with 1 / L[j][j] on the
// diagonal.  Column j is scaled by 1 / sqrt(A[j][j]) and the trailing rows take the outer product
// off at once (31 - j independent updates).  The multipliers L[k][j], k > j, are the same for every
// lane: the column goes to LDS and comes back as broadcast ds_read_b128s -- a v_readlane per
// multiplier (SGPR write + wait states + one VALU slot each) was 3x slower on the serial path.
__device__ __forceinline__ float rsqrt_newton(float d) {
  // 1 / sqrt(d): v_rsq_f32 plus one Newton step (~1 ulp; the raw approximation alone costs the 1e-5
  // parity budget over 32 columns, the IEEE sqrt + division sequences are ~35 dependent instructions
  // per column on the serial path).  NaN for d < 0 (matrix not positive definite).
  const float y0 = __builtin_amdgcn_rsqf(d);
  const float e0 = fmaf(-(d * y0), y0, 1.0f);
  return fmaf(0.5f * y0, e0, y0);
}

#ifndef BF_MFMA_POLL_SLEEP
#define BF_MFMA_POLL_SLEEP 1  // x 64 cycles between polls of the progress counter
#endif
#ifndef BF_MFMA_PUB
#define BF_MFMA_PUB 4  // columns per progress publication of the factorizing wave
#endif

// Software pipelining: the LDS round trip of column j + 1 (its write, and the broadcast reads of its
// multipliers) is issued BEFORE the trailing update of column j: the look-ahead already brings a[j + 1] up to date, so
// column j + 1 can be scaled and published one iteration early and its multipliers arrive while the 30 - j
// multiply-adds of column j issue.  Two multiplier buffers alternate by column parity.
template <bool PUBLISH, bool INVERT, class LP>
__device__ __forceinline__ void chol32_rows_pipe(float* a, float& rdv, int li, LP Lc, lds_i* progress, float* x) {
  float mult[2][32];
  auto load_mult = [&](auto C, float* dst) {  // multipliers L[k][c], k >= c + 2, of column c from LDS
    constexpr int c = decltype(C)::value;
    constexpr int kq = (c + 2 + 3) / 4 * 4;
    static_for<c + 2, (kq < 32 ? kq : 32)>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      dst[k] = Lc[32 * c + k];
    });
    static_for<kq / 4, 8>([&](auto Qd) {
      constexpr int q = decltype(Qd)::value;
      const v4f v = *reinterpret_cast<lds_f4*>(Lc + 32 * c + 4 * q);
      dst[4 * q + 0] = v.x;
      dst[4 * q + 1] = v.y;
      dst[4 * q + 2] = v.z;
      dst[4 * q + 3] = v.w;
    });
  };
  float rinv = rsqrt_newton(rdlane(a[0], 0));  // 1 / L[j][j] of the column whose trailing update is running
  float lj = a[0] * rinv;
  rdv = (li == 0) ? rinv : rdv;
  a[0] = lj;
  Lc[li] = (li == 0) ? rinv : lj;
  float l1 = rdlane(lj, 1);                     // L[j + 1][j]
  a[1] = fmaf(-lj, l1, a[1]);
  float rinv_n = rsqrt_newton(rdlane(a[1], 1));
  wave_lds_order();
  load_mult(std::integral_constant<int, 0>{}, mult[0]);
  static_for<0, 31>([&](auto J) {
    constexpr int j = decltype(J)::value;
    float* cur = mult[j & 1];
    float* nxt = mult[(j + 1) & 1];
    // column j + 1: scale, publish
    const float lj_n = a[j + 1] * rinv_n;
    rdv = (li == j + 1) ? rinv_n : rdv;
    a[j + 1] = lj_n;
    Lc[32 * (j + 1) + li] = (li == j + 1) ? rinv_n : lj_n;
    wave_lds_order();
    if constexpr (PUBLISH && (j + 1) % BF_MFMA_PUB == BF_MFMA_PUB - 1) *progress = j + 2;
    if constexpr (j < 30) load_mult(std::integral_constant<int, j + 1>{}, nxt);
    // trailing update of column j with the multipliers loaded one iteration ago
    static_for<j + 2, 32>([&](auto Kk) {
      constexpr int k = decltype(Kk)::value;
      a[k] = fmaf(-lj, cur[k], a[k]);
    });
    if constexpr (INVERT) {
      // forward substitution for L^-1 on the same multipliers (lane = column of the inverse, x[r] = its row r):
      // step j needs column j of L only, which is exactly what this iteration holds -- no second wave, no second
      // set of LDS reads, and its multiply-adds fill the latency bubbles of the factorization chain
      const float xj = x[j] * rinv;
      x[j] = xj;
      x[j + 1] = fmaf(-l1, xj, x[j + 1]);
      static_for<j + 2, 32>([&](auto R) {
        constexpr int r = decltype(R)::value;
        x[r] = fmaf(-cur[r], xj, x[r]);
        asm volatile("" : "+v"(x[r]));  // keep the update here (see invert_following)
      });
    }
    float l1_n = 0.f;
    if constexpr (j < 30) {  // look-ahead: row j + 2 of column j + 1, next reciprocal square root
      l1_n = rdlane(lj_n, j + 2);
      a[j + 2] = fmaf(-lj_n, l1_n, a[j + 2]);
    }
    rinv = rinv_n;
    if constexpr (j < 30) rinv_n = rsqrt_newton(rdlane(a[j + 2], j + 2));
    l1 = l1_n;
    lj = lj_n;
  });
  if constexpr (INVERT) x[31] *= rinv;  // rinv is 1 / L[31][31] after the last iteration
}

// Phase C of the scan as out-of-line functions: everything they touch lives in LDS, so the call costs
// a few scalar moves, and the factorization gets a register allocation of its own (inlined into the
// 10-phase kernel body the compiler hoists address arithmetic across the time loop and spills).
//
// chol(S + 1e-6), published four columns at a time: Lc[32 j + i] = L[i][j] (1 / L[j][j] on the
// diagonal) and *progress = number of finished columns
__device__ __attribute__((noinline)) void factor_publish(lds_f* sS, lds_f* Lc, lds_i* progress, int lane) {
  if (lane >= 32) return;  // rows live in lanes 0..31: the upper half would only double the LDS return traffic
  constexpr int PS = 33;
  const int lr = lane & 31;
  float a[32];
  BF_UNROLL for (int k = 0; k < 32; ++k) a[k] = sS[lr * PS + k] + 1e-6f;  // psd_solve's jitter on every entry
  float rdv = 0.f;
  chol32_rows_pipe<true, false>(a, rdv, lr, Lc, progress, nullptr);
}

// chol(S + 1e-6) and its inverse in ONE wave: sLi[i][c] = (L^-1)[i][c] (lane (l & 31) holds column c).  The factor
// columns still pass through Lc (the broadcast of the multipliers), nothing polls.  Measured (-DBF_MFMA_FUSED_INVERSE):
// phase C drops from 9.3 to 6.7 us, yet the scan is 7 % SLOWER (2.98e7 against 3.20e7 steps/s): the two workgroups of
// a CU alternate -- one factorizes while the other runs its MFMA phases -- and a shorter phase C only makes their MFMA
// phases collide.  Kept as the starting point for a design with a third workgroup per CU.
__device__ __attribute__((noinline)) void factor_invert(lds_f* sS, lds_f* Lc, lds_f* sLi, int lane) {
  constexpr int PS = 33;
  const int lr = lane & 31;
  float a[32], x[32];
  BF_UNROLL for (int k = 0; k < 32; ++k) a[k] = sS[lr * PS + k] + 1e-6f;  // psd_solve's jitter on every entry
  BF_UNROLL for (int i = 0; i < 32; ++i) x[i] = (lr == i) ? 1.f : 0.f;
  float rdv = 0.f;
  chol32_rows_pipe<false, true>(a, rdv, lr, Lc, nullptr, x);
  if (lane < 32) BF_UNROLL for (int i = 0; i < 32; ++i) sLi[i * PS + lr] = x[i];
}

// The inverse of that factor, computed by another wave while the factorization is still running:
// step i of the forward substitution needs column i of L only, so this wave trails the factorizing
// one by a block of columns (it polls *progress) and the two serial chains overlap instead of adding
// up.  sLi[i][c] = (L^-1)[i][c]; lane (l & 31) holds column c = l & 31.
__device__ __attribute__((noinline)) void invert_following(lds_f* Lc, lds_i* progress, lds_f* sLi, int lane) {
  if (lane >= 32) return;  // rows live in lanes 0..31: the upper half would only double the LDS return traffic
  constexpr int PS = 33;
  const int lr = lane & 31;
  float x[32];
  BF_UNROLL for (int i = 0; i < 32; ++i) x[i] = (lr == i) ? 1.f : 0.f;
  // column i of the factor (reciprocal diagonal + the multipliers below it) is loaded while column i - 1 is applied:
  // two buffers by column parity, fenced per column so that the loads of a whole block are not hoisted together
  // (they were: 248 VGPRs and spills to scratch on the critical path)
  float mult[2][32], dg[2];
  auto wait_for = [&](int cols) {  // wave-uniform; the producer always reaches 32
    int seen = *(volatile lds_i*)progress;
    while (seen < cols) {
      __builtin_amdgcn_s_sleep(BF_MFMA_POLL_SLEEP);  // a tight poll floods the LDS queue the factorizing waves live on
      seen = *(volatile lds_i*)progress;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  // LDS only: a full fence would also drain the wave's output stores
  };
  auto load_col = [&](auto C, float* dst, float& d) {
    constexpr int c = decltype(C)::value;
    d = Lc[32 * c + c];  // 1 / L[c][c]
    constexpr int rq = (c + 1 + 3) / 4 * 4;
    static_for<c + 1, (rq < 32 ? rq : 32)>([&](auto R) {
      constexpr int r = decltype(R)::value;
      dst[r] = Lc[32 * c + r];
    });
    static_for<rq / 4, 8>([&](auto Qd) {
      constexpr int q = decltype(Qd)::value;
      const v4f v = *reinterpret_cast<lds_f4*>(Lc + 32 * c + 4 * q);
      dst[4 * q + 0] = v.x;
      dst[4 * q + 1] = v.y;
      dst[4 * q + 2] = v.z;
      dst[4 * q + 3] = v.w;
    });
  };
  wait_for(BF_MFMA_PUB);
  load_col(std::integral_constant<int, 0>{}, mult[0], dg[0]);
  static_for<0, 32>([&](auto I) {
    constexpr int i = decltype(I)::value;
    if constexpr (i < 31) {
      if constexpr ((i + 1) % BF_MFMA_PUB == 0) wait_for(i + 1 + BF_MFMA_PUB);
      load_col(std::integral_constant<int, i + 1>{}, mult[(i + 1) & 1], dg[(i + 1) & 1]);
    }
    const float* cur = mult[i & 1];
    x[i] *= dg[i & 1];
    static_for<i + 1, 32>([&](auto R) {
      constexpr int r = decltype(R)::value;
      x[r] = fmaf(-cur[r], x[i], x[r]);
      // pin the update here: without it every multiply-add sinks below the loads of ALL later columns (the
      // loads have no ordering against pure arithmetic), which is what cost 248 VGPRs and the spills
      asm volatile("" : "+v"(x[r]) : : "memory");
    });
  });
  if (lane < 32) BF_UNROLL for (int i = 0; i < 32; ++i) sLi[i * PS + lr] = x[i];
}

// chol(S) (no jitter), z = L^-1 v, log N(v; 0, S) -- inference.py:104, :24
__device__ __attribute__((noinline)) float factor_loglik(lds_f* sS, lds_f* Lc, lds_f* sv, int lane) {
  if (lane >= 32) return 0.f;  // rows live in lanes 0..31 (the caller reads the result in lane 0)
  constexpr int PS = 33;
  const int lr = lane & 31;
  float a[32];
  BF_UNROLL for (int k = 0; k < 32; ++k) a[k] = sS[lr * PS + k];
  float rdv = 0.f;
  chol32_rows_pipe<false, false>(a, rdv, lr, Lc, nullptr, nullptr);
  // z = L^-1 v by forward substitution across lanes; lane i carries the running residual of row i
  float acc = sv[lr], quad = 0.f, dprod = 1.f;
  static_for<0, 32>([&](auto Kk) {
    constexpr int k = decltype(Kk)::value;
    const float zk = rdlane(acc, k) * rdlane(rdv, k);
    quad = fmaf(zk, zk, quad);
    dprod *= rdlane(a[k], k);
    acc = fmaf(-a[k], zk, acc);
  });
  // sum of 32 log-diagonals as log of the product (32 factors of O(1) stay in range)
  return -0.5f * quad - 0.5f * 32.0f * 1.8378770664093453f - fast_log(dprod);
}

template <int N, int M>
__global__ void __launch_bounds__(256, 2)
kf_scan_mfma_kernel(const MfmaConst<N, M>* __restrict__ cst, CView y, CarryView carry, OutViews out, long long B, long long T) {
  static_assert(N == 64 && M == 32, "tile assignment is written for n = 64, m = 32");
  constexpr int PP = N + 1;  // LDS pitch of 64-wide matrices
  constexpr int PS = M + 1;  // LDS pitch of 32-wide matrices
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // Roles are assigned to a rotated wave index: the wave that runs the serial factorization (role 0)
  // then sits on a different SIMD in neighbouring workgroups, so the two workgroups a CU holds do
  // not queue their scalar-heavy phases on the same SIMD.
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + (int)blockIdx.x) & 3);
  const int ti = wave >> 1, tj = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;
  const long long b = blockIdx.x;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sP = lds;                 // [64][65]  current covariance
  float* sT = sP + N * PP;         // [64][65]  H P / W / X, then A P+
  float* sKS = sT + N * PP;        // [64][33]  -(K S)
  float* sS = sKS + N * PS;        // [32][33]  S
  float* sLi = sS + M * PS;        // [32][33]  inverse Cholesky factor of S + 1e-6
  float* sm = sLi + M * PS;        // [64] mean
  float* sm2 = sm + N;             // [64] mean (ping-pong)
  float* sv = sm2 + N;             // [32] innovation
  float* sy = sv + M;              // [32] observation
  int* sflag = reinterpret_cast<int*>(sy + M);  // [4] columns of chol(S + 1e-6) published so far
  float* sH = sy + M + 4;          // [32][65]  H (operand source; keeping it in VGPRs spills the factorization)
  float* sD = sH + M * PP;         // [32][33]  D R D^T
  float* sA = sD + M * PS;         // [64][65]  A

  // ---- constant operands: A, H, D R D^T in LDS (the MFMA operand pattern X[32*blk + (l & 31)][2 s + (l >> 5)]
  // is conflict-free with the odd pitch); keeping A in VGPRs spills the factorization
  for (int e = tid; e < N * N; e += 256) sA[(e / N) * PP + (e % N)] = cst->A[e];
  for (int e = tid; e < M * N; e += 256) sH[(e / N) * PP + (e % N)] = cst->H[e];
  for (int e = tid; e < M * M; e += 256) sD[(e / M) * PS + (e % M)] = cst->DRD[e];
  // the wave's tile of P lives in LDS between phases (and in the accumulators inside H and J):
  // nothing but scalars is live in registers across the factorization phase
  BF_UNROLL for (int r = 0; r < 16; ++r)
    sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = carry.P_in[b * N * N + (32 * ti + c_row(r, lane)) * N + 32 * tj + lr];
  if (tid < N) sm[tid] = carry.m_in[b * N + tid];
  if (tid == 0) sflag[0] = 0;
  float w = carry.w_in ? carry.w_in[b] : 1.0f;
  float ynext = (wave == 2 && lane < M) ? y.p[b * y.sB + lane * y.sE] : 0.f;
  __syncthreads();

#ifdef BF_MFMA_PHASE_TIMERS  // debug build: per-phase wall-clock ticks of workgroup 0 (scripts/mfma_phase_probe.py)
  long long tacc[10] = {0};
  long long tprev = wall_clock64();
#define BF_TICK(i) { const long long tn_ = wall_clock64(); tacc[i] += tn_ - tprev; tprev = tn_; }
#else
#define BF_TICK(i)
#endif
  float* mcur = sm;
  float* mnxt = sm2;
  for (long long t = 0; t < T; ++t) {
    // ================= phase A: H P (waves 0,1); innovation (wave 2)
    if (wave < 2) {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 32; ++s) acc = mfma2(sH[lr * PP + 2 * s + lk], sP[(2 * s + lk) * PP + 32 * tj + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[c_row(r, lane) * PP + 32 * tj + lr] = acc[r];
    } else if (wave == 2) {
      if (lane < M) sy[lane] = ynext;
      const long long tn = t + 1 < T ? t + 1 : t;
      if (lane < M) ynext = y.p[b * y.sB + tn * y.sT + lane * y.sE];  // prefetch
      float s = 0.f;
      BF_UNROLL for (int q = 0; q < 32; ++q) s = fmaf(sH[lr * PP + 2 * q + lk], mcur[2 * q + lk], s);
      s += __shfl_xor(s, 32, 64);
      if (lane < M) sv[lane] = sy[lane] - (s + cst->Dr0[lane]);
    }
    BF_TICK(0)
    lds_barrier();
    // ================= phase B: S = (H P) H^T + D R D^T (wave 3)
    if (wave == 3) {
      f32x16 acc;
      BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = sD[c_row(r, lane) * PS + lr];
      BF_UNROLL for (int s = 0; s < 32; ++s) acc = mfma2(sT[lr * PP + 2 * s + lk], sH[lr * PP + 2 * s + lk], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sS[c_row(r, lane) * PS + lr] = acc[r];
    }
    BF_TICK(1)
    lds_barrier();
    // ================= phase C: factorizations (waves 0 and 3)
    float ll = 0.f;
    if (wave == 0) {
      // -(K S) is dead between phases H and G: scratch for the factor columns of both factorizations
#ifndef BF_MFMA_FUSED_INVERSE
      factor_publish((lds_f*)sS, (lds_f*)sKS, (lds_i*)sflag, lane);
    } else if (wave == 1) {
      invert_following((lds_f*)sKS, (lds_i*)sflag, (lds_f*)sLi, lane);
#else
      factor_invert((lds_f*)sS, (lds_f*)sKS, (lds_f*)sLi, lane);
#endif
    } else if (wave == 3) {
      ll = factor_loglik((lds_f*)sS, (lds_f*)(sKS + 1024), (lds_f*)sv, lane);
    }
    BF_TICK(2)
    lds_barrier();
    // ================= phase E: W = L^-1 (H P) -> sT rows 32..63 (waves 2,3; K = 32)
    if (wave >= 2) {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 16; ++s)
          acc = mfma2(sLi[lr * PS + 2 * s + lk], sT[(2 * s + lk) * PP + 32 * tj + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[(32 + c_row(r, lane)) * PP + 32 * tj + lr] = acc[r];
    }
    BF_TICK(3)
    lds_barrier();
    // ================= phase F: X = L^-T W -> sT rows 0..31 (waves 2,3)
    if (wave >= 2) {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 16; ++s)
          acc = mfma2(sLi[(2 * s + lk) * PS + lr], sT[(32 + 2 * s + lk) * PP + 32 * tj + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[c_row(r, lane) * PP + 32 * tj + lr] = acc[r];
    }
    BF_TICK(4)
    lds_barrier();
    // ================= phase G: -(K S) = -(X^T S), row block tj (waves 2,3); m+ (wave 1)
    if (wave >= 2) {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 16; ++s)
          acc = mfma2(sT[(2 * s + lk) * PP + 32 * tj + lr], sS[(2 * s + lk) * PS + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sKS[(32 * tj + c_row(r, lane)) * PS + lr] = -acc[r];
    } else if (wave == 1) {
      float s = mcur[lane];
      BF_UNROLL for (int a = 0; a < M; ++a) s = fmaf(sT[a * PP + lane], sv[a], s);
      mnxt[lane] = s;  // filtered mean
    }
    BF_TICK(5)
    lds_barrier();
    // ================= phase H: P+ = P - (K S) X (all waves; K = 32); emit filtered streams
    f32x16 Pacc;
    BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr];
    BF_UNROLL for (int s = 0; s < 16; ++s)
        Pacc = mfma2(sKS[(32 * ti + lr) * PS + 2 * s + lk], sT[(2 * s + lk) * PP + 32 * tj + lr], Pacc);
    BF_UNROLL for (int r = 0; r < 16; ++r) sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = Pacc[r];
    if (out.P.p) BF_UNROLL for (int r = 0; r < 16; ++r)
        out.P.p[b * out.P.sB + t * out.P.sT + ((32 * ti + c_row(r, lane)) * N + 32 * tj + lr) * out.P.sE] = Pacc[r];
    if (wave == 2 && out.m.p) out.m.p[b * out.m.sB + t * out.m.sT + lane * out.m.sE] = mnxt[lane];
    if (wave == 0 && lane == 0) sflag[0] = 0;  // re-armed two barriers before the next factorization
    if (wave == 3 && lane == 0) {
      w = reweight_single(ll, w);
      if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
      if (out.ll.p) out.ll.p[b * out.ll.sB + t * out.ll.sT] = ll;
    }
    BF_TICK(6)
    lds_barrier();
    // ================= phase I: A P+ -> sT (all waves; K = 64); m- = A m+ + G q0 (waves 0, 3)
    {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 32; ++s) acc = mfma2(sA[(32 * ti + lr) * PP + 2 * s + lk], sP[(2 * s + lk) * PP + 32 * tj + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = acc[r];
    }
    if (wave == 0 || wave == 3) {
      float s = 0.f;
      BF_UNROLL for (int q = 0; q < 32; ++q) s = fmaf(sA[(32 * ti + lr) * PP + 2 * q + lk], mnxt[2 * q + lk], s);
      s += __shfl_xor(s, 32, 64);
      if (lane < 32) mcur[32 * ti + lane] = s + cst->Gq0[32 * ti + lane];  // predicted mean
    }
    BF_TICK(7)
    lds_barrier();
    // ================= phase J: P- = (A P+) A^T + G Q G^T (all waves; K = 64); emit predicted streams
    BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = cst->GQG[(32 * ti + c_row(r, lane)) * N + 32 * tj + lr];
    BF_UNROLL for (int s = 0; s < 32; ++s) Pacc = mfma2(sT[(32 * ti + lr) * PP + 2 * s + lk], sA[(32 * tj + lr) * PP + 2 * s + lk], Pacc);
    BF_UNROLL for (int r = 0; r < 16; ++r) sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = Pacc[r];
    if (out.pP.p) BF_UNROLL for (int r = 0; r < 16; ++r)
        out.pP.p[b * out.pP.sB + t * out.pP.sT + ((32 * ti + c_row(r, lane)) * N + 32 * tj + lr) * out.pP.sE] = Pacc[r];
    if (wave == 2 && out.pm.p) out.pm.p[b * out.pm.sB + t * out.pm.sT + lane * out.pm.sE] = mcur[lane];
    BF_TICK(8)
    lds_barrier();
  }

  if (carry.P_out) BF_UNROLL for (int r = 0; r < 16; ++r)
      carry.P_out[b * N * N + (32 * ti + c_row(r, lane)) * N + 32 * tj + lr] = sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr];
  if (carry.m_out && tid < N) carry.m_out[b * N + tid] = mcur[tid];
  if (carry.w_out && wave == 3 && lane == 0) carry.w_out[b] = w;
#ifdef BF_MFMA_PHASE_TIMERS
  __syncthreads();
  if (b == 0 && lane == 0 && carry.P_out) for (int i = 0; i < 10; ++i) carry.P_out[wave * 16 + i] = (float)tacc[i];
#endif
}


// =======================================================================================================================
// Measured on one MI355X, BASELINE configs[4] (B = 32 768, T = 2 000, all five streams in T-chunks of 100; bench.py
// --config kalman64), same box, steps/s:   variant 1 (round 1) 3.28e7 | 2 3.73e7 | 3 3.69e7 | 4 3.1e7 | 5 (default) 4.90e7.
// Variants 1-4 run the five products on v_mfma_f32_32x32x2_f32, which shares the SIMD's fp32 datapath with the vector
// instructions of the factorization (see variant 5's header and profiles/r02_f32_pipe_probe.txt); variant 5 moves them to
// the bf16 matrix pipe as three-term splits at fp32-level rounding.  Errors against the oracle over the 2 000 steps
// (scripts/mfma_parity_probe.py): variant 2 means 3.0e-6, covariances 1.2e-6, log-likelihood 3.8e-6; variant 5 3.6e-6,
// 1.5e-6, 4.6e-6 (budget 1e-5).
// What moved variants 2 and 3 from 3.54e7 / 3.07e7 (first cut of this round) to these numbers:
//   * no loop-invariant operand lives in registers across steps (per_step() below): the compiler had hoisted the sixteen
//     64-bit store addresses of each output stream and the loads of G Q G^T out of the time loop and spilled them
//     (62 / 87 VGPRs spilled -> 0 / 0);
//   * the serial phase's elimination steps are packed (v_pk_fma_f32 on (row entry, right-hand-side entry) pairs):
//     2 295 -> 1 668 vector instructions in chol_w_rows (a v_pk_fma_f32 costs 1.75 v_fma_f32, so the gain is in the
//     instruction count around them, not in the multiply-adds);
//   * roles placed by the hardware's wave placement rather than by blockIdx (+1 %, see the kernel).
// Tried and measured flat or worse: a half-step start offset for the second workgroup of a CU (0 %); s_setprio 3 around
// the factorization (-2 %); dropping the log-likelihood factorization altogether as an upper bound for deriving it from
// the jittered factor by the matrix determinant lemma (0 %: that wave is not on the critical path).
// Phase timers (scripts/mfma_phase_probe.py, us per step): alone on its CU a workgroup takes 9.9 (A 1.3, S 1.1,
// factorization + W 3.6, H 1.0, I 1.5, J 1.3); with a second workgroup on the CU 14.1 per workgroup, i.e. 7.0 per step and
// CU: the factorization stretches to 6.3 beside the other workgroup's phases although neither the vector nor the matrix
// pipe is more than half busy on average.  A third workgroup per CU (variant 4: A and D R D^T read from L2 per step
// to fit the LDS, 168 VGPRs) is slower, 7.5 us per step and CU: its global operand loads sit on the critical path.
//
// Variant 2 (default): the gain is never formed.  With L L^T = S + 1e-6 (every entry: the psd_solve jitter J = 1e-6 1 1^T),
// W = L^-1 (H P), g = L^-1 1 and z = L^-1 v:
//     K S K^T = X^T (S_j - J) X = W^T W - 1e-6 (W^T g)(W^T g)^T          (X = S_j^-1 H P = L^-T W)
//     K v     = X^T v = W^T z
// -- the same quantities as P - K S K^T and m + K (y - h(m)) of inference.py:102-103, to rounding.  That removes the
// explicit inverse, the X = L^-T W and K S products and four of the nine barriers, and the whole serial phase fits one
// wave's REGISTERS: lane r holds row r of S + 1e-6; by symmetry the multipliers L[k][j] of column j are lane j's own
// row entries, broadcast with v_readlane (no LDS round trip per column), and every lane c = 0..63 carries column c of
// H P through the forward substitution in the same loop, fed by the same broadcasts.  A second wave factorizes the
// un-jittered S the same way for the log-likelihood (inference.py:104).  Per step:
//   A  H P (waves 0,1; K = 64)                                  y, H m, v (wave 2)
//   B  S = (H P) H^T + D R D^T, computed by waves 2 AND 3 (each needs it in registers; different SIMDs)
//   C  wave 3: chol(S + 1e-6) fused with W, g, z; c = 1e-3 W^T g; m+ = m + W^T z     wave 2: chol(S), log-likelihood
//   H  P+ = P - W^T W + c c^T (all waves; K = 32 + 2)
//   I  A P+ (all waves; K = 64), m- = A m+ + G q0               J  P- = (A P+) A^T + G Q G^T
#ifndef BF_MFMA_RDB
#define BF_MFMA_RDB 8  // broadcasts issued ahead of their consumers
#endif
__device__ __forceinline__ float rdlane_u(float v, int l) {  // v_readlane_b32: lane l's value as a wave-uniform scalar
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// wave 3: S (acc layout in `sc`, [32][33]) -> rows; chol(S + 1e-6); W = L^-1 (H P) -> sT rows 32..63; c, m+
// The function is VALU-issue bound (a wave64 instruction occupies the SIMD for 4 cycles; ~2 300 of them were the 3.8 us of
// this phase), so entry k of the lane's row of S and entry k of its column of H P travel as ONE register pair and every
// elimination step is one v_pk_fma_f32 on (a[k], w[k]) with the broadcast multiplier as its scalar operand -- the same
// fmas in the same order as the unpacked form, half the instructions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using lds_u32x2 = __attribute__((address_space(3))) u32x2;
using lds_u32x4 = __attribute__((address_space(3))) u32x4;
using lds_c = __attribute__((address_space(3))) char;

// x0, x1 -> three packed pairs of bf16 (round to nearest even, v_cvt_pk_bf16_f32) with x = hi + mid + lo EXACTLY: the
// residual of a 24-bit significand after an 8-bit term has at most 16 bits, after two terms at most 8.
struct Split3 {
  unsigned hi, mid, lo;
};
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
__device__ __forceinline__ Split3 split_pair(float x0, float x1) {
  Split3 o;
  o.hi = pk_bf16(x0, x1);
  const float r0 = x0 - bf_lo(o.hi), r1 = x1 - bf_hi(o.hi);
  o.mid = pk_bf16(r0, r1);
  const float q0 = r0 - bf_lo(o.mid), q1 = r1 - bf_hi(o.mid);
  o.lo = pk_bf16(q0, q1);
  return o;
}

// BF = false: W (fp32) -> sT rows 32..63.  BF = true (variant 5): W^T as three bf16 terms, wt[p][lane][k], 80-byte rows.
// NCOL = 64: every lane carries its own column of H P (pitch 65); NCOL = 32 (the one-wave kernel for n <= 32): the upper
// half-wave repeats the lower one's columns (pitch 33), its stores land on the same addresses with the same values.
// LL: also returns log N(v; 0, S) for the UN-jittered S, from this factorization of S_j = S + eps 1 1^T (eps = 1e-6) by
// the matrix determinant lemma and Sherman-Morrison: with g = L^-1 1, z = L^-1 v (both carried through the loop anyway)
//   det S = det S_j (1 - eps g^T g),   v^T S^-1 v = z^T z + eps (g^T z)^2 / (1 - eps g^T g)
// -- exact identities, evaluated in fp32 (eps g^T g is O(1e-4) for a conditioned S): the separate factorization of S that
// inference.py:104 implies (a second 1 200-instruction serial chain) is not needed.
// DUAL (NCOL = 32 only): TWO chains per wave, chain c in the half-wave of lanes 32 c .. 32 c + 31 -- the single-chain form lets the
// upper half repeat the lower one's work -- with chain 1's arrays `dual_stride` bytes behind chain 0's.  The column broadcasts
// then differ between the halves, so they cannot be v_readlane scalars: ds_swizzle (BitMode and = 0, or = j: lane j of each
// group of 32, through the LDS crossbar without touching memory) delivers them in a vector register.
template <int J>
__device__ __forceinline__ float bcast_half(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (J & 31) << 5));
}
template <bool BF, int NCOL = 64, bool LL = false, bool DUAL = false>
__device__ __forceinline__ float chol_w_rows_impl(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv,
                                                  lds_c* wt, int lane_in, int dual_stride = 0) {
  static_assert(!DUAL || NCOL == 32, "two chains per wave: 32 columns each");
  constexpr int PP = NCOL + 1, PS = 33, WT_TERM_B = NCOL * 80;
  const int r = lane_in & 31;
  const int lane = NCOL == 64 ? lane_in : r;
  if constexpr (DUAL) {
    const int off = (lane_in >> 5) * dual_stride;
    sc = (lds_f*)((lds_c*)sc + off); sT = (lds_f*)((lds_c*)sT + off); sv = (lds_f*)((lds_c*)sv + off);
    mcur = (lds_f*)((lds_c*)mcur + off); mnxt = (lds_f*)((lds_c*)mnxt + off); scv = (lds_f*)((lds_c*)scv + off);
    wt = wt + off;
  }
  auto bc = [&](float v, auto J) __attribute__((always_inline)) {
    if constexpr (DUAL) return bcast_half<decltype(J)::value>(v);
    else return rdlane_u(v, decltype(J)::value);
  };
  f32x2 aw[32];  // .x: row r of S + 1e-6 (psd_solve's jitter on every entry, utils.py:258); .y: column `lane` of H P
  BF_UNROLL for (int k = 0; k < 32; ++k) aw[k] = f32x2{sc[r * PS + k] + 1e-6f, sT[k * PP + lane]};
  f32x2 rgz = f32x2{1.0f, sv[r]};  // residuals of g = L^-1 1, z = L^-1 v (row r)
  f32x2 acc_cm = f32x2{0.f, 0.f};  // (W^T g)[lane], (W^T z)[lane]
  float s_gg = 0.f, s_gz = 0.f, s_zz = 0.f, rprod = 1.f;   // LL: g^T g, g^T z, z^T z, prod 1 / L_jj (wave-uniform)
  Split3 wsp[4];
  static_for<0, 32>([&](auto J) {
    constexpr int j = decltype(J)::value;
    const float rinv = rsqrt_newton(bc(aw[j].x, J));                  // 1 / L[j][j], wave-uniform (DUAL: per half-wave)
    const f32x2 lw = aw[j] * rinv;                                     // L[r][j] (meaningful for r >= j), W[j][lane] (final)
    const f32x2 gz = f32x2{bc(rgz.x, J), bc(rgz.y, J)} * rinv;         // g[j], z[j]: wave-uniform
    aw[j] = lw;
    rgz = __builtin_elementwise_fma(f32x2{-lw.x, -lw.x}, gz, rgz);
    acc_cm = __builtin_elementwise_fma(f32x2{lw.y, lw.y}, gz, acc_cm);
    if constexpr (LL) {
      s_gg = fmaf(gz.x, gz.x, s_gg);
      s_gz = fmaf(gz.x, gz.y, s_gz);
      s_zz = fmaf(gz.y, gz.y, s_zz);
      rprod *= rinv;
    }
    const f32x2 ntq = -(lw * rinv);                                    // -a[r][j] / d_j, -w[j] / d_j
    // L[k][j] sqrt(d_j) = a[k][j] = a[j][k] by symmetry: lane j's own entries, read BEFORE this step updates them.
    // The broadcasts go out in batches of BF_MFMA_RDB ahead of the multiply-adds that consume them: a v_readlane's
    // scalar result takes several issue slots to become readable, and back-to-back (readlane, fma) pairs stall on it.
    static_for<0, (31 - j + BF_MFMA_RDB - 1) / BF_MFMA_RDB>([&](auto Cb) {
      constexpr int k0 = j + 1 + decltype(Cb)::value * BF_MFMA_RDB;
      constexpr int nk = (32 - k0) < BF_MFMA_RDB ? (32 - k0) : BF_MFMA_RDB;
      float sb[BF_MFMA_RDB];
      static_for<0, nk>([&](auto I) { sb[decltype(I)::value] = bc(aw[k0 + decltype(I)::value].x, J); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, nk>([&](auto I) {
        constexpr int k = k0 + decltype(I)::value;
        aw[k] = __builtin_elementwise_fma(ntq, f32x2{sb[decltype(I)::value], sb[decltype(I)::value]}, aw[k]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    // variant 5: rows j - 1, j of W are final; their bf16 terms are formed here, in the issue gaps of the dependent
    // chain (the loop runs at ~55 % of the issue rate), and go out eight rows per 16-byte store
    if constexpr (BF && (j & 1)) {
      wsp[(j >> 1) & 3] = split_pair(aw[j - 1].y, aw[j].y);
      if constexpr ((j & 7) == 7) {
        constexpr int q = j >> 3;
        *reinterpret_cast<lds_u32x4*>(wt + 0 * WT_TERM_B + lane * 80 + q * 16) = u32x4{wsp[0].hi, wsp[1].hi, wsp[2].hi, wsp[3].hi};
        *reinterpret_cast<lds_u32x4*>(wt + 1 * WT_TERM_B + lane * 80 + q * 16) = u32x4{wsp[0].mid, wsp[1].mid, wsp[2].mid, wsp[3].mid};
        *reinterpret_cast<lds_u32x4*>(wt + 2 * WT_TERM_B + lane * 80 + q * 16) = u32x4{wsp[0].lo, wsp[1].lo, wsp[2].lo, wsp[3].lo};
      }
    }
  });
  if constexpr (!BF) {
    BF_UNROLL for (int i = 0; i < 32; ++i) sT[(32 + i) * PP + lane] = aw[i].y;
  }
  scv[lane] = acc_cm.x * 1e-3f;                            // sqrt(1e-6) (W^T g): enters P+ as + c c^T
  mnxt[lane] = mcur[lane] + acc_cm.y;                      // filtered mean
  if constexpr (LL) {
    const float one_m = fmaf(-1e-6f, s_gg, 1.0f);                                   // 1 - eps g^T g
    const float quad = s_zz + (1e-6f * s_gz) * s_gz / one_m;
    return -0.5f * quad - 0.5f * 32.0f * 1.8378770664093453f + fast_log(rprod) - 0.5f * fast_log(one_m);
  } else {
    return 0.f;
  }
}
#ifndef BF_V5_INLINE
#define BF_V5_INLINE 0
#endif
#ifndef BF_V5_LL_LEMMA
#define BF_V5_LL_LEMMA 0   // 1: log-likelihood from the jittered factorization (chol_w_rows_impl<.., LL = true>), wave 2 idle in phases B + C.
                           // Measured -7 % (4.23e7 against 4.56e7 on one box): the extra sums lengthen the critical wave's chain, the
                           // factorization they replace ran beside it.  (The one-wave kernel, where both chains are serial, gains 33 %.)
#endif
#ifndef BF_V5_HOP_RELOAD
#define BF_V5_HOP_RELOAD 0
#endif
__device__ __attribute__((noinline)) void chol_w_rows_bf(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv, lds_c* wt,
                                                         int lane) {
  chol_w_rows_impl<true>(sc, sT, sv, mcur, mnxt, scv, wt, lane);
}
__device__ __attribute__((noinline)) float chol_w_rows_bf_ll(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv, lds_c* wt,
                                                             int lane) {
  return chol_w_rows_impl<true, 64, true>(sc, sT, sv, mcur, mnxt, scv, wt, lane);
}
__device__ __attribute__((noinline)) float chol_w_rows_bf32(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv, lds_c* wt,
                                                            int lane) {
  return chol_w_rows_impl<true, 32, true>(sc, sT, sv, mcur, mnxt, scv, wt, lane);
}
// two chains per wave (kf_scan_bf32x2_kernel): pointers of chain 0, chain 1's arrays `stride` bytes behind; lanes of half c
// return chain c's log-likelihood
__device__ __attribute__((noinline)) float chol_w_rows_bf32x2(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv, lds_c* wt,
                                                              int lane, int stride) {
  return chol_w_rows_impl<true, 32, true, true>(sc, sT, sv, mcur, mnxt, scv, wt, lane, stride);
}
// out of line for variants 2 / 4 (a register allocation of its own); variant 5 inlines the body (it holds 112 operand
// registers across the factorization, which a call would spill and reload)
__device__ __attribute__((noinline)) void chol_w_rows(lds_f* sc, lds_f* sT, lds_f* sv, lds_f* mcur, lds_f* mnxt, lds_f* scv, int lane) {
  chol_w_rows_impl<false>(sc, sT, sv, mcur, mnxt, scv, nullptr, lane);
}

// wave 2: chol(S) (no jitter), z = L^-1 v, log N(v; 0, S) -- inference.py:104, :24
// Packed like chol_w_rows, here two neighbouring entries of the row per register pair and two broadcasts per scalar pair.
__device__ __forceinline__ float chol_loglik_rows_impl(lds_f* sc, lds_f* sv, int lane) {
  constexpr int PS = 33;
  const int r = lane & 31;
  f32x2 ap[16];  // (a[2 i], a[2 i + 1]) of row r
  BF_UNROLL for (int i = 0; i < 16; ++i) ap[i] = f32x2{sc[r * PS + 2 * i], sc[r * PS + 2 * i + 1]};
  float rz = sv[r], quad = 0.f, rprod = 1.f;
  static_for<0, 32>([&](auto J) {
    constexpr int j = decltype(J)::value;
    const float ajj = (j & 1) ? ap[j / 2].y : ap[j / 2].x;
    const float rinv = rsqrt_newton(rdlane_u(ajj, j));
    const float lj = ajj * rinv;
    const float zj = rdlane_u(rz, j) * rinv;
    rz = fmaf(-lj, zj, rz);
    quad = fmaf(zj, zj, quad);
    rprod *= rinv;
    const float nt = -(lj * rinv);
    // pairs i >= (j + 1) / 2; for even j the first pair is (a[j], a[j + 1]) and its .x -- the finished column entry,
    // never read again -- is updated along with the live .y
    constexpr int i_first = (j + 1) / 2;
    static_for<0, (16 - i_first + BF_MFMA_RDB / 2 - 1) / (BF_MFMA_RDB / 2)>([&](auto Cb) {
      constexpr int i0 = i_first + decltype(Cb)::value * (BF_MFMA_RDB / 2);
      constexpr int ni = (16 - i0) < BF_MFMA_RDB / 2 ? (16 - i0) : BF_MFMA_RDB / 2;
      f32x2 sb[BF_MFMA_RDB / 2];
      static_for<0, ni>([&](auto I) {
        constexpr int i = i0 + decltype(I)::value;
        sb[decltype(I)::value] = f32x2{rdlane_u(ap[i].x, j), rdlane_u(ap[i].y, j)};
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, ni>([&](auto I) {
        constexpr int i = i0 + decltype(I)::value;
        ap[i] = __builtin_elementwise_fma(f32x2{nt, nt}, sb[decltype(I)::value], ap[i]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  // -sum log L_jj = log prod (1 / L_jj) (32 factors of O(1) stay in range)
  return -0.5f * quad - 0.5f * 32.0f * 1.8378770664093453f + fast_log(rprod);
}

__device__ __attribute__((noinline)) float chol_loglik_rows(lds_f* sc, lds_f* sv, int lane) { return chol_loglik_rows_impl(sc, sv, lane); }

// ---- Variant 3: the factorization itself on the matrix cores.  In the accumulator layout of the 32x32 MFMA shapes a
// lane holds 16 entries of ONE column of S; S is symmetric, so register r_j of the 32 lanes of half h_j (j = r & 3 +
// 8 (r >> 2) + 4 h) is the whole pivot row j = the whole column j.  One right-looking elimination step -- S -= l l^T
// with l = column j / sqrt(d_j) -- is then ONE v_mfma_f32_32x32x2_f32 whose two operands are that register (scaled,
// masked to its half): no broadcast of multipliers at all, one v_readlane per column for the pivot.  The same step
// applied to the right-hand sides H P (kept in the accumulators of the waves that produced them) is the forward
// substitution W = L^-1 (H P): another MFMA per 32x32 tile, its operands the published columns of L (LDS, 128 bytes
// per column) and the tile's own pivot rows.  A few dependent MFMAs replace ~2 300 VALU / v_readlane instructions.
#ifndef BF_MFMA_PUB3
#define BF_MFMA_PUB3 4  // columns per publication of the factorizing wave
#endif
#ifndef BF_MFMA_RHSB
#define BF_MFMA_RHSB 8  // columns per batch of the forward-substituting waves (a multiple of BF_MFMA_PUB3)
#endif

// Two columns per MFMA (the instruction's K = 2): columns j and j + 1 (j even) live in the same half-wave; the 2 x 2
// pivot block is resolved in vector registers (L[:, j], then a'[j+1][:] = a[j+1][:] - L[j+1][j] L[:, j], its pivot and
// L[:, j+1]), one of the two operand columns crosses to the other half-wave (one cross-half shuffle), and ONE MFMA applies
// the rank-2 update.  16 dependent MFMAs per factorization.  The first pivot of the NEXT pair is known before the MFMA
// has finished: a[j+2][j+2] - L[j+2][j]^2 - L[j+2][j+1]^2 as two fmas in the matrix core's own order, so its reciprocal
// square root is computed in the shadow of the MFMA.
__device__ __forceinline__ float pair_operand(float c0, float c1, int h) {
  // slice k = 0 (lanes 0..31) <- column c0, slice k = 1 (lanes 32..63) <- column c1.  Both columns live in half h and
  // are ZERO in the other half, so one v_permlane32_swap (VALU; no LDS crossbar trip) assembles the operand:
  // (a, b) -> a = [a.lo | b.lo], b = [a.hi | b.hi]
  float a = c0, b = c1;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return h == 0 ? a : b;
}
__device__ __forceinline__ float both_halves(float v) {  // v (zero in one half) -> the same 32 values in both halves
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;   // a = [v.lo | v.lo], b = [v.hi | v.hi]; one of them is zero
}
// the 2 x 2 pivot block of the NEXT pair of columns, from its entries before the running MFMA's update (p..) and the
// current pair's columns at its two rows (u = row j+2, v = row j+3): the fmas the matrix core applies to those entries,
// in its order, so the scalars below equal what the accumulators will hold bit for bit
struct PivotBlock {
  float rinv0, s, rinv1;  // 1 / L[j][j], L[j+1][j], 1 / L[j+1][j+1]
};
__device__ __forceinline__ PivotBlock pivot_block(float p00, float p10, float p11, float u0, float u1, float v0, float v1) {
  const float a00 = fmaf(-u1, u1, fmaf(-u0, u0, p00));
  const float a10 = fmaf(-v1, u1, fmaf(-v0, u0, p10));
  const float a11 = fmaf(-v1, v1, fmaf(-v0, v0, p11));
  PivotBlock b;
  b.rinv0 = rsqrt_newton(a00);
  b.s = a10 * b.rinv0;
  b.rinv1 = rsqrt_newton(fmaf(-b.s, b.s, a11));
  return b;
}

// wave 3: chol(S + 1e-6); publishes L by columns (sL[32 j + i] = L[i][j], zeros above the diagonal), 1 / L[j][j] and the
// number of finished columns
__device__ __forceinline__ void eliminate_publish(f32x16& acc, lds_f* sL, lds_f* sRinv, lds_i* progress, int lane) {
  const int lr = lane & 31, half = lane >> 5;
  PivotBlock pb = pivot_block(rdlane_u(acc[0], 0), rdlane_u(acc[1], 0), rdlane_u(acc[1], 1), 0.f, 0.f, 0.f, 0.f);
  static_for<0, 16>([&](auto Pp) {
    constexpr int j = 2 * decltype(Pp)::value;
    constexpr int h = (j >> 2) & 1, rj = (j & 3) + 4 * (j >> 3);
    const float l0 = (half == h && lr >= j) ? acc[rj] * pb.rinv0 : 0.f;              // L[lr][j]
    const float row1 = fmaf(-pb.s, l0, acc[rj + 1]);                                 // a'[j+1][lr]
    const float l1 = (half == h && lr >= j + 1) ? row1 * pb.rinv1 : 0.f;             // L[lr][j+1]
    float p00 = 1.f, p10 = 0.f, p11 = 1.f, u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
    if constexpr (j < 30) {
      constexpr int h2 = ((j + 2) >> 2) & 1, r2 = ((j + 2) & 3) + 4 * ((j + 2) >> 3);
      p00 = rdlane_u(acc[r2], 32 * h2 + j + 2);                                      // the next block before this pair's update
      p10 = rdlane_u(acc[r2 + 1], 32 * h2 + j + 2);
      p11 = rdlane_u(acc[r2 + 1], 32 * h2 + j + 3);
      u0 = rdlane_u(l0, 32 * h + j + 2);
      u1 = rdlane_u(l1, 32 * h + j + 2);
      v0 = rdlane_u(l0, 32 * h + j + 3);
      v1 = rdlane_u(l1, 32 * h + j + 3);
    }
    if (half == h) {
      sL[32 * j + lr] = l0;
      sL[32 * (j + 1) + lr] = l1;
    }
    if (lane == 0) {
      sRinv[j] = pb.rinv0;
      sRinv[j + 1] = pb.rinv1;
    }
    if constexpr ((j + 2) % BF_MFMA_PUB3 == 0) {
      wave_lds_order();
      if (lane == 0) *progress = j + 2;
    }
    const float op = pair_operand(l0, l1, h);
    acc = mfma2(-op, op, acc);                                                       // a[i][n] -= L[i][j] L[n][j] + L[i][j+1] L[n][j+1]
    if constexpr (j < 30) pb = pivot_block(p00, p10, p11, u0, u1, v0, v1);           // in the shadow of the MFMA
  });
}

// wave 2: chol(S) the same way, z = L^-1 v, log N(v; 0, S) -- inference.py:104, :24
__device__ __forceinline__ float eliminate_loglik(f32x16& acc, lds_f* sv, int lane) {
  const int lr = lane & 31, half = lane >> 5;
  float rz = sv[lr], quad = 0.f, rprod = 1.f;
  PivotBlock pb = pivot_block(rdlane_u(acc[0], 0), rdlane_u(acc[1], 0), rdlane_u(acc[1], 1), 0.f, 0.f, 0.f, 0.f);
  static_for<0, 16>([&](auto Pp) {
    constexpr int j = 2 * decltype(Pp)::value;
    constexpr int h = (j >> 2) & 1, rj = (j & 3) + 4 * (j >> 3);
    const float l0 = (half == h && lr >= j) ? acc[rj] * pb.rinv0 : 0.f;
    const float row1 = fmaf(-pb.s, l0, acc[rj + 1]);
    const float l1 = (half == h && lr >= j + 1) ? row1 * pb.rinv1 : 0.f;
    const float r0 = pb.rinv0, r1 = pb.rinv1;
    float p00 = 1.f, p10 = 0.f, p11 = 1.f, u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
    if constexpr (j < 30) {
      constexpr int h2 = ((j + 2) >> 2) & 1, r2 = ((j + 2) & 3) + 4 * ((j + 2) >> 3);
      p00 = rdlane_u(acc[r2], 32 * h2 + j + 2);
      p10 = rdlane_u(acc[r2 + 1], 32 * h2 + j + 2);
      p11 = rdlane_u(acc[r2 + 1], 32 * h2 + j + 3);
      u0 = rdlane_u(l0, 32 * h + j + 2);
      u1 = rdlane_u(l1, 32 * h + j + 2);
      v0 = rdlane_u(l0, 32 * h + j + 3);
      v1 = rdlane_u(l1, 32 * h + j + 3);
    }
    const float op = pair_operand(l0, l1, h);
    acc = mfma2(-op, op, acc);
    if constexpr (j < 30) pb = pivot_block(p00, p10, p11, u0, u1, v0, v1);
    // z = L^-1 v on full-wave copies of the two columns
    const float f0 = both_halves(l0), f1 = both_halves(l1);
    const float z0 = rdlane_u(rz, j) * r0;
    rz = fmaf(-f0, z0, rz);
    const float z1 = rdlane_u(rz, j + 1) * r1;
    rz = fmaf(-f1, z1, rz);
    quad = fmaf(z1, z1, fmaf(z0, z0, quad));
    rprod *= r0 * r1;
  });
  return -0.5f * quad - 0.5f * 32.0f * 1.8378770664093453f + fast_log(rprod);
}

// waves 0, 1: forward substitution of their tile of H P (accumulators of phase A) behind the factorizing wave, two
// rows per MFMA; W -> sT rows 32..63, c = 1e-3 W^T g and m+ = m + W^T z for their 32 columns
__device__ __forceinline__ void eliminate_rhs(f32x16& acc, int c, lds_f* sL, lds_f* sRinv, lds_i* progress, lds_f* sT, lds_f* sv,
                                              lds_f* mcur, lds_f* mnxt, lds_f* scv, int lane) {
  constexpr int PP = 65;
  const int lr = lane & 31, half = lane >> 5;
  float rg = 1.0f, rz = sv[lr], acc_c = 0.f, acc_m = 0.f;
  auto wait_for = [&](int cols) {  // wave-uniform; the producer always reaches 32
    int seen = *(volatile lds_i*)progress;
    while (seen < cols) {
      __builtin_amdgcn_s_sleep(2);
      seen = *(volatile lds_i*)progress;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  // LDS only: a full fence would also drain the wave's output stores
  };
  // in batches of BF_MFMA_RHSB columns, a batch behind the factorizing wave: one poll and one round of LDS reads per batch
  static_for<0, 32 / BF_MFMA_RHSB>([&](auto Bt) {
    constexpr int j0 = decltype(Bt)::value * BF_MFMA_RHSB;
    wait_for(j0 + BF_MFMA_RHSB);
    float Lb[BF_MFMA_RHSB], rb[BF_MFMA_RHSB], sb[BF_MFMA_RHSB / 2];
    static_for<0, BF_MFMA_RHSB>([&](auto I) {
      Lb[decltype(I)::value] = sL[32 * (j0 + decltype(I)::value) + lr];   // L[lr][j], both halves
      rb[decltype(I)::value] = sRinv[j0 + decltype(I)::value];
    });
    static_for<0, BF_MFMA_RHSB / 2>([&](auto I) {
      constexpr int j = j0 + 2 * decltype(I)::value;
      sb[decltype(I)::value] = sL[32 * j + j + 1];                         // L[j+1][j]
    });
    static_for<0, BF_MFMA_RHSB / 2>([&](auto I) {
      constexpr int q = decltype(I)::value, j = j0 + 2 * q;
      constexpr int h = (j >> 2) & 1, rj = (j & 3) + 4 * (j >> 3);
      const float La = Lb[2 * q], Lc = Lb[2 * q + 1], ra = rb[2 * q], rc = rb[2 * q + 1];
      const float W0 = acc[rj] * ra;                                       // W[j][32 c + lr] in the lanes of half h
      const float W1 = fmaf(-sb[q], W0, acc[rj + 1]) * rc;                 // W[j+1][..]
      if (half == h) {
        sT[(32 + j) * PP + 32 * c + lr] = W0;
        sT[(33 + j) * PP + 32 * c + lr] = W1;
      }
      const float W0m = (half == h) ? W0 : 0.f, W1m = (half == h) ? W1 : 0.f;
      const float Bop = pair_operand(W0m, W1m, h);
      const float Aop = half == 0 ? -La : -Lc;
      acc = mfma2(Aop, Bop, acc);                                          // rhs[k][n] -= L[k][j] W[j][n] + L[k][j+1] W[j+1][n]
      const float g0 = rdlane_u(rg, j) * ra, z0 = rdlane_u(rz, j) * ra;    // g = L^-1 1, z = L^-1 v
      rg = fmaf(-La, g0, rg);
      rz = fmaf(-La, z0, rz);
      const float g1 = rdlane_u(rg, j + 1) * rc, z1 = rdlane_u(rz, j + 1) * rc;
      rg = fmaf(-Lc, g1, rg);
      rz = fmaf(-Lc, z1, rz);
      acc_c = fmaf(W1m, g1, fmaf(W0m, g0, acc_c));
      acc_m = fmaf(W1m, z1, fmaf(W0m, z0, acc_m));
    });
  });
  acc_c += __shfl_xor(acc_c, 32, 64);
  acc_m += __shfl_xor(acc_m, 32, 64);
  if (lane < 32) {
    scv[32 * c + lane] = acc_c * 1e-3f;                                // sqrt(1e-6) (W^T g): enters P+ as + c c^T
    mnxt[32 * c + lane] = mcur[32 * c + lane] + acc_m;                 // filtered mean
  }
}

// Loop-invariant operands are NOT to be kept in registers across steps: the compiler hoists the 16 + 16 + 32 loads of
// G Q G^T, D R D^T and A and the sixteen 64-bit store addresses of every output stream out of the time loop, and then
// spills them (106 scratch stores ahead of the loop, ~90 reloads per step at three workgroups per CU).  A wave-uniform
// base laundered through an empty asm once per step keeps each access a (scalar base + lane offset + immediate) form.
typedef __attribute__((address_space(1))) float gl_f;              // global memory: the laundered pointer must not decay to a flat one
typedef const __attribute__((address_space(1))) float gl_cf;
__device__ __forceinline__ int opaque_szero() {  // the scalar-register sibling of opaque_zero(): addresses stay wave-uniform
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}
__device__ __forceinline__ gl_f* per_step(float* p) { return (gl_f*)p + opaque_szero(); }
__device__ __forceinline__ gl_cf* per_step(const float* p) { return (gl_cf*)p + opaque_szero(); }
// one 32x32 accumulator tile (pi, pj) of a [N][N] stream entry at (b, t)
template <int N>
__device__ __forceinline__ void store_tile(const SView& sv, long long b, long long t, int pi, int pj, int lane, const f32x16& acc, int k = 0) {
  if (!sv.p) return;
  const int lr = lane & 31, lk = lane >> 5;
  gl_f* base = per_step(sv.p + b * sv.sB + k * sv.sK + t * sv.sT);
  const unsigned e0 = (unsigned)((32 * pi + 4 * lk) * N + 32 * pj + lr);
  if (sv.sE == 1) {
    BF_UNROLL for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], base + (e0 + ((r & 3) + 8 * (r >> 2)) * N));
  } else {
    const long long sE = sv.sE + (long long)opaque_szero();  // the sixteen 64-bit products below are per-step work too, not pre-loop registers
    const unsigned e0s = e0 + (unsigned)opaque_zero();
    BF_UNROLL for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[r], base + (long long)(e0s + ((r & 3) + 8 * (r >> 2)) * N) * sE);
  }
}

template <int N, int M, int VAR>
__global__ void __launch_bounds__(256, VAR == 4 ? 3 : 2)
kf_scan_mfma2_kernel(const MfmaConst<N, M>* __restrict__ cst, CView y, CarryView carry, OutViews out, long long B, long long T,
                     int rot_mode) {
  static_assert(N == 64 && M == 32, "tile assignment is written for n = 64, m = 32");
  constexpr int PP = N + 1, PS = M + 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // Which wave plays which role.  The two factorizing roles (2, 3) are VALU-issue bound -- one wave64 instruction per 4
  // cycles of their SIMD -- so they want the other workgroup's MFMA-only roles (0, 1) as SIMD partners, not its
  // factorizations.  Placement as measured (scripts/probes/hwid_probe.hip): the waves of a workgroup go round the SIMDs
  // in the order 0, 2, 1, 3 from wave 0's SIMD k, the second workgroup of a CU (b + 256, wave slot 1) starts one SIMD
  // further on, and blockIdx-rotated roles give both the same rotation: A's factorization of S + 1e-6 then shares a SIMD
  // with B's factorization of S.  Taking the rotation from where the hardware put wave 0 (HW_ID: SIMD, wave slot s) puts
  // role r at position (r + 2 s) & 3 of that order: the heavy roles of the two workgroups sit on disjoint SIMD pairs.
  // Any rotation is a valid assignment of roles; only the speed depends on it.
  __shared__ int s_rot;
  if (tid == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID: wave slot [3:0], SIMD [5:4]
    const unsigned k = (hw >> 4) & 3, pos = ((k & 1) << 1) | (k >> 1);
    s_rot = rot_mode == 0 ? (int)blockIdx.x : (int)(pos + 2 * (hw & 15));
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + s_rot) & 3);
  const int ti = wave >> 1, tj = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;
  const long long b = blockIdx.x;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sP = lds;                 // [64][65]  current covariance
  float* sT = sP + N * PP;         // [64][65]  rows 0..31: H P; rows 32..63: W; then A P+
  // Variant 4 (three workgroups per CU): A is read from the constant block into 32 registers per wave and step (its MFMA
  // operand for BOTH products, see phase J), D R D^T likewise, which brings the workgroup under a third of the CU's LDS.
  constexpr bool AREG = VAR == 4;
  float* sA = sT + N * PP;                     // [64][65]  A   (not in variant 4)
  float* sH = sA + (AREG ? 0 : N * PP);        // [32][65]  H
  float* sD = sH + M * PP;                     // [32][33]  D R D^T   (not in variant 4)
  float* sc2 = sD + (AREG ? 0 : M * PS);       // [32][33]  S as wave 2 sees it (layout change through LDS)
  float* sc3 = sc2 + M * PS;       // [32][33]  S as wave 3 sees it
  float* sm = sc3 + M * PS;        // [64] mean
  float* sm2 = sm + N;             // [64] mean (ping-pong)
  float* sv = sm2 + N;             // [32] innovation
  float* scv = sv + M;             // [64] 1e-3 W^T g

  if constexpr (!AREG) {
    for (int e = tid; e < N * N; e += 256) sA[(e / N) * PP + (e % N)] = cst->A[e];
    for (int e = tid; e < M * M; e += 256) sD[(e / M) * PS + (e % M)] = cst->DRD[e];
  }
  for (int e = tid; e < M * N; e += 256) sH[(e / N) * PP + (e % N)] = cst->H[e];
  BF_UNROLL for (int r = 0; r < 16; ++r)
    sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = carry.P_in[b * N * N + (32 * ti + c_row(r, lane)) * N + 32 * tj + lr];
  if (tid < N) sm[tid] = carry.m_in[b * N + tid];
  float w = carry.w_in ? carry.w_in[b] : 1.0f;
  float ynext = (wave == 2 && lane < M) ? y.p[b * y.sB + lane * y.sE] : 0.f;
  if (tid == 0) *reinterpret_cast<int*>(sc2 + 32) = 0;
  __syncthreads();

#ifdef BF_MFMA_PHASE_TIMERS
  long long tacc[12] = {0};
  long long tprev = wall_clock64();
#endif
  float* mcur = sm;
  float* mnxt = sm2;
  for (long long t = 0; t < T; ++t) {
    // ================= phase A: H P (waves 0,1); innovation (wave 2)
    f32x16 hp = {0};  // variant 3: the wave's tile of H P stays in its accumulators for the forward substitution
    if (wave < 2) {
      BF_UNROLL for (int s = 0; s < 32; ++s) hp = mfma2(sH[lr * PP + 2 * s + lk], sP[(2 * s + lk) * PP + 32 * tj + lr], hp);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[c_row(r, lane) * PP + 32 * tj + lr] = hp[r];
    } else if (wave == 2) {
      const float yv = ynext;
      const long long tn = t + 1 < T ? t + 1 : t;
      if (lane < M) ynext = y.p[b * y.sB + tn * y.sT + lane * y.sE];  // prefetch
      float s = 0.f;
      BF_UNROLL for (int q = 0; q < 32; ++q) s = fmaf(sH[lr * PP + 2 * q + lk], mcur[2 * q + lk], s);
      s += __shfl_xor(s, 32, 64);
      if (lane < M) sv[lane] = yv - (s + cst->Dr0[lane]);
    }
    BF_TICK(0)
    lds_barrier();
    BF_TICK(1)
    // ================= phases B + C (waves 2 and 3): S, its factorizations, W, c, m+
    float ll = 0.f;
    if (wave >= 2) {
      f32x16 acc;
      if constexpr (AREG) {
        gl_cf* drd = per_step(cst->DRD);
        BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = drd[c_row(r, lane) * M + lr];
      } else {
        BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = sD[c_row(r, lane) * PS + lr];
      }
      BF_UNROLL for (int s = 0; s < 32; ++s) acc = mfma2(sT[lr * PP + 2 * s + lk], sH[lr * PP + 2 * s + lk], acc);
      if constexpr (VAR == 2 || VAR == 4) {
        float* sc = wave == 2 ? sc2 : sc3;
        BF_UNROLL for (int r = 0; r < 16; ++r) sc[c_row(r, lane) * PS + lr] = acc[r];
        wave_lds_order();
        BF_TICK(10)
        if (wave == 3) chol_w_rows((lds_f*)sc3, (lds_f*)sT, (lds_f*)sv, (lds_f*)mcur, (lds_f*)mnxt, (lds_f*)scv, lane);
        else ll = chol_loglik_rows((lds_f*)sc2, (lds_f*)sv, lane);
      } else {
        if (wave == 3) {
          BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] += 1e-6f;   // psd_solve's jitter on every entry (utils.py:258)
          eliminate_publish(acc, (lds_f*)sc3, (lds_f*)sc2, (lds_i*)(sc2 + 32), lane);
        } else {
          ll = eliminate_loglik(acc, (lds_f*)sv, lane);
        }
      }
    } else if constexpr (VAR == 3) {
      eliminate_rhs(hp, tj, (lds_f*)sc3, (lds_f*)sc2, (lds_i*)(sc2 + 32), (lds_f*)sT, (lds_f*)sv, (lds_f*)mcur, (lds_f*)mnxt,
                    (lds_f*)scv, lane);
    }
    BF_TICK(2)
    lds_barrier();
    BF_TICK(3)
    // ================= phase H: P+ = P - W^T W + c c^T (all waves; K = 32 + 2); emit filtered streams
    f32x16 Pacc;
    // variant 4: A[32 ti + lr][2 s + lk], the wave's operand of A in phases I and J, fetched (L2-resident, 16 KB shared by
    // every workgroup) a phase ahead of its use and dropped before the factorization: nothing long-lived in registers
    float aop[AREG ? 32 : 1];
    if constexpr (AREG) {
      gl_cf* arow = per_step(cst->A) + ((32 * ti + lr) * N + lk);
      BF_UNROLL for (int q = 0; q < 32; ++q) aop[q] = arow[2 * q];
    }
    BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr];
    BF_UNROLL for (int s = 0; s < 16; ++s)
        Pacc = mfma2(-sT[(32 + 2 * s + lk) * PP + 32 * ti + lr], sT[(32 + 2 * s + lk) * PP + 32 * tj + lr], Pacc);
    Pacc = mfma2(lk == 0 ? scv[32 * ti + lr] : 0.f, lk == 0 ? scv[32 * tj + lr] : 0.f, Pacc);
    BF_UNROLL for (int r = 0; r < 16; ++r) sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = Pacc[r];
    store_tile<N>(out.P, b, t, ti, tj, lane, Pacc);
    if (wave == 1 && out.m.p) out.m.p[b * out.m.sB + t * out.m.sT + lane * out.m.sE] = mnxt[lane];
    if (VAR == 3 && wave == 3 && lane == 0) *reinterpret_cast<int*>(sc2 + 32) = 0;  // progress counter re-armed (three barriers ahead of its next use)
    if (wave == 2 && lane == 0) {
      w = reweight_single(ll, w);
      if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
      if (out.ll.p) out.ll.p[b * out.ll.sB + t * out.ll.sT] = ll;
    }
    BF_TICK(4)
    lds_barrier();
    BF_TICK(5)
    // ================= phase I: A P+ -> sT (all waves; K = 64); m- = A m+ + G q0 (waves 0, 3)
    {
      f32x16 acc = {0};
      BF_UNROLL for (int s = 0; s < 32; ++s)
          acc = mfma2(AREG ? aop[s] : sA[(32 * ti + lr) * PP + 2 * s + lk], sP[(2 * s + lk) * PP + 32 * tj + lr], acc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sT[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr] = acc[r];
    }
    if (wave == 0 || wave == 3) {
      float s = 0.f;
      BF_UNROLL for (int q = 0; q < 32; ++q) s = fmaf(AREG ? aop[q] : sA[(32 * ti + lr) * PP + 2 * q + lk], mnxt[2 * q + lk], s);
      s += __shfl_xor(s, 32, 64);
      if (lane < 32) mcur[32 * ti + lane] = s + cst->Gq0[32 * ti + lane];  // predicted mean
    }
    BF_TICK(6)
    lds_barrier();
    BF_TICK(7)
    // ================= phase J: P- = (A P+) A^T + G Q G^T (all waves; K = 64); emit predicted streams
    // Variant 4: the wave computes tile (tj, ti) instead of (ti, tj): its B operand A^T[k][32 ti + lr] = A[32 ti + lr][k]
    // is then the same 32 registers that were its A operand in phase I.
    {
      const int pi = AREG ? tj : ti, pj = AREG ? ti : tj;
      gl_cf* gqg = per_step(cst->GQG);
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = gqg[(32 * pi + c_row(r, lane)) * N + 32 * pj + lr];
      BF_UNROLL for (int s = 0; s < 32; ++s)
          Pacc = mfma2(sT[(32 * pi + lr) * PP + 2 * s + lk], AREG ? aop[s] : sA[(32 * pj + lr) * PP + 2 * s + lk], Pacc);
      BF_UNROLL for (int r = 0; r < 16; ++r) sP[(32 * pi + c_row(r, lane)) * PP + 32 * pj + lr] = Pacc[r];
      store_tile<N>(out.pP, b, t, pi, pj, lane, Pacc);
    }
    if (wave == 2 && out.pm.p) out.pm.p[b * out.pm.sB + t * out.pm.sT + lane * out.pm.sE] = mcur[lane];
    BF_TICK(8)
    lds_barrier();
    BF_TICK(9)
  }

  if (carry.P_out) BF_UNROLL for (int r = 0; r < 16; ++r)
      carry.P_out[b * N * N + (32 * ti + c_row(r, lane)) * N + 32 * tj + lr] = sP[(32 * ti + c_row(r, lane)) * PP + 32 * tj + lr];
  if (carry.m_out && tid < N) carry.m_out[b * N + tid] = mcur[tid];
  if (carry.w_out && wave == 2 && lane == 0) carry.w_out[b] = w;
#ifdef BF_MFMA_PHASE_TIMERS
  __syncthreads();
  if (b == 0 && lane == 0 && carry.P_out) for (int i = 0; i < 12; ++i) carry.P_out[wave * 16 + i] = (float)tacc[i];
#endif
}

// =======================================================================================================================
// store_tile for a model of nr <= N states riding zero-padded in the N x N tiles: entries (row, col) with both < nr, at the
// model's own row length
template <int N>
__device__ __forceinline__ void store_tile_n(const SView& sv, long long b, long long t, int pi, int pj, int lane, const f32x16& acc, int nr, int k = 0) {
  if (nr == N) return store_tile<N>(sv, b, t, pi, pj, lane, acc, k);
  if (!sv.p) return;
  const int lr = lane & 31, lk = lane >> 5;
  gl_f* base = per_step(sv.p + b * sv.sB + k * sv.sK + t * sv.sT);
  const int col = 32 * pj + lr;
  const long long sE = sv.sE + (long long)opaque_szero();
  BF_UNROLL for (int r = 0; r < 16; ++r) {
    const int row = 32 * pi + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (col < nr && row < nr) __builtin_nontemporal_store(acc[r], base + (long long)(row * nr + col) * sE);
  }
}

// Variant 5: the five matrix products off the fp32 datapath.  On gfx950 v_mfma_f32_32x32x2_f32 and the fp32 vector
// instructions share ONE datapath per SIMD (profiles/r02_f32_pipe_probe.txt), so the fp32 products (13.3 us of the
// 20.4 us of SIMD-time a step needs) cannot hide behind the factorization.  v_mfma_f32_32x32x16_bf16 runs at 16x the
// rate; with every operand written as the EXACT sum of three bf16 terms (x = hi + mid + lo: 8 + 8 + 8 significand bits)
// and the six cross terms of weight >= 2^-16 accumulated in fp32, a product costs 6/16 of the fp32 MFMA time at the same
// rounding (scripts/probes/bf16x3_check.hip: 9.3e-8 of sum |terms| against 1.1e-7 for the fp32 MFMA).
//
// Layouts.  The bf16 MFMA wants 8 consecutive k per lane for both operands: A-operand X[m][k] row-major, B-operand as
// Yt[n][k] = Y[k][n].  An accumulator tile holds, per lane, one column and four groups of four consecutive rows, so its
// cheap store is the TRANSPOSED one, T[col][row] (8-byte stores of 4 terms): stored that way a result Z serves as the
// A-operand of Z^T . and as the B-operand of . Z.  The products are arranged so that nothing else is ever needed:
//   A   Z = (H P-)^T = P-^T H^T        A-op: P- as stored (J), B-op: H (registers)          -> Z stored; H P also fp32
//   B   S^T = H Z                       A-op: H (registers),    B-op: Z as stored
//   C   factorizations as in variant 2; W^T written as bf16 terms row-major (each lane owns a column of W)
//   H   P+ = P- - W^T W + c c^T         A-op: -W^T, B-op: W^T (the same array); P- from the accumulators of J
//   I   Y^T = (A P+)^T = P+^T A^T       A-op: P+ as stored (H), B-op: A rows (registers)    -> Y^T stored
//   J   P- = Y A^T + G Q G^T            A-op: Y^T as stored (I) = Y row-major, B-op: A rows (registers)
// A and H live in registers as bf16 terms (2 x 48 VGPRs per wave); the matrix-vector products rebuild their fp32
// values from the terms (hi + mid + lo is exact).  LDS: 72.8 KB per workgroup, two workgroups per CU.
__device__ __forceinline__ f32x16 mfma_bf6(const u32x4* a, const u32x4* b, f32x16 c) {  // smallest cross terms first
  auto m = [](u32x4 x, u32x4 y, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0);
  };
  c = m(a[1], b[1], c);
  c = m(a[0], b[2], c);
  c = m(a[2], b[0], c);
  c = m(a[0], b[1], c);
  c = m(a[1], b[0], c);
  c = m(a[0], b[0], c);
  return c;
}
// accumulator tile (row tile rt, column tile ct) -> dst[term][32 ct + col][32 rt + row] as bf16 terms, `pitch` bytes per column
__device__ __forceinline__ void store_terms_transposed(lds_c* dst, int term_bytes, int pitch, int rt, int ct, int lane, const f32x16& acc) {
  const int lr = lane & 31, lk = lane >> 5;
  lds_c* base = dst + (32 * ct + lr) * pitch + (32 * rt + 4 * lk) * 2;
  BF_UNROLL for (int g = 0; g < 4; ++g) {  // rows 8 g + 4 lk + 0..3 of the tile
    const Split3 a = split_pair(acc[4 * g], acc[4 * g + 1]), b = split_pair(acc[4 * g + 2], acc[4 * g + 3]);
    *reinterpret_cast<lds_u32x2*>(base + 16 * g) = u32x2{a.hi, b.hi};
    *reinterpret_cast<lds_u32x2*>(base + term_bytes + 16 * g) = u32x2{a.mid, b.mid};
    *reinterpret_cast<lds_u32x2*>(base + 2 * term_bytes + 16 * g) = u32x2{a.lo, b.lo};
  }
}
__device__ __forceinline__ void load_terms(u32x4* dst, const lds_c* arr, int term_bytes, int pitch, int row, int chunk, int lk) {
  const lds_c* p = arr + row * pitch + (16 * chunk + 8 * lk) * 2;
  BF_UNROLL for (int t = 0; t < 3; ++t) dst[t] = *reinterpret_cast<const lds_u32x4*>(p + t * term_bytes);
}
// sum_k X[row][k] v[k] over the lane's 32 k (16 c + 8 lk + 0..7) from the register terms of X
__device__ __forceinline__ float dot_terms(const u32x4 (*x)[4], const float* v, int lk) {
  float s = 0.f;
  BF_UNROLL for (int c = 0; c < 4; ++c) BF_UNROLL for (int d = 0; d < 4; ++d) {
    const float x0 = (bf_lo(x[0][c][d]) + bf_lo(x[1][c][d])) + bf_lo(x[2][c][d]);
    const float x1 = (bf_hi(x[0][c][d]) + bf_hi(x[1][c][d])) + bf_hi(x[2][c][d]);
    s = fmaf(x0, v[16 * c + 8 * lk + 2 * d], s);
    s = fmaf(x1, v[16 * c + 8 * lk + 2 * d + 1], s);
  }
  return s;
}

// the same over half of the k range (chunks 2 h, 2 h + 1)
__device__ __forceinline__ float dot_terms_half(const u32x4 (*x)[4], const float* v, int lk, int h) {
  float s = 0.f;
  BF_UNROLL for (int cc = 0; cc < 2; ++cc) BF_UNROLL for (int d = 0; d < 4; ++d) {
    const u32x4 x0v = h ? x[0][2 + cc] : x[0][cc], x1v = h ? x[1][2 + cc] : x[1][cc], x2v = h ? x[2][2 + cc] : x[2][cc];
    const float x0 = (bf_lo(x0v[d]) + bf_lo(x1v[d])) + bf_lo(x2v[d]);
    const float x1 = (bf_hi(x0v[d]) + bf_hi(x1v[d])) + bf_hi(x2v[d]);
    const int k = 16 * (2 * h + cc) + 8 * lk + 2 * d;
    s = fmaf(x0, v[k], s);
    s = fmaf(x1, v[k + 1], s);
  }
  return s;
}

// MULTI: the K Gaussian-sum components of a LINEAR model as independent chains (see kf_scan_bf32_kernel): workgroup c =
// trajectory * K + component reads trajectory c / K's observations and writes component c % K's streams and per-step
// log-likelihood; the weights follow in gsf_reweight_kernel.  TV: per-step G Q_t G^T / D R_t D^T tables (64 x 64 and 32 x 32
// floats per step) instead of the constants -- inference.py:21,337-353.
// DYN: 0 = linear; 1 = Lorenz-96, 2 = sine dynamics as extended Kalman chains (see kf_scan_bf32_kernel): the wave's rows of
// F = df/dx at the filtered mean are re-evaluated and re-split into the A operand registers every step.
template <int N, int M, bool MULTI = false, bool TV = false, int DYN = 0>
__global__ void __launch_bounds__(256, 2)
kf_scan_mfma5_kernel(const MfmaConst<N, M>* __restrict__ cst, CView y, CarryView carry, OutViews out, long long B, long long T,
                     int rot_mode, int nr, int mr, int K, const float* __restrict__ tvq, const float* __restrict__ tvr) {
  // nr <= 64, mr <= 32: the model's own dimensions (streams and carry are laid out for them); inside, everything is (64, 32)
  static_assert(N == 64 && M == 32, "tile assignment is written for n = 64, m = 32");
  constexpr int PS = M + 1, HPP = N + 1;
  constexpr int PITCH = 144, PN_TERM = 64 * PITCH, ZN_TERM = 32 * PITCH, WT_PITCH = 80, WT_TERM = 64 * WT_PITCH;
  constexpr int OFF_ZN = 3 * PN_TERM, OFF_WT = OFF_ZN + 3 * ZN_TERM, OFF_YN = OFF_ZN, OFF_F32 = OFF_WT + 3 * WT_TERM;
  static_assert(3 * PN_TERM <= 3 * ZN_TERM + 3 * WT_TERM, "Y^T aliases Z and W^T");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int lr = lane & 31, lk = lane >> 5;
  __shared__ int s_rot;
  if (tid == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned k = (hw >> 4) & 3, pos = ((k & 1) << 1) | (k >> 1);
    s_rot = rot_mode == 0 ? (int)blockIdx.x : (int)(pos + 2 * (hw & 15));
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + s_rot) & 3);
  const int ti = wave >> 1, tj = wave & 1;
  const long long b = blockIdx.x;                  // chain: carry index
  const long long bt = MULTI ? b / K : b;          // trajectory: observations, stream batch index
  const int kc = MULTI ? (int)(b % K) : 0;         // component: stream component index

  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_c* L = (lds_c*)reinterpret_cast<char*>(lds);
  lds_c* Pn = L;             // [3][64][144 B]  P- (after J) / P+ (after H), transposed terms
  lds_c* Zn = L + OFF_ZN;    // [3][32][144 B]  Z = (H P-)^T
  lds_c* Wt = L + OFF_WT;    // [3][64][80 B]   W^T, row-major terms
  lds_c* Yn = L + OFF_YN;    // [3][64][144 B]  Y^T (aliases Zn, Wt)
  float* sHP = lds + OFF_F32 / 4;   // [32][65]  H P- in fp32 for the forward substitution
  float* sc2 = sHP + M * HPP;       // [32][33]  S as wave 2 sees it
  float* sc3 = sc2 + M * PS;        // [32][33]  S as wave 3 sees it
  float* sm = sc3 + M * PS;         // [64] mean
  float* sm2 = sm + N;              // [64] mean (ping-pong)
  float* sv = sm2 + N;              // [32] innovation
  float* scv = sv + M;              // [64] 1e-3 W^T g
  float* part = scv + N;            // [2][64] halves (over k) of the two matrix-vector products
  float* sv3 = part + 2 * N;        // [32] innovation, wave 3's copy

  // A rows 32 tj + lr and H row lr as bf16 terms: the wave's B-operand in phases A, I, J and A-operand in phase B
  u32x4 hop[3][4], aop[3][4];
  BF_UNROLL for (int t = 0; t < 3; ++t) BF_UNROLL for (int c = 0; c < 4; ++c) {
    hop[t][c] = *reinterpret_cast<const u32x4*>(&cst->H3[t][lr * N + 16 * c + 8 * lk]);
    if constexpr (DYN == 0) aop[t][c] = *reinterpret_cast<const u32x4*>(&cst->A3[t][(32 * tj + lr) * N + 16 * c + 8 * lk]);
  }
  f32x16 Pacc;  // the wave's tile of P-: carried in registers from phase J to phase H
  const bool col_ok = 32 * tj + lr < nr;
  BF_UNROLL for (int r = 0; r < 16; ++r) {
    const int row = 32 * ti + c_row(r, lane);
    Pacc[r] = (col_ok && row < nr) ? carry.P_in[b * nr * nr + row * nr + 32 * tj + lr] : 0.f;
  }
  store_terms_transposed(Pn, PN_TERM, PITCH, ti, tj, lane, Pacc);
  if (tid < N) sm[tid] = tid < nr ? carry.m_in[b * nr + tid] : 0.f;
  float w = (!MULTI && carry.w_in) ? carry.w_in[b] : 1.0f;
  float ynext = (wave >= 2 && lane < mr) ? y.p[bt * y.sB + lane * y.sE] : 0.f;
  const float ll_pad = 0.5f * 1.8378770664093453f * (float)(M - mr);   // the padded observations' log N(0; 0, 1), taken off
  const float dr0 = cst->Dr0[lr], gq0 = cst->Gq0[lane];
  __syncthreads();

#ifdef BF_MFMA_PHASE_TIMERS
  long long tacc[12] = {0};
  long long tprev = wall_clock64();
#define BF_TICK5(i) { const long long tn_ = wall_clock64(); tacc[i] += tn_ - tprev; tprev = tn_; }
#else
#define BF_TICK5(i)
#endif
  float* mcur = sm;
  float* mnxt = sm2;
  for (long long t = 0; t < T; ++t) {
    float yv_t = ynext;                      // this step's observation (waves 2, 3), the same for every component
    if (wave >= 2) {
      const long long tn = t + 1 < T ? t + 1 : t;
      if (lane < mr) ynext = y.p[bt * y.sB + tn * y.sT + lane * y.sE];  // prefetch
    }
    gl_cf* const drd_t = TV && tvr ? per_step(tvr + t * (M * M)) : per_step(cst->DRD);
    gl_cf* const gqg_t = TV && tvq ? per_step(tvq + t * (N * N)) : per_step(cst->GQG);
    // ================= phase A: Z = P-^T H^T (waves 0, 1: row tile = wave); innovation (wave 2)
    if (wave < 2) {
      f32x16 z = {0};
      BF_UNROLL for (int c = 0; c < 4; ++c) {
        u32x4 a[3];
        load_terms(a, Pn, PN_TERM, PITCH, 32 * wave + lr, c, lk);
        const u32x4 bh[3] = {hop[0][c], hop[1][c], hop[2][c]};
        z = mfma_bf6(a, bh, z);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) sHP[lr * HPP + 32 * wave + c_row(r, lane)] = z[r];   // (H P-)[lr][.]
      store_terms_transposed(Zn, ZN_TERM, PITCH, wave, 0, lane, z);
    } else {
      // H m-: waves 2 and 3 (idle in this phase) take half of the k range each; the halves meet after the barrier
      float s = dot_terms_half(hop, mcur, lk, wave - 2);
      s += __shfl_xor(s, 32, 64);
      if (lane < M) part[(wave - 2) * N + lane] = s;
    }
    BF_TICK5(0)
    lds_barrier();
    BF_TICK5(1)
    // ================= phases B + C (waves 2 and 3): S^T = H Z + (D R D^T)^T, the factorizations, W^T, c, m+
    float ll = 0.f;
    if (wave >= 2 + BF_V5_LL_LEMMA) {
      {  // innovation v = y - (H m- + D r0): both factorizing waves form it (same bits) for their own use
        if (lane < M) (wave == 2 ? sv : sv3)[lane] = yv_t - ((part[lane] + part[N + lane]) + dr0);
      }
      f32x16 acc;
      BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = drd_t[lr * M + c_row(r, lane)];
      BF_UNROLL for (int c = 0; c < 4; ++c) {
        u32x4 bz[3];
        load_terms(bz, Zn, ZN_TERM, PITCH, lr, c, lk);
        const u32x4 ah[3] = {hop[0][c], hop[1][c], hop[2][c]};
        acc = mfma_bf6(ah, bz, acc);
      }
      float* sc = wave == 2 ? sc2 : sc3;
      BF_UNROLL for (int r = 0; r < 16; ++r) sc[lr * PS + c_row(r, lane)] = acc[r];   // S[lr][.] = S^T[.][lr]
      wave_lds_order();
      BF_TICK5(10)
#if BF_V5_INLINE
      if (wave == 3) chol_w_rows_impl<true>((lds_f*)sc3, (lds_f*)sHP, (lds_f*)sv3, (lds_f*)mcur, (lds_f*)mnxt, (lds_f*)scv, Wt, lane);
      else ll = chol_loglik_rows_impl((lds_f*)sc2, (lds_f*)sv, lane) + ll_pad;
#elif BF_V5_LL_LEMMA
      ll = chol_w_rows_bf_ll((lds_f*)sc3, (lds_f*)sHP, (lds_f*)sv3, (lds_f*)mcur, (lds_f*)mnxt, (lds_f*)scv, Wt, lane) + ll_pad;
#else
      if (wave == 3) chol_w_rows_bf((lds_f*)sc3, (lds_f*)sHP, (lds_f*)sv3, (lds_f*)mcur, (lds_f*)mnxt, (lds_f*)scv, Wt, lane);
      else ll = chol_loglik_rows((lds_f*)sc2, (lds_f*)sv, lane) + ll_pad;
#endif
    }
    BF_TICK5(2)
    lds_barrier();
    BF_TICK5(3)
    // ================= phase H: P+ = P- - W^T W + c c^T (K = 32 + 2); emit filtered streams; P+ stored as terms
    {
      f32x16 acc = Pacc;
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 a[3], bw[3];
        load_terms(a, Wt, WT_TERM, WT_PITCH, 32 * ti + lr, c, lk);
        load_terms(bw, Wt, WT_TERM, WT_PITCH, 32 * tj + lr, c, lk);
        BF_UNROLL for (int q = 0; q < 3; ++q) a[q] ^= 0x80008000u;   // -W^T
        acc = mfma_bf6(a, bw, acc);
      }
      acc = mfma2(lk == 0 ? scv[32 * ti + lr] : 0.f, lk == 0 ? scv[32 * tj + lr] : 0.f, acc);
      store_tile_n<N>(out.P, bt, t, ti, tj, lane, acc, nr, kc);
      store_terms_transposed(Pn, PN_TERM, PITCH, ti, tj, lane, acc);
    }
    if (wave == 1 && out.m.p && lane < nr) out.m.p[bt * out.m.sB + kc * out.m.sK + t * out.m.sT + lane * out.m.sE] = mnxt[lane];
    if (wave == 2 + BF_V5_LL_LEMMA && lane == 0) {
      if constexpr (!MULTI) {
        w = reweight_single(ll, w);
        if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
      }
      if (out.ll.p) out.ll.p[bt * out.ll.sB + kc * out.ll.sK + t * out.ll.sT] = ll;   // (MULTI: the launcher always provides it)
    }
    BF_TICK5(4)
    lds_barrier();
    BF_TICK5(5)
    // ================= phase I: Y^T = P+^T A^T (K = 64); m- = A m+ + G q0 (waves 0, 1: rows 32 tj + lr)
    // Fetched here, a phase ahead of their use: the tile of G Q G^T (added after phase J's products) and H's terms for the
    // next step's phases A and B (12 KB shared by every wave of the CU) -- an L2 round trip under this load is ~1 us, and
    // held through the factorization instead the 48 + 16 registers spill
    float gq[16];
    {
#if BF_V5_HOP_RELOAD
      const int oz = opaque_szero();
      BF_UNROLL for (int q = 0; q < 3; ++q) BF_UNROLL for (int c = 0; c < 4; ++c)
          hop[q][c] = *reinterpret_cast<const u32x4*>(&cst->H3[q][lr * N + 16 * c + 8 * lk + oz]);
#endif
      BF_UNROLL for (int r = 0; r < 16; ++r) gq[r] = gqg_t[(32 * ti + c_row(r, lane)) * N + 32 * tj + lr];
    }
    if constexpr (DYN != 0) {   // row 32 tj + lr of F at the filtered mean (mnxt), columns 16 c + 8 lk + e; f of that row
      gl_cf* th = per_step(cst->dth);
      const int row = 32 * tj + lr;
      float fv = 0.f;
      BF_UNROLL for (int c = 0; c < 4; ++c) {
        float fr[8];
        BF_UNROLL for (int e = 0; e < 8; ++e) fr[e] = 0.f;
        if (row < nr) {
          if constexpr (DYN == 1) {
            const float alpha = th[0], beta = th[1], gamma = th[2], dt = th[3];
            const bool mp = th[4] != 0.f;
            const int im1 = (row + nr - 1) % nr, ip1 = (row + 1) % nr, im2 = (row + 2 * nr - 2) % nr;
            const float xi = mnxt[row], ax = mnxt[im1];
            const float bx = mp ? (mnxt[ip1] - mnxt[im2]) : 0.f;
            fv = xi + dt * (alpha * (ax * bx) - beta * xi + gamma);
            BF_UNROLL for (int e = 0; e < 8; ++e) {
              const int j = 16 * c + 8 * lk + e;
              float v = 0.f;
              if (j == row) v += 1.0f - dt * beta;
              if (mp) {
                if (j == im1) v += dt * alpha * bx;
                if (j == ip1) v += dt * alpha * ax;
                if (j == im2) v -= dt * alpha * ax;
              }
              fr[e] = v;
            }
          } else {
            const float w0 = th[0], xi = mnxt[row];
            fv = sinf(w0 * xi);
            const float d = w0 * cosf(w0 * xi);
            BF_UNROLL for (int e = 0; e < 8; ++e) fr[e] = (16 * c + 8 * lk + e == row) ? d : 0.f;
          }
        }
        BF_UNROLL for (int d = 0; d < 4; ++d) {
          const Split3 sp = split_pair(fr[2 * d], fr[2 * d + 1]);
          aop[0][c][d] = sp.hi; aop[1][c][d] = sp.mid; aop[2][c][d] = sp.lo;
        }
      }
      if (ti == 0 && lane < 32) { part[32 * tj + lane] = fv; part[N + 32 * tj + lane] = 0.f; }   // (read back as part[.] + part[N + .])
    }
    {
      f32x16 acc = {0};
      BF_UNROLL for (int c = 0; c < 4; ++c) {
        u32x4 a[3];
        load_terms(a, Pn, PN_TERM, PITCH, 32 * ti + lr, c, lk);
        const u32x4 ba[3] = {aop[0][c], aop[1][c], aop[2][c]};
        acc = mfma_bf6(a, ba, acc);
      }
      store_terms_transposed(Yn, PN_TERM, PITCH, ti, tj, lane, acc);
    }
    if constexpr (DYN == 0) {  // A m+: the two waves holding rows 32 tj + lr take half of the k range each
      float s = dot_terms_half(aop, mnxt, lk, ti);
      s += __shfl_xor(s, 32, 64);
      if (lane < 32) part[ti * N + 32 * tj + lane] = s;
    }
    BF_TICK5(6)
    lds_barrier();
    BF_TICK5(7)
    // ================= phase J: P- = Y A^T + G Q G^T (K = 64); emit predicted streams; P- stored as terms
    {
      // H's terms for the next step's phases A and B are fetched here (12 KB shared by every wave of the CU: L1 hits) rather
      // than held through the factorization, where 48 more live registers spill
      if (wave == 2) mcur[lane] = (part[lane] + part[N + lane]) + gq0;   // predicted mean m- = A m+ + G q0
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = 0.f;
      BF_UNROLL for (int c = 0; c < 4; ++c) {
        u32x4 a[3];
        load_terms(a, Yn, PN_TERM, PITCH, 32 * ti + lr, c, lk);
        const u32x4 ba[3] = {aop[0][c], aop[1][c], aop[2][c]};
        Pacc = mfma_bf6(a, ba, Pacc);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] += gq[r];
      store_tile_n<N>(out.pP, bt, t, ti, tj, lane, Pacc, nr, kc);
      store_terms_transposed(Pn, PN_TERM, PITCH, ti, tj, lane, Pacc);
    }
    if (wave == 2 && out.pm.p && lane < nr) out.pm.p[bt * out.pm.sB + kc * out.pm.sK + t * out.pm.sT + lane * out.pm.sE] = (part[lane] + part[N + lane]) + gq0;
    BF_TICK5(8)
    lds_barrier();
    BF_TICK5(9)
  }

  if (carry.P_out && col_ok) BF_UNROLL for (int r = 0; r < 16; ++r) {
      const int row = 32 * ti + c_row(r, lane);
      if (row < nr) carry.P_out[b * nr * nr + row * nr + 32 * tj + lr] = Pacc[r];
    }
  if (carry.m_out && tid < nr) carry.m_out[b * nr + tid] = mcur[tid];
  if (!MULTI && carry.w_out && wave == 2 + BF_V5_LL_LEMMA && lane == 0) carry.w_out[b] = w;
#ifdef BF_MFMA_PHASE_TIMERS
  __syncthreads();
  if (b == 0 && lane == 0 && carry.P_out) for (int i = 0; i < 12; ++i) carry.P_out[wave * 16 + i] = (float)tacc[i];
#endif
}

// =======================================================================================================================
// n <= 32, m <= 32: ONE WAVE per trajectory.  Every matrix is a single 32 x 32 tile, so the whole step of variant 5 --
// Z = P-^T H^T, S^T = H Z, the two factorizations, P+ = P- - W^T W + c c^T, Y^T = P+^T A^T, P- = Y A^T + G Q G^T, all as
// three-term bf16 products -- runs inside one wave without a single barrier (a wave's LDS traffic executes in issue
// order), no wave ever waits for another's factorization, and the CU holds eight independent trajectories (16.6 KB of LDS
// each, two per workgroup) instead of two workgroups with three of four waves idle through the serial phase.
// Smaller models ride zero-padded in the tile exactly as in launch_kf_mfma.
struct Bf32Const {
  unsigned short A3[3][32 * 32], H3[3][32 * 32];
  float GQG[32 * 32], DRD[32 * 32], Gq0[32], Dr0[32];
  float dth[8];   // DYN != 0: the registry dynamics' scalars (Lorenz-96: alpha, beta, gamma, dt, mode; sine: w0)
};
__device__ __forceinline__ float dot_terms32(const u32x4 (*x)[2], const float* v, int lk) {  // sum over the lane's 16 k
  float s = 0.f;
  BF_UNROLL for (int c = 0; c < 2; ++c) BF_UNROLL for (int d = 0; d < 4; ++d) {
    const float x0 = (bf_lo(x[0][c][d]) + bf_lo(x[1][c][d])) + bf_lo(x[2][c][d]);
    const float x1 = (bf_hi(x[0][c][d]) + bf_hi(x[1][c][d])) + bf_hi(x[2][c][d]);
    s = fmaf(x0, v[16 * c + 8 * lk + 2 * d], s);
    s = fmaf(x1, v[16 * c + 8 * lk + 2 * d + 1], s);
  }
  return s;
}
// one 32 x 32 accumulator tile of a [nr][nr] stream entry at (b, t)
__device__ __forceinline__ void store_tile32(const SView& sv, long long b, long long t, int lane, const f32x16& acc, int nr, int k = 0) {
  if (!sv.p) return;
  const int lr = lane & 31, lk = lane >> 5;
  gl_f* base = per_step(sv.p + b * sv.sB + k * sv.sK + t * sv.sT);
  const long long sE = sv.sE + (long long)opaque_szero();
  BF_UNROLL for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (lr < nr && row < nr) __builtin_nontemporal_store(acc[r], base + (long long)(row * nr + lr) * sE);
  }
}

// bytes per trajectory: [P-/P+ terms | fp32 H P and S, which live only between the last read of P-'s terms (phase A) and the
// factorization's first instructions, while that buffer is idle] + [Z / W^T / Y^T terms] + four 32-vectors
constexpr int BF32_PN_BYTES = 2 * 32 * 33 * 4;   // 8 448 >= 3 * 32 * 80
constexpr int BF32_WAVE_LDS = BF32_PN_BYTES + 3 * 32 * 80 + 4 * 32 * 4;

// MULTI: the K Gaussian-sum components of a LINEAR model (inference.py:345-353 vmaps _condition_on / _predict over them).  Their
// mean / covariance recursions do not depend on the weights, so every (trajectory, component) pair is a chain of its own:
// chain c = trajectory * K + component reads trajectory c / K's observations, writes component c % K's streams and its
// per-step log-likelihood; the weight recursion (the only coupling) runs afterwards over the stored log-likelihoods
// (gsf_reweight_kernel).  B counts chains.  TV: per-step G Q_t G^T / D R_t D^T tables (_get_params(x, 2, t),
// inference.py:21,337-340) instead of the constants of Bf32Const.
// DYN: 0 = linear dynamics (A as constant operand registers); 1 = Lorenz-96, 2 = sine (models.hpp: DYN_LORENZ96 / DYN_SINE):
// an extended Kalman filter chain -- row lr of F = df/dx at the filtered mean is evaluated analytically every step
// (inference.py:328, :61-62 take it with jacfwd), split into its three bf16 terms in the SAME operand registers, and the
// predicted mean is f(m+) + F_q q0 instead of A m+ + G q0 (identity noise input).
template <bool MULTI, bool TV, int DYN = 0>
__global__ void __launch_bounds__(128, 2)
kf_scan_bf32_kernel(const Bf32Const* __restrict__ cst, CView y, CarryView carry, OutViews out, long long B, long long T, int nr, int mr,
                    int K, const float* __restrict__ tvq, const float* __restrict__ tvr) {
  constexpr int PITCH = 80, TERM = 32 * PITCH, PS = 33;
  const int lane = threadIdx.x & 63;
  const int lr = lane & 31, lk = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long b_raw = (long long)blockIdx.x * 2 + wv;
  if (b_raw >= B) return;   // (no workgroup barrier anywhere below)
  const long long b = b_raw;                       // chain: carry index
  const long long bt = MULTI ? b / K : b;          // trajectory: observations, stream batch index
  const int kc = MULTI ? (int)(b % K) : 0;         // component: stream component index

  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_c* L = (lds_c*)reinterpret_cast<char*>(lds) + wv * BF32_WAVE_LDS;
  lds_c* Pn = L;                       // [3][32][80 B]  P- / P+, transposed terms
  lds_c* Zn = L + BF32_PN_BYTES;       // [3][32][80 B]  Z = (H P-)^T; later W^T, then Y^T
  lds_c* Wt = Zn;
  lds_c* Yn = Zn;
  float* sHP = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + wv * BF32_WAVE_LDS);   // [32][33] H P- (fp32), over Pn
  float* sc = sHP + 32 * PS;           // [32][33]  S, over Pn
  float* sm = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + wv * BF32_WAVE_LDS + BF32_PN_BYTES + 3 * TERM);   // [32] predicted mean
  float* sm2 = sm + 32;                // [32] filtered mean
  float* sv = sm2 + 32;                // [32] innovation
  float* scv = sv + 32;                // [32] 1e-3 W^T g

  u32x4 hop[3][2], aop[3][2];          // row lr of H and of A as bf16 terms
  BF_UNROLL for (int q = 0; q < 3; ++q) BF_UNROLL for (int c = 0; c < 2; ++c) {
    hop[q][c] = *reinterpret_cast<const u32x4*>(&cst->H3[q][lr * 32 + 16 * c + 8 * lk]);
    if constexpr (DYN == 0) aop[q][c] = *reinterpret_cast<const u32x4*>(&cst->A3[q][lr * 32 + 16 * c + 8 * lk]);
  }
  const float dr0 = cst->Dr0[lr], gq0 = cst->Gq0[lr];
  f32x16 Pacc;
  BF_UNROLL for (int r = 0; r < 16; ++r) {
    const int row = c_row(r, lane);
    Pacc[r] = (lr < nr && row < nr) ? carry.P_in[b * nr * nr + row * nr + lr] : 0.f;
  }
  store_terms_transposed(Pn, TERM, PITCH, 0, 0, lane, Pacc);
  sm[lr] = lr < nr ? carry.m_in[b * nr + lr] : 0.f;
  float w = (!MULTI && carry.w_in) ? carry.w_in[b] : 1.0f;
  float ynext = lr < mr ? y.p[bt * y.sB + lr * y.sE] : 0.f;
  const float ll_pad = 0.5f * 1.8378770664093453f * (float)(32 - mr);
  wave_lds_order();

  for (long long t = 0; t < T; ++t) {
    const float yv = ynext;
    {
      const long long tn = t + 1 < T ? t + 1 : t;
      if (lr < mr) ynext = y.p[bt * y.sB + tn * y.sT + lr * y.sE];
    }
    gl_cf* const drd_t = TV && tvr ? per_step(tvr + t * 1024) : per_step(cst->DRD);
    gl_cf* const gqg_t = TV && tvq ? per_step(tvq + t * 1024) : per_step(cst->GQG);
    // ---- Z = P-^T H^T; H P- in fp32 for the forward substitution; innovation
    {
      f32x16 z = {0};
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 a[3];
        load_terms(a, Pn, TERM, PITCH, lr, c, lk);
        const u32x4 bh[3] = {hop[0][c], hop[1][c], hop[2][c]};
        z = mfma_bf6(a, bh, z);
      }
      wave_lds_order();   // P-'s terms have been read: their buffer now takes H P (fp32) and, below, S
      BF_UNROLL for (int r = 0; r < 16; ++r) sHP[lr * PS + c_row(r, lane)] = z[r];
      store_terms_transposed(Zn, TERM, PITCH, 0, 0, lane, z);
      float s = dot_terms32(hop, sm, lk);
      s += __shfl_xor(s, 32, 64);
      sv[lr] = yv - (s + dr0);
    }
    wave_lds_order();
    // ---- S^T = H Z + (D R D^T)^T
    {
      f32x16 acc;
      BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = drd_t[lr * 32 + c_row(r, lane)];
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 bz[3];
        load_terms(bz, Zn, TERM, PITCH, lr, c, lk);
        const u32x4 ah[3] = {hop[0][c], hop[1][c], hop[2][c]};
        acc = mfma_bf6(ah, bz, acc);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) sc[lr * PS + c_row(r, lane)] = acc[r];
    }
    wave_lds_order();
    // ---- ONE factorization, chol(S + 1e-6): W^T (over Z's terms), c, m+, and the log-likelihood of the un-jittered S
    const float ll = chol_w_rows_bf32((lds_f*)sc, (lds_f*)sHP, (lds_f*)sv, (lds_f*)sm, (lds_f*)sm2, (lds_f*)scv, Wt, lane) + ll_pad;
    wave_lds_order();
    // ---- P+ = P- - W^T W + c c^T; filtered streams
    {
      f32x16 acc = Pacc;
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 a[3], bw[3];
        load_terms(bw, Wt, TERM, PITCH, lr, c, lk);
        BF_UNROLL for (int q = 0; q < 3; ++q) a[q] = bw[q] ^ 0x80008000u;
        acc = mfma_bf6(a, bw, acc);
      }
      const float cv = lk == 0 ? scv[lr] : 0.f;
      acc = mfma2(cv, cv, acc);
      store_tile32(out.P, bt, t, lane, acc, nr, kc);
      wave_lds_order();   // (W^T's terms are read before Y^T overwrites them below; P-'s before P+'s here)
      store_terms_transposed(Pn, TERM, PITCH, 0, 0, lane, acc);
      if (out.m.p && lane < nr) out.m.p[bt * out.m.sB + kc * out.m.sK + t * out.m.sT + lane * out.m.sE] = sm2[lane];
      if (lane == 0) {
        if constexpr (!MULTI) {
          w = reweight_single(ll, w);
          if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
        }
        if (out.ll.p) out.ll.p[bt * out.ll.sB + kc * out.ll.sK + t * out.ll.sT] = ll;   // (MULTI: the launcher always provides it)
      }
    }
    wave_lds_order();
    float fval = 0.f;
    if constexpr (DYN != 0) {   // F's row lr at the filtered mean (sm2), columns 16 c + 8 lk + e, and f_lr(m+)
      gl_cf* th = per_step(cst->dth);
      float fr[2][8];
      BF_UNROLL for (int c = 0; c < 2; ++c) BF_UNROLL for (int e = 0; e < 8; ++e) fr[c][e] = 0.f;
      if (lr < nr) {
        if constexpr (DYN == 1) {   // models.hpp: DYN_LORENZ96 (gaussfiltax/nonlinearities.py:37-50)
          const float alpha = th[0], beta = th[1], gamma = th[2], dt = th[3];
          const bool mp = th[4] != 0.f;
          const int im1 = (lr + nr - 1) % nr, ip1 = (lr + 1) % nr, im2 = (lr + 2 * nr - 2) % nr;
          const float xi = sm2[lr], ax = sm2[im1];
          const float bx = mp ? (sm2[ip1] - sm2[im2]) : 0.f;
          fval = xi + dt * (alpha * (ax * bx) - beta * xi + gamma);
          BF_UNROLL for (int c = 0; c < 2; ++c) BF_UNROLL for (int e = 0; e < 8; ++e) {
            const int j = 16 * c + 8 * lk + e;
            float v = 0.f;
            if (j == lr) v += 1.0f - dt * beta;
            if (mp) {
              if (j == im1) v += dt * alpha * bx;
              if (j == ip1) v += dt * alpha * ax;
              if (j == im2) v -= dt * alpha * ax;
            }
            fr[c][e] = v;
          }
        } else {                    // models.hpp: DYN_SINE
          const float w0 = th[0], xi = sm2[lr];
          fval = sinf(w0 * xi);
          const float d = w0 * cosf(w0 * xi);
          BF_UNROLL for (int c = 0; c < 2; ++c) BF_UNROLL for (int e = 0; e < 8; ++e) fr[c][e] = (16 * c + 8 * lk + e == lr) ? d : 0.f;
        }
      }
      BF_UNROLL for (int c = 0; c < 2; ++c) BF_UNROLL for (int d = 0; d < 4; ++d) {
        const Split3 sp = split_pair(fr[c][2 * d], fr[c][2 * d + 1]);
        aop[0][c][d] = sp.hi; aop[1][c][d] = sp.mid; aop[2][c][d] = sp.lo;
      }
    }
    // ---- Y^T = P+^T A^T; m- = A m+ + G q0 (DYN: f(m+) + F_q q0)
    {
      f32x16 acc = {0};
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 a[3];
        load_terms(a, Pn, TERM, PITCH, lr, c, lk);
        const u32x4 ba[3] = {aop[0][c], aop[1][c], aop[2][c]};
        acc = mfma_bf6(a, ba, acc);
      }
      store_terms_transposed(Yn, TERM, PITCH, 0, 0, lane, acc);
      if constexpr (DYN == 0) {
        float s = dot_terms32(aop, sm2, lk);
        s += __shfl_xor(s, 32, 64);
        sm[lr] = s + gq0;
      } else {
        sm[lr] = fval + gq0;
      }
    }
    wave_lds_order();
    // ---- P- = Y A^T + G Q G^T; predicted streams
    {
      float gq[16];
      BF_UNROLL for (int r = 0; r < 16; ++r) gq[r] = gqg_t[c_row(r, lane) * 32 + lr];
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] = 0.f;
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        u32x4 a[3];
        load_terms(a, Yn, TERM, PITCH, lr, c, lk);
        const u32x4 ba[3] = {aop[0][c], aop[1][c], aop[2][c]};
        Pacc = mfma_bf6(a, ba, Pacc);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[r] += gq[r];
      store_tile32(out.pP, bt, t, lane, Pacc, nr, kc);
      store_terms_transposed(Pn, TERM, PITCH, 0, 0, lane, Pacc);
      if (out.pm.p && lane < nr) out.pm.p[bt * out.pm.sB + kc * out.pm.sK + t * out.pm.sT + lane * out.pm.sE] = sm[lane];
    }
    wave_lds_order();
  }

  if (carry.P_out && lr < nr) BF_UNROLL for (int r = 0; r < 16; ++r) {
      const int row = c_row(r, lane);
      if (row < nr) carry.P_out[b * nr * nr + row * nr + lr] = Pacc[r];
    }
  if (carry.m_out && lane < nr) carry.m_out[b * nr + lane] = sm[lane];
  if (!MULTI && carry.w_out && lane == 0) carry.w_out[b] = w;
}


// Two chains per wave.  In kf_scan_bf32_kernel the factorization -- two thirds of a step's instructions -- works on the 32 rows of S
// in lanes 0 .. 31 while lanes 32 .. 63 repeat them.  Here the upper half-wave carries a SECOND chain (the next trajectory, or the
// next component of a Gaussian sum) through the same instruction stream: one factorization serves two chains (its column
// broadcasts by ds_swizzle instead of v_readlane, chol_w_rows_impl<.., DUAL>), the matrix-core phases run once per chain on that
// chain's LDS block (33.2 KB per wave, 128-thread workgroups, two per CU: one wave per SIMD with the 512-register budget), and the
// two chains' independent product phases interleave in the one wave's issue slots.
template <bool MULTI, bool TV, int DYN = 0>
__global__ void __launch_bounds__(128, 1)
kf_scan_bf32x2_kernel(const Bf32Const* __restrict__ cst, CView y, CarryView carry, OutViews out, long long B, long long T, int nr, int mr,
                      int K, const float* __restrict__ tvq, const float* __restrict__ tvr) {
  constexpr int PITCH = 80, TERM = 32 * PITCH, PS = 33;
  const int lane = threadIdx.x & 63;
  const int lr = lane & 31, lk = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long b0 = ((long long)blockIdx.x * 2 + wv) * 2;
  if (b0 >= B) return;   // (no workgroup barrier anywhere below)
  long long b[2], bt[2];
  int kc[2];
  bool ok[2];
  BF_UNROLL for (int c = 0; c < 2; ++c) {
    ok[c] = b0 + c < B;
    b[c] = ok[c] ? b0 + c : b0;            // an odd tail: the second slot shadows the first and stores nothing
    bt[c] = MULTI ? b[c] / K : b[c];
    kc[c] = MULTI ? (int)(b[c] % K) : 0;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_c* L0 = (lds_c*)reinterpret_cast<char*>(lds) + wv * (2 * BF32_WAVE_LDS);
  auto Pn = [&](int c) { return L0 + c * BF32_WAVE_LDS; };                        // [3][32][80 B]  P- / P+, transposed terms
  auto Zn = [&](int c) { return L0 + c * BF32_WAVE_LDS + BF32_PN_BYTES; };        // Z = (H P-)^T; later W^T, then Y^T
  auto sHP = [&](int c) { return reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + wv * (2 * BF32_WAVE_LDS) + c * BF32_WAVE_LDS); };
  auto sc = [&](int c) { return sHP(c) + 32 * PS; };
  auto sm = [&](int c) { return reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + wv * (2 * BF32_WAVE_LDS) + c * BF32_WAVE_LDS + BF32_PN_BYTES + 3 * TERM); };
  auto sm2 = [&](int c) { return sm(c) + 32; };
  auto sv = [&](int c) { return sm(c) + 64; };
  auto scv = [&](int c) { return sm(c) + 96; };

  u32x4 hop[3][2], aop[2][3][2];       // row lr of H, and of A (DYN: of each chain's F) as bf16 terms
  BF_UNROLL for (int q = 0; q < 3; ++q) BF_UNROLL for (int c = 0; c < 2; ++c) {
    hop[q][c] = *reinterpret_cast<const u32x4*>(&cst->H3[q][lr * 32 + 16 * c + 8 * lk]);
    if constexpr (DYN == 0) aop[0][q][c] = aop[1][q][c] = *reinterpret_cast<const u32x4*>(&cst->A3[q][lr * 32 + 16 * c + 8 * lk]);
  }
  const float dr0 = cst->Dr0[lr], gq0 = cst->Gq0[lr];
  f32x16 Pacc[2];
  float w[2], ynext[2];
  BF_UNROLL for (int c = 0; c < 2; ++c) {
    BF_UNROLL for (int r = 0; r < 16; ++r) {
      const int row = c_row(r, lane);
      Pacc[c][r] = (lr < nr && row < nr) ? carry.P_in[b[c] * nr * nr + row * nr + lr] : 0.f;
    }
    store_terms_transposed(Pn(c), TERM, PITCH, 0, 0, lane, Pacc[c]);
    sm(c)[lr] = lr < nr ? carry.m_in[b[c] * nr + lr] : 0.f;
    w[c] = (!MULTI && carry.w_in) ? carry.w_in[b[c]] : 1.0f;
    ynext[c] = lr < mr ? y.p[bt[c] * y.sB + lr * y.sE] : 0.f;
  }
  const float ll_pad = 0.5f * 1.8378770664093453f * (float)(32 - mr);
  wave_lds_order();

  for (long long t = 0; t < T; ++t) {
    float yv[2];
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      yv[c] = ynext[c];
      const long long tn = t + 1 < T ? t + 1 : t;
      if (lr < mr) ynext[c] = y.p[bt[c] * y.sB + tn * y.sT + lr * y.sE];
    }
    gl_cf* const drd_t = TV && tvr ? per_step(tvr + t * 1024) : per_step(cst->DRD);
    gl_cf* const gqg_t = TV && tvq ? per_step(tvq + t * 1024) : per_step(cst->GQG);
    // ---- Z = P-^T H^T; H P- in fp32 for the forward substitution; innovation
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      f32x16 z = {0};
      BF_UNROLL for (int cc = 0; cc < 2; ++cc) {
        u32x4 a[3];
        load_terms(a, Pn(c), TERM, PITCH, lr, cc, lk);
        const u32x4 bh[3] = {hop[0][cc], hop[1][cc], hop[2][cc]};
        z = mfma_bf6(a, bh, z);
      }
      wave_lds_order();   // P-'s terms have been read: their buffer now takes H P (fp32) and, below, S
      BF_UNROLL for (int r = 0; r < 16; ++r) sHP(c)[lr * PS + c_row(r, lane)] = z[r];
      store_terms_transposed(Zn(c), TERM, PITCH, 0, 0, lane, z);
      float s = dot_terms32(hop, sm(c), lk);
      s += __shfl_xor(s, 32, 64);
      sv(c)[lr] = yv[c] - (s + dr0);
    }
    wave_lds_order();
    // ---- S^T = H Z + (D R D^T)^T
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      f32x16 acc;
      BF_UNROLL for (int r = 0; r < 16; ++r) acc[r] = drd_t[lr * 32 + c_row(r, lane)];
      BF_UNROLL for (int cc = 0; cc < 2; ++cc) {
        u32x4 bz[3];
        load_terms(bz, Zn(c), TERM, PITCH, lr, cc, lk);
        const u32x4 ah[3] = {hop[0][cc], hop[1][cc], hop[2][cc]};
        acc = mfma_bf6(ah, bz, acc);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) sc(c)[lr * PS + c_row(r, lane)] = acc[r];
    }
    wave_lds_order();
    // ---- ONE factorization for both chains (chain c in half-wave c)
    const float ll2 = chol_w_rows_bf32x2((lds_f*)sc(0), (lds_f*)sHP(0), (lds_f*)sv(0), (lds_f*)sm(0), (lds_f*)sm2(0), (lds_f*)scv(0), Zn(0), lane,
                                         BF32_WAVE_LDS) + ll_pad;
    wave_lds_order();
    float llc[2];
    llc[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ll2), 0));
    llc[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ll2), 32));
    // ---- P+ = P- - W^T W + c c^T; filtered streams
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      f32x16 acc = Pacc[c];
      BF_UNROLL for (int cc = 0; cc < 2; ++cc) {
        u32x4 a[3], bw[3];
        load_terms(bw, Zn(c), TERM, PITCH, lr, cc, lk);
        BF_UNROLL for (int q = 0; q < 3; ++q) a[q] = bw[q] ^ 0x80008000u;
        acc = mfma_bf6(a, bw, acc);
      }
      const float cv = lk == 0 ? scv(c)[lr] : 0.f;
      acc = mfma2(cv, cv, acc);
      if (ok[c]) store_tile32(out.P, bt[c], t, lane, acc, nr, kc[c]);
      wave_lds_order();   // (W^T's terms are read before Y^T overwrites them below; P-'s before P+'s here)
      store_terms_transposed(Pn(c), TERM, PITCH, 0, 0, lane, acc);
      if (ok[c] && out.m.p && lane < nr) out.m.p[bt[c] * out.m.sB + kc[c] * out.m.sK + t * out.m.sT + lane * out.m.sE] = sm2(c)[lane];
      if (ok[c] && lane == 0) {
        if constexpr (!MULTI) {
          w[c] = reweight_single(llc[c], w[c]);
          if (out.w.p) out.w.p[b[c] * out.w.sB + t * out.w.sT] = w[c];
        }
        if (out.ll.p) out.ll.p[bt[c] * out.ll.sB + kc[c] * out.ll.sK + t * out.ll.sT] = llc[c];
      }
    }
    wave_lds_order();
    float fval[2] = {0.f, 0.f};
    if constexpr (DYN != 0) {   // each chain's F row lr at its filtered mean, and f_lr(m+)
      gl_cf* th = per_step(cst->dth);
      BF_UNROLL for (int c = 0; c < 2; ++c) {
        float fr[2][8];
        BF_UNROLL for (int cc = 0; cc < 2; ++cc) BF_UNROLL for (int e = 0; e < 8; ++e) fr[cc][e] = 0.f;
        if (lr < nr) {
          if constexpr (DYN == 1) {
            const float alpha = th[0], beta = th[1], gamma = th[2], dt = th[3];
            const bool mp = th[4] != 0.f;
            const int im1 = (lr + nr - 1) % nr, ip1 = (lr + 1) % nr, im2 = (lr + 2 * nr - 2) % nr;
            const float xi = sm2(c)[lr], ax = sm2(c)[im1];
            const float bx = mp ? (sm2(c)[ip1] - sm2(c)[im2]) : 0.f;
            fval[c] = xi + dt * (alpha * (ax * bx) - beta * xi + gamma);
            BF_UNROLL for (int cc = 0; cc < 2; ++cc) BF_UNROLL for (int e = 0; e < 8; ++e) {
              const int j = 16 * cc + 8 * lk + e;
              float v = 0.f;
              if (j == lr) v += 1.0f - dt * beta;
              if (mp) {
                if (j == im1) v += dt * alpha * bx;
                if (j == ip1) v += dt * alpha * ax;
                if (j == im2) v -= dt * alpha * ax;
              }
              fr[cc][e] = v;
            }
          } else {
            const float w0 = th[0], xi = sm2(c)[lr];
            fval[c] = sinf(w0 * xi);
            const float d = w0 * cosf(w0 * xi);
            BF_UNROLL for (int cc = 0; cc < 2; ++cc) BF_UNROLL for (int e = 0; e < 8; ++e) fr[cc][e] = (16 * cc + 8 * lk + e == lr) ? d : 0.f;
          }
        }
        BF_UNROLL for (int cc = 0; cc < 2; ++cc) BF_UNROLL for (int d = 0; d < 4; ++d) {
          const Split3 sp = split_pair(fr[cc][2 * d], fr[cc][2 * d + 1]);
          aop[c][0][cc][d] = sp.hi; aop[c][1][cc][d] = sp.mid; aop[c][2][cc][d] = sp.lo;
        }
      }
    }
    // ---- Y^T = P+^T A^T; m- = A m+ + G q0 (DYN: f(m+) + F_q q0)
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      f32x16 acc = {0};
      BF_UNROLL for (int cc = 0; cc < 2; ++cc) {
        u32x4 a[3];
        load_terms(a, Pn(c), TERM, PITCH, lr, cc, lk);
        const u32x4 ba[3] = {aop[c][0][cc], aop[c][1][cc], aop[c][2][cc]};
        acc = mfma_bf6(a, ba, acc);
      }
      store_terms_transposed(Zn(c), TERM, PITCH, 0, 0, lane, acc);
      if constexpr (DYN == 0) {
        float s = dot_terms32(aop[c], sm2(c), lk);
        s += __shfl_xor(s, 32, 64);
        sm(c)[lr] = s + gq0;
      } else {
        sm(c)[lr] = fval[c] + gq0;
      }
    }
    wave_lds_order();
    // ---- P- = Y A^T + G Q G^T; predicted streams
    BF_UNROLL for (int c = 0; c < 2; ++c) {
      float gq[16];
      BF_UNROLL for (int r = 0; r < 16; ++r) gq[r] = gqg_t[c_row(r, lane) * 32 + lr];
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[c][r] = 0.f;
      BF_UNROLL for (int cc = 0; cc < 2; ++cc) {
        u32x4 a[3];
        load_terms(a, Zn(c), TERM, PITCH, lr, cc, lk);
        const u32x4 ba[3] = {aop[c][0][cc], aop[c][1][cc], aop[c][2][cc]};
        Pacc[c] = mfma_bf6(a, ba, Pacc[c]);
      }
      BF_UNROLL for (int r = 0; r < 16; ++r) Pacc[c][r] += gq[r];
      if (ok[c]) store_tile32(out.pP, bt[c], t, lane, Pacc[c], nr, kc[c]);
      store_terms_transposed(Pn(c), TERM, PITCH, 0, 0, lane, Pacc[c]);
      if (ok[c] && out.pm.p && lane < nr) out.pm.p[bt[c] * out.pm.sB + kc[c] * out.pm.sK + t * out.pm.sT + lane * out.pm.sE] = sm(c)[lane];
    }
    wave_lds_order();
  }

  BF_UNROLL for (int c = 0; c < 2; ++c) {
    if (!ok[c]) continue;
    if (carry.P_out && lr < nr) BF_UNROLL for (int r = 0; r < 16; ++r) {
        const int row = c_row(r, lane);
        if (row < nr) carry.P_out[b[c] * nr * nr + row * nr + lr] = Pacc[c][r];
      }
    if (carry.m_out && lane < nr) carry.m_out[b[c] * nr + lane] = sm(c)[lane];
    if (!MULTI && carry.w_out && lane == 0) carry.w_out[b[c]] = w[c];
  }
}


// The weight recursion of the Gaussian-sum filter (inference.py:347-350) on stored per-step log-likelihoods: one wave per
// trajectory, component k in lane k (K <= 64), w_t = exp(ll_t - max ll_t) w_{t-1} / sum, the max and the sum as xor-butterflies
// over the lanes = the oracle's adjacent-pair trees (lanes beyond K carry -inf / 0, the trees' identities).
__global__ void __launch_bounds__(64)
gsf_reweight_kernel(SView ll, SView wout, const float* __restrict__ w_in, float* __restrict__ w_out, long long T, int K) {
  const long long b = blockIdx.x;
  const int lane = threadIdx.x;
  float w = lane < K ? (w_in ? w_in[b * K + lane] : 1.0f / (float)K) : 0.f;
  for (long long t = 0; t < T; ++t) {
    const float l = lane < K ? ll.p[b * ll.sB + lane * ll.sK + t * ll.sT] : -__builtin_inff();
    float mx = l;
    BF_UNROLL for (int off = 1; off < 64; off <<= 1) {
      const float o = __shfl_xor(mx, off, 64);
      mx = (mx != mx || o != o) ? __builtin_nanf("") : fmaxf(mx, o);   // jnp.max propagates NaN
    }
    const float e = lane < K ? expf(l - mx) * w : 0.f;
    float tot = e;
    BF_UNROLL for (int off = 1; off < 64; off <<= 1) tot += __shfl_xor(tot, off, 64);
    w = e / tot;
    if (lane < K && wout.p) wout.p[b * wout.sB + lane * wout.sK + t * wout.sT] = w;
  }
  if (w_out && lane < K) w_out[b * K + lane] = w;
}

// MULTI launches: the per-step log-likelihoods go to the caller's stream when there is one, else to a stream-ordered scratch
// [B][K][T] (freed by finish_multi after the weight pass)
static int begin_multi(const bf_out_desc* out, long long B, long long T, int K, hipStream_t stream, OutViews& ov, float** scratch) {
  *scratch = nullptr;
  if (!ov.ll.p) {
    BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(scratch), sizeof(float) * (size_t)B * K * T, stream));
    ov.ll = SView{*scratch, (long long)K * T, T, 1, 1};
  }
  (void)out;
  return BF_OK;
}
static int finish_multi(const OutViews& ov, const bf_carry* carry, long long B, long long T, int K, hipStream_t stream, float* scratch) {
  hipLaunchKernelGGL(gsf_reweight_kernel, dim3((unsigned)B), dim3(64), 0, stream, ov.ll, ov.w, carry->w_in, carry->w_out, T, K);
  const hipError_t le = hipGetLastError();
  if (scratch) (void)hipFreeAsync(scratch, stream);
  BF_HIP_CHECK(le);
  return BF_OK;
}

// Per-step covariance products for the matrix-core kernels, formed ON THE DEVICE: out[t] = W C_t W^T zero-padded into an
// NP x NP block (W = G, C = Q: n x dq; or W = D, C = R: m x dr), with diag_from .. NP - 1 set to 1 (the unit noise of padded
// observations).  Same association and the same k-ascending fma chains as the host code for constant covariances
// (inference.py:69,:100: (W C) W^T), so a constant table equals the constant block bit for bit.  One workgroup per step.
__global__ void __launch_bounds__(256)
tv_table_kernel(const float* __restrict__ W, const float* __restrict__ C, int rows, int d, int NP, int diag_from, float* __restrict__ out) {
  extern __shared__ float wc[];   // [rows][d]  W C_t
  const float* Ct = C + (size_t)blockIdx.x * d * d;
  float* o = out + (size_t)blockIdx.x * NP * NP;
  for (int e = threadIdx.x; e < rows * d; e += blockDim.x) {
    const int i = e / d, l = e % d;
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = fmaf(W ? W[i * d + k] : (i == k ? 1.f : 0.f), Ct[k * d + l], s);
    wc[e] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NP * NP; e += blockDim.x) {
    const int i = e / NP, j = e % NP;
    float s = 0.f;
    if (i < rows && j < rows) {
      for (int l = 0; l < d; ++l) s = fmaf(wc[i * d + l], W ? W[j * d + l] : (j == l ? 1.f : 0.f), s);
    } else if (i == j && i >= diag_from) {
      s = 1.0f;
    }
    o[e] = s;
  }
}

// d_out: a stream-ordered allocation the caller frees with hipFreeAsync after its launch
static int tv_table_on_device(const float* W_host, const float* C_host, long long T, int rows, int d, int NP, int diag_from,
                              hipStream_t stream, float** d_out) {
  const void* dW = nullptr;
  const void* dC = nullptr;
  int rc = BF_OK;
  if (W_host && (rc = device_constants(W_host, sizeof(float) * (size_t)rows * d, stream, &dW)) != BF_OK) return rc;
  if ((rc = device_constants(C_host, sizeof(float) * (size_t)T * d * d, stream, &dC)) != BF_OK) return rc;
  BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(d_out), sizeof(float) * (size_t)T * NP * NP, stream));
  hipLaunchKernelGGL(tv_table_kernel, dim3((unsigned)T), dim3(256), sizeof(float) * (size_t)rows * d, stream,
                     static_cast<const float*>(dW), static_cast<const float*>(dC), rows, d, NP, diag_from, *d_out);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

// K = 1: bf_kalman_filter_f32; K >= 1: the Gaussian-sum filter of a linear model (bf_gsf_ekf_f32), components in turn.
// dyn_kind: 0 = linear (p->A), 1 = Lorenz-96, 2 = sine with scalars dth (identity noise input: p->G == NULL, dq == n); nonlinear
// chains always run as `multi` (K >= 1).
int launch_kf_bf32(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry, const bf_out_desc* out,
                   hipStream_t stream, int K, bool multi, int dyn_kind, const float* dth, bool two_per_wave) {
  constexpr int N = 32;
  const int nr = p->n, mr = p->m, dq = p->dq, dr = p->dr;
  if (nr > N || mr > N) return set_error(BF_EUNSUPPORTED, "one-wave matrix-core Kalman kernel: n <= 32 and m <= 32");
  if (K > 64) return set_error(BF_EUNSUPPORTED, "one-wave matrix-core kernel: at most 64 components (one per lane in the weight update)");
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  Bf32Const* h = new Bf32Const();
  std::memset(h, 0, sizeof(*h));
  auto Gat = [&](int i, int k) { return p->G ? p->G[i * dq + k] : (i == k ? 1.f : 0.f); };
  auto Dat = [&](int i, int k) { return p->D ? p->D[i * dr + k] : (i == k ? 1.f : 0.f); };
  auto bf = [](float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  auto fl = [](unsigned short hbits) {
    const uint32_t u = (uint32_t)hbits << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  if (dyn_kind != 0 && (!multi || p->G || dq != nr)) { delete h; return set_error(BF_EINVAL, "nonlinear chains: identity noise input, multi launch"); }
  for (int i = 0; i < 8; ++i) h->dth[i] = (dyn_kind != 0 && dth) ? dth[i] : 0.f;
  if (dyn_kind == 0)
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nr; ++j) {
        float x = p->A[i * nr + j];
        for (int t3 = 0; t3 < 3; ++t3) { const unsigned short hb = bf(x); h->A3[t3][i * N + j] = hb; x -= fl(hb); }
      }
  for (int i = 0; i < mr; ++i)
    for (int j = 0; j < nr; ++j) {
      float x = p->H[i * nr + j];
      for (int t3 = 0; t3 < 3; ++t3) { const unsigned short hb = bf(x); h->H3[t3][i * N + j] = hb; x -= fl(hb); }
    }
  // (G Q) G^T and (D R) D^T, association of inference.py:69,:100, into a zero-padded 32 x 32 block
  std::vector<float> GQ((size_t)nr * dq), DRm((size_t)mr * dr);
  auto gqg_of = [&](const float* Q, float* dst) {
    for (int i = 0; i < nr; ++i)
      for (int l = 0; l < dq; ++l) {
        float s = 0.f;
        for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), Q[k * dq + l], s);
        GQ[i * dq + l] = s;
      }
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nr; ++j) {
        float s = 0.f;
        for (int l = 0; l < dq; ++l) s = fmaf(GQ[i * dq + l], Gat(j, l), s);
        dst[i * N + j] = s;
      }
  };
  auto drd_of = [&](const float* R, float* dst) {
    for (int i = 0; i < mr; ++i)
      for (int l = 0; l < dr; ++l) {
        float s = 0.f;
        for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), R[k * dr + l], s);
        DRm[i * dr + l] = s;
      }
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < mr; ++j) {
        float s = 0.f;
        for (int l = 0; l < dr; ++l) s = fmaf(DRm[i * dr + l], Dat(j, l), s);
        dst[i * N + j] = s;
      }
    for (int i = mr; i < N; ++i) dst[i * N + i] = 1.0f;   // padded observations: unit noise
  };
  gqg_of(p->Q, h->GQG);
  drd_of(p->R, h->DRD);
  // per-step tables (_get_params(x, 2, t), inference.py:21): T blocks of 32 x 32, formed on the device
  float *d_tvq = nullptr, *d_tvr = nullptr;
  if (p->Q_steps > 1) {
    const int rc = tv_table_on_device(p->G, p->Q, T, nr, dq, N, N, stream, &d_tvq);
    if (rc != BF_OK) { delete h; return rc; }
  }
  if (p->R_steps > 1) {
    const int rc = tv_table_on_device(p->D, p->R, T, mr, dr, N, mr, stream, &d_tvr);
    if (rc != BF_OK) { delete h; if (d_tvq) (void)hipFreeAsync(d_tvq, stream); return rc; }
  }
  for (int i = 0; i < nr; ++i) {
    float s = 0.f;
    for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), p->q0 ? p->q0[k] : 0.f, s);
    h->Gq0[i] = s;
  }
  for (int i = 0; i < mr; ++i) {
    float s = 0.f;
    for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), p->r0 ? p->r0[k] : 0.f, s);
    h->Dr0[i] = s;
  }
  const void* dv = nullptr;
  const int crc = device_constants(h, sizeof(*h), stream, &dv);
  delete h;
  if (crc != BF_OK) {
    if (d_tvq) (void)hipFreeAsync(d_tvq, stream);
    if (d_tvr) (void)hipFreeAsync(d_tvr, stream);
    return crc;
  }
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  const Bf32Const* dc = static_cast<const Bf32Const*>(dv);
  const long long chains = multi ? B * K : B;
  const dim3 grid((unsigned)((chains + 1) / 2)), block(128);
  const bool tv = d_tvq || d_tvr;
  auto free_tables = [&]() {
    if (d_tvq) (void)hipFreeAsync(d_tvq, stream);
    if (d_tvr) (void)hipFreeAsync(d_tvr, stream);
  };
  float* llscratch = nullptr;
  if (multi) {
    const int rc = begin_multi(out, B, T, K, stream, ov, &llscratch);
    if (rc != BF_OK) { free_tables(); return rc; }
  }
  if (two_per_wave) {   // kf_scan_bf32x2_kernel: two chains per wave, four per 128-thread workgroup
    const dim3 grid2((unsigned)((chains + 3) / 4));
    const int lds2 = 4 * BF32_WAVE_LDS;
    auto go = [&](auto kern) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess) (void)hipGetLastError();
      hipLaunchKernelGGL(kern, grid2, block, lds2, stream, dc, yv, cv, ov, chains, T, nr, mr, multi ? K : 1, tv ? d_tvq : nullptr, tv ? d_tvr : nullptr);
    };
    if (multi && dyn_kind == 1) { if (tv) go(kf_scan_bf32x2_kernel<true, true, 1>); else go(kf_scan_bf32x2_kernel<true, false, 1>); }
    else if (multi && dyn_kind == 2) { if (tv) go(kf_scan_bf32x2_kernel<true, true, 2>); else go(kf_scan_bf32x2_kernel<true, false, 2>); }
    else if (multi) { if (tv) go(kf_scan_bf32x2_kernel<true, true>); else go(kf_scan_bf32x2_kernel<true, false>); }
    else { if (tv) go(kf_scan_bf32x2_kernel<false, true>); else go(kf_scan_bf32x2_kernel<false, false>); }
  } else if (multi && dyn_kind == 1) {
    if (tv) hipLaunchKernelGGL((kf_scan_bf32_kernel<true, true, 1>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, d_tvq, d_tvr);
    else hipLaunchKernelGGL((kf_scan_bf32_kernel<true, false, 1>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, nullptr, nullptr);
  } else if (multi && dyn_kind == 2) {
    if (tv) hipLaunchKernelGGL((kf_scan_bf32_kernel<true, true, 2>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, d_tvq, d_tvr);
    else hipLaunchKernelGGL((kf_scan_bf32_kernel<true, false, 2>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, nullptr, nullptr);
  } else if (multi) {
    if (tv) hipLaunchKernelGGL((kf_scan_bf32_kernel<true, true>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, d_tvq, d_tvr);
    else hipLaunchKernelGGL((kf_scan_bf32_kernel<true, false>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, K, nullptr, nullptr);
  } else {
    if (tv) hipLaunchKernelGGL((kf_scan_bf32_kernel<false, true>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, 1, d_tvq, d_tvr);
    else hipLaunchKernelGGL((kf_scan_bf32_kernel<false, false>), grid, block, 2 * BF32_WAVE_LDS, stream, dc, yv, cv, ov, chains, T, nr, mr, 1, nullptr, nullptr);
  }
  const hipError_t le0 = hipGetLastError();
  free_tables();
  if (le0 != hipSuccess && llscratch) (void)hipFreeAsync(llscratch, stream);
  BF_HIP_CHECK(le0);
  if (multi) return finish_multi(ov, carry, B, T, K, stream, llscratch);
  return BF_OK;
}

// ---------------------------------------------------------------------------------------
// K = 1, multi = false: bf_kalman_filter_f32; multi: the Gaussian-sum filter of a linear model (bf_gsf_ekf_f32), components in turn.
int launch_kf_mfma(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                   const bf_out_desc* out, hipStream_t stream, int K, bool multi, int dyn_kind, const float* dth) {
  constexpr int N = 64, M = 32;
  // Smaller models ride in the (64, 32) tiles zero-padded (variant 5): A, H, G Q G^T padded with zeros keep the padded
  // block of P at exactly zero; the padded observations are y = 0 with unit noise and H rows of zero, independent of the
  // real ones up to the 1e-6 jitter's O(1e-12) coupling; each contributes log N(0; 0, 1) to the log-likelihood, taken
  // off again in the kernel.
  const int nr = p->n, mr = p->m;
  const bool padded = nr != N || mr != M;
  if (nr > N || mr > M || (padded && g_kf_mfma_variant.load() != 5))
    return set_error(BF_EUNSUPPORTED, "MFMA Kalman kernel: n <= 64 and m <= 32 (smaller than (64, 32) on variant 5 only)");
  const bool tv = p->Q_steps > 1 || p->R_steps > 1;
  if (dyn_kind != 0 && (!multi || p->G || p->dq != p->n)) return set_error(BF_EINVAL, "nonlinear chains: identity noise input, multi launch");
  if ((tv || multi) && g_kf_mfma_variant.load() != 5)
    return set_error(BF_EUNSUPPORTED, "MFMA Kalman kernel: per-step covariances and Gaussian-sum components need variant 5");
  if (K > 64) return set_error(BF_EUNSUPPORTED, "MFMA Kalman kernel: at most 64 components (one per lane in the weight update)");
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  const int dq = p->dq, dr = p->dr;
  MfmaConst<N, M>* h = new MfmaConst<N, M>();  // zero-filled: the constant cache compares contents
  auto Gat = [&](int i, int k) { return p->G ? p->G[i * dq + k] : (i == k ? 1.f : 0.f); };
  auto Dat = [&](int i, int k) { return p->D ? p->D[i * dr + k] : (i == k ? 1.f : 0.f); };
  for (int i = 0; i < 8; ++i) h->dth[i] = (dyn_kind != 0 && dth) ? dth[i] : 0.f;
  if (dyn_kind == 0) for (int i = 0; i < nr; ++i) for (int j = 0; j < nr; ++j) h->A[i * N + j] = p->A[i * nr + j];
  for (int i = 0; i < mr; ++i) for (int j = 0; j < nr; ++j) h->H[i * N + j] = p->H[i * nr + j];
  {  // x = hi + mid + lo, three bf16 terms (round to nearest even), exact for finite x
    auto bf = [](float x) {
      uint32_t u;
      std::memcpy(&u, &x, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);
      return (unsigned short)(u >> 16);
    };
    auto fl = [](unsigned short hbits) {
      const uint32_t u = (uint32_t)hbits << 16;
      float f;
      std::memcpy(&f, &u, 4);
      return f;
    };
    auto split3 = [&](const float* src, int cnt, unsigned short (*dst)[N * N], unsigned short (*dstH)[M * N]) {
      for (int i = 0; i < cnt; ++i) {
        float x = src[i];
        for (int t = 0; t < 3; ++t) {
          const unsigned short hb = bf(x);
          if (dst) dst[t][i] = hb; else dstH[t][i] = hb;
          x -= fl(hb);
        }
      }
    };
    split3(h->A, N * N, h->A3, nullptr);
    split3(h->H, M * N, nullptr, h->H3);
  }
  // (G Q) G^T and (D R) D^T, association of inference.py:69,:100, zero-padded into N x N / M x M blocks
  std::vector<float> GQv((size_t)N * dq), DRv((size_t)M * dr);
  auto gqg_of = [&](const float* Q, float* dst) {
    for (int i = 0; i < nr; ++i)
      for (int l = 0; l < dq; ++l) {
        float s = 0.f;
        for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), Q[k * dq + l], s);
        GQv[i * dq + l] = s;
      }
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nr; ++j) {
        float s = 0.f;
        for (int l = 0; l < dq; ++l) s = fmaf(GQv[i * dq + l], Gat(j, l), s);
        dst[i * N + j] = s;
      }
  };
  auto drd_of = [&](const float* R, float* dst) {
    for (int i = 0; i < mr; ++i)
      for (int l = 0; l < dr; ++l) {
        float s = 0.f;
        for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), R[k * dr + l], s);
        DRv[i * dr + l] = s;
      }
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < mr; ++j) {
        float s = 0.f;
        for (int l = 0; l < dr; ++l) s = fmaf(DRv[i * dr + l], Dat(j, l), s);
        dst[i * M + j] = s;
      }
    for (int i = mr; i < M; ++i) dst[i * M + i] = 1.0f;   // padded observations: unit noise
  };
  gqg_of(p->Q, h->GQG);
  drd_of(p->R, h->DRD);
  // per-step tables (_get_params(x, 2, t), inference.py:21), formed on the device
  float *d_tvq = nullptr, *d_tvr = nullptr;
  if (p->Q_steps > 1) {
    const int rc = tv_table_on_device(p->G, p->Q, T, nr, dq, N, N, stream, &d_tvq);
    if (rc != BF_OK) { delete h; return rc; }
  }
  if (p->R_steps > 1) {
    const int rc = tv_table_on_device(p->D, p->R, T, mr, dr, M, mr, stream, &d_tvr);
    if (rc != BF_OK) { delete h; if (d_tvq) (void)hipFreeAsync(d_tvq, stream); return rc; }
  }
  for (int i = 0; i < nr; ++i) {
    float s = 0.f;
    for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), p->q0 ? p->q0[k] : 0.f, s);
    h->Gq0[i] = s;
  }
  for (int i = 0; i < mr; ++i) {
    float s = 0.f;
    for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), p->r0 ? p->r0[k] : 0.f, s);
    h->Dr0[i] = s;
  }
  const void* dv = nullptr;
  const int crc = device_constants(h, sizeof(*h), stream, &dv);
  delete h;
  if (crc != BF_OK) {
    if (d_tvq) (void)hipFreeAsync(d_tvq, stream);
    if (d_tvr) (void)hipFreeAsync(d_tvr, stream);
    return crc;
  }
  const MfmaConst<N, M>* d = static_cast<const MfmaConst<N, M>*>(dv);

  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  if (g_kf_mfma_variant.load() == 1) {
    const size_t lds_bytes = sizeof(float) * (size_t)(3 * N * (N + 1) + N * (M + 1) + 3 * M * (M + 1) + M * (N + 1) + 2 * N + 2 * M + 4);
    auto kern = kf_scan_mfma_kernel<N, M>;
    if (lds_bytes > 64 * 1024)
      BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(256), lds_bytes, stream, d, yv, cv, ov, B, T);
  } else {
    const int var = g_kf_mfma_variant.load();
    static const int rot_mode = [] { const char* e = std::getenv("BAYESFILT_MFMA_ROT"); return e ? std::atoi(e) : 1; }();
    if (var == 5) {
      const size_t lds5 = 3 * 64 * 144 + 3 * 32 * 144 + 3 * 64 * 80 + sizeof(float) * (size_t)(M * (N + 1) + 2 * M * (M + 1) + 5 * N + 2 * M);
      const long long chains = multi ? B * K : B;
      float* llscratch = nullptr;
      auto free_tables = [&]() {
        if (d_tvq) (void)hipFreeAsync(d_tvq, stream);
        if (d_tvr) (void)hipFreeAsync(d_tvr, stream);
      };
      if (multi) {
        const int rc = begin_multi(out, B, T, K, stream, ov, &llscratch);
        if (rc != BF_OK) { free_tables(); return rc; }
      }
      auto go = [&](auto kern5) {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kern5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
        if (ae != hipSuccess) return ae;
        hipLaunchKernelGGL(kern5, dim3((unsigned)chains), dim3(256), lds5, stream, d, yv, cv, ov, chains, T, rot_mode, nr, mr, K, d_tvq, d_tvr);
        return hipGetLastError();
      };
      hipError_t le;
      if (multi && dyn_kind == 1) le = tv ? go(kf_scan_mfma5_kernel<N, M, true, true, 1>) : go(kf_scan_mfma5_kernel<N, M, true, false, 1>);
      else if (multi && dyn_kind == 2) le = tv ? go(kf_scan_mfma5_kernel<N, M, true, true, 2>) : go(kf_scan_mfma5_kernel<N, M, true, false, 2>);
      else if (multi) le = tv ? go(kf_scan_mfma5_kernel<N, M, true, true>) : go(kf_scan_mfma5_kernel<N, M, true, false>);
      else le = tv ? go(kf_scan_mfma5_kernel<N, M, false, true>) : go(kf_scan_mfma5_kernel<N, M, false, false>);
      free_tables();
      if (le != hipSuccess && llscratch) (void)hipFreeAsync(llscratch, stream);
      BF_HIP_CHECK(le);
      if (multi) return finish_multi(ov, carry, B, T, K, stream, llscratch);
      return BF_OK;
    }
    const size_t lds_bytes = var == 4 ? sizeof(float) * (size_t)(2 * N * (N + 1) + M * (N + 1) + 2 * M * (M + 1) + 3 * N + M)
                                      : sizeof(float) * (size_t)(3 * N * (N + 1) + M * (N + 1) + 3 * M * (M + 1) + 3 * N + M);
    auto kern = var == 2 ? kf_scan_mfma2_kernel<N, M, 2> : var == 3 ? kf_scan_mfma2_kernel<N, M, 3> : kf_scan_mfma2_kernel<N, M, 4>;
    if (lds_bytes > 64 * 1024)
      BF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(256), lds_bytes, stream, d, yv, cv, ov, B, T, rot_mode);
  }
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
