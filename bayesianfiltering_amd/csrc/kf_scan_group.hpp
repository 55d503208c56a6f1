// kf_scan_group: batched Kalman filter, NL lanes per trajectory, CPL covariance COLUMNS per lane.
//
// Replaces, for linear f/h and one component, the lax.scan body of gaussian_sum_filter
// (gaussfiltax/inference.py:333-371): per step  _condition_on (:72-105)  ->  reweight
// (:347-350)  ->  _predict (:51-70), emitting the five posterior streams of :357-363.
//
// Mapping (gfx950).  NL consecutive lanes (a power of two) form a group that advances ONE
// trajectory; lane jl of the group owns the CPL = ceil(n / NL) columns jl*CPL .. jl*CPL+CPL-1
// of P and the matching entries of m.  A wave64 carries CPW = 64 / NL trajectories.
//   * column-local products (H P, A P, K S, the gain solve) need no communication;
//   * products that contract over the column index (P+ = P - (K S) K^T, P- = (A P) A^T,
//     m- = A m+) read the other lanes' registers through DPP quad_perm / ds_swizzle
//     broadcasts (lane_group.hpp) while every lane multiplies by its own rows of A / K;
//   * S = (H P) H^T and h(m) = H m are group all-reduces (xor butterflies);
//   * the tiny m x m LU factorisation / Cholesky are done redundantly by every lane.
// NL trades VALU work against occupancy: NL = 1 is one trajectory per lane (no redundancy,
// 1 wave per SIMD at cfg2's 65,536 trajectories), NL = n is one column per lane (4 waves per
// SIMD at n = 4, but the solve/Cholesky/log-likelihood are repeated in every lane).  The
// kernel is instruction-issue bound, so the default is the NL with the fewest VALU
// instructions per trajectory-step that still keeps >= 2 waves per SIMD.
//
// Stores (the roofline: 172 B per trajectory-step at n=4, m=2).  EMIT_STAGED (contiguous
// reference layout [B][K][T][E]) transposes time through per-wave LDS tiles so that each
// trajectory's stream leaves the CU as 64..128-byte contiguous runs written by dwordx4 stores;
// EMIT_SCALAR handles arbitrary strides with one dword store per element (contiguous across
// trajectories for the batch-inner layout).
// Loads.  Observations are fetched one block of steps ahead by LDS-DMA (global_load_lds_dword):
// no VGPR destination, so neither the compiler nor the wave ever waits on a load inside a
// step; completion is awaited once per block with a counted s_waitcnt vmcnt(N) that leaves the
// wave's newer stores in flight (gfx9 counts loads and stores on one in-order counter).
#pragma once
#include <cstdlib>
#include <vector>
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "lane_group.hpp"
#include "scan_common.hpp"

namespace bf {

template <int N, int M>
struct KFConst {
  float A[N * N];    // F_x
  float H[M * N];    // H_x
  float GQG[N * N];  // F_q Q F_q^T
  float DRD[M * M];  // H_r R H_r^T
  float Gq0[N];      // F_q q0
  float Dr0[M];      // H_r r0
};

template <int NS, int M, int NL>
struct GroupCfg {
  static_assert((NL & (NL - 1)) == 0 && NL >= 1 && NL <= 64, "lanes per trajectory must be a power of two");
  static constexpr int CPL = (NS + NL - 1) / NL;  // columns per lane
  static constexpr int CPW = 64 / NL;             // trajectories per wave
  static constexpr int EP = NS * NS;
  // floats per staged row: 128-byte rows (or the next multiple of a step that is also a multiple of 16 bytes) for the
  // matrix streams, and never fewer than 64 float4 chunks per wave tile where that keeps a flush whole store instructions
  static constexpr int lcm4(int e) { return e % 4 == 0 ? e : (e % 2 == 0 ? 2 * e : 4 * e); }
  static constexpr int row_floats(int e, int want) { return ((want + lcm4(e) - 1) / lcm4(e)) * lcm4(e); }
  static constexpr int WMIN = 4 * NL;
  static constexpr int WP = row_floats(EP, (EP >= 32 ? EP : 32) > WMIN ? (EP >= 32 ? EP : 32) : WMIN);
  // (BF_KF_ROW_FLOATS: 128-byte rows for the mean / weight streams too when the occupancy target leaves the LDS
  // for it -- NL >= 2 runs 8 waves per CU; a 64-byte run is half a cache line and costs ~15 % of the HBM
  // write rate in scripts/store_pattern_bench.hip)
#ifndef BF_KF_WSM
#define BF_KF_WSM 32
#endif
  static constexpr int WSM = (NL >= 2 && NS <= 4) ? BF_KF_WSM : 16;
  static constexpr int WM = row_floats(NS, (NS >= WSM ? NS : WSM) > WMIN ? (NS >= WSM ? NS : WSM) : WMIN);
  static constexpr int WW = WSM > WMIN ? WSM : WMIN;
  using TP = Tile<EP, WP, CPW, 4>;
  using TM = Tile<NS, WM, CPW, 4>;
  using TW = Tile<1, WW, CPW, (NL == 1 ? 0 : 4)>;  // NL = 1 fills the 160 KiB exactly without the pad
  // observation blocks: YS steps (>= 8 floats per trajectory) fetched by LDS-DMA one block ahead
  static constexpr int YS = (M >= 8) ? 1 : 8 / M;
  static constexpr int YW = YS * M;               // floats per trajectory per block
  static constexpr int YTILE = CPW * YW;          // floats per block tile (DMA writes it in lane order)
  static constexpr int YDMA = (YTILE + 63) / 64;  // LDS-DMA instructions per block
  static constexpr int YBUF = YDMA * 64;          // floats reserved per buffer
  // n = NL * CPL: every lane owns real columns; otherwise (n = 3, 5, 6, 7 with power-of-two lane groups) the padding
  // lanes skip their LDS writes
  static constexpr bool FULL_COLS = (NS == NL * CPL);
  static constexpr bool STAGED_OK = TP::GOK && TM::GOK && TW::GOK;
};

template <int NS, int M, int NL, int MODE, bool TV>
__global__ void __launch_bounds__(256, (NL >= 4 ? 4 : NL))
kf_scan_group_kernel(KFConst<NS, M> c, const float* __restrict__ gqg_t, const float* __restrict__ drd_t, CView y,
                     CarryView carry, OutViews out, long long B, long long T, int lds_per_wave, int vm_younger, int wscalar) {
  using Cfg = GroupCfg<NS, M, NL>;
  constexpr int CPL = Cfg::CPL, CPW = Cfg::CPW, EP = Cfg::EP, YS = Cfg::YS;
  using TP = typename Cfg::TP;
  using TM = typename Cfg::TM;
  using TW = typename Cfg::TW;

  const int lane = threadIdx.x & 63;
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long gwave = (long long)blockIdx.x * 4 + wave_in_blk;
  const long long b0w = gwave * CPW;  // first trajectory of this wave
  if (b0w >= B) return;               // whole wave out of range (uniform)
  const int g = lane / NL;
  const int jl = lane % NL;
  const long long b_raw = b0w + g;
  const bool chain_ok = (MODE == EMIT_STAGED) ? true : (b_raw < B);  // staged launches hold full waves only
  const long long b = chain_ok ? b_raw : B - 1;

  // ---- per-lane constants for the CPL owned columns: rows of A, columns of H and of G Q G^T.
  // Columns past n (padding lanes when NL*CPL > n) carry zeros and stay zero.
  bool col_ok[CPL];
  float Arow[CPL][NS], Hcol[CPL][M], Gcol[CPL][NS], gq0[CPL];
  BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
    col_ok[cc] = (jl * CPL + cc) < NS;
    BF_UNROLL for (int l = 0; l < NS; ++l)
        Arow[cc][l] = col_ok[cc] ? pick<NL>(c.A, NS * NS, jl, CPL * NS, cc * NS + l) : 0.f;
    BF_UNROLL for (int a = 0; a < M; ++a) Hcol[cc][a] = col_ok[cc] ? pick<NL>(c.H, M * NS, jl, CPL, a * NS + cc) : 0.f;
    BF_UNROLL for (int i = 0; i < NS; ++i) Gcol[cc][i] = col_ok[cc] ? pick<NL>(c.GQG, NS * NS, jl, CPL, i * NS + cc) : 0.f;
    gq0[cc] = col_ok[cc] ? pick<NL>(c.Gq0, NS, jl, CPL, cc) : 0.f;
  }

  // ---- state: Pc[cc][i] = P[i][jl*CPL + cc], mj[cc] = m[jl*CPL + cc]
  float Pc[CPL][NS], mj[CPL], w;
  BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
    const int col = col_ok[cc] ? jl * CPL + cc : 0;
    BF_UNROLL for (int i = 0; i < NS; ++i) Pc[cc][i] = col_ok[cc] ? carry.P_in[b * EP + i * NS + col] : 0.f;
    mj[cc] = col_ok[cc] ? carry.m_in[b * NS + col] : 0.f;
  }
  w = carry.w_in ? carry.w_in[b] : 1.0f;

  // ---- LDS carve (dynamic; only enabled streams take space).  Tiles are addressed as
  // lds + integer offset so that every access stays a DS instruction; a disabled stream
  // aliases the observation tile (always large enough to be read from).
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int q = wave_in_blk * lds_per_wave;
  const int oY = q;
  q += 2 * Cfg::YBUF;
  int oP = oY, opP = oY, oM = oY, opM = oY, oW = oY, oL = oY;
  if constexpr (MODE == EMIT_STAGED) {
    if (out.P.p) { oP = q; q += TP::FLOATS; }
    if (out.pP.p) { opP = q; q += TP::FLOATS; }
    if (out.m.p) { oM = q; q += TM::FLOATS; }
    if (out.pm.p) { opM = q; q += TM::FLOATS; }
    if (out.w.p && !wscalar) { oW = q; q += TW::FLOATS; }
    if (out.ll.p && !wscalar) { oL = q; q += TW::FLOATS; }
  }
  const unsigned offP = TP::lane_off(lane, T * EP);
  const unsigned offM = TM::lane_off(lane, T * NS);
  const unsigned offW = TW::lane_off(lane, T);
  // per-lane LDS positions for the per-step writes
  const int putP = g * TP::PITCH + jl * CPL;
  const int putM = g * TM::PITCH + jl * CPL;
  const int putW = g * TW::PITCH;

  // ---- observation stream: LDS-DMA, block k+1 lands while block k is consumed
  const float* ysrc[Cfg::YDMA];
  BF_UNROLL for (int i = 0; i < Cfg::YDMA; ++i) {
    const int e = lane + 64 * i;
    const int ch = (e / Cfg::YW) < CPW ? (e / Cfg::YW) : CPW - 1, f = e % Cfg::YW;
    const long long bb = (b0w + ch < B) ? b0w + ch : B - 1;
    // address of (trajectory, entry f % M) at step 0; the step is added per block
    ysrc[i] = y.p + bb * y.sB + (long long)(f % M) * y.sE;
  }
  const unsigned ybase = lds_byte_addr(lds + oY);
  auto y_fetch = [&](long long tb, int buf) __attribute__((always_inline)) {
    BF_UNROLL for (int i = 0; i < Cfg::YDMA; ++i) {
      const int e = lane + 64 * i;
      const int f = e % Cfg::YW;
      long long tt = tb + f / M;  // steps past the end re-read the last valid step (value unused)
      tt = tt < T ? tt : T - 1;
      if (Cfg::YTILE % 64 == 0 || e < Cfg::YTILE)
        lds_dma_dword(ysrc[i] + tt * y.sT, ybase + (unsigned)(buf * Cfg::YBUF + 64 * i) * 4u);
    }
  };

  // ---- one filter step
  auto step = [&](long long t) __attribute__((always_inline)) {
    const int ys = (int)(t % YS);
    const int ybuf = (int)((t / YS) & 1);
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = lds[oY + ybuf * Cfg::YBUF + g * Cfg::YW + ys * M + a];

    // (values are copied, never selected through a pointer: a pointer that may address either a
    // kernel argument or a private array becomes a flat pointer and defeats register promotion)
    float gqv[CPL][NS], DRD[M * M];
    BF_UNROLL for (int i = 0; i < M * M; ++i) DRD[i] = c.DRD[i];
    if constexpr (TV) {
      if (gqg_t) {
        BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i)
            gqv[cc][i] = col_ok[cc] ? gqg_t[t * EP + i * NS + jl * CPL + cc] : 0.f;
      } else {
        BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i) gqv[cc][i] = Gcol[cc][i];
      }
      if (drd_t) BF_UNROLL for (int i = 0; i < M * M; ++i) DRD[i] = drd_t[t * M * M + i];
    }

    // ================= _condition_on (inference.py:72-105) =================
    // innovation v = y - (H m + H_r r0): group all-reduce over the state index
    float v[M];
    BF_UNROLL for (int a = 0; a < M; ++a) {
      float s = Hcol[0][a] * mj[0];
      BF_UNROLL for (int cc = 1; cc < CPL; ++cc) s = fmaf(Hcol[cc][a], mj[cc], s);
      v[a] = yv[a] - (group_sum<NL>(s) + c.Dr0[a]);
    }
    // owned columns of H_x P, laid out [a][cc] (the right-hand sides of the gain solve)
    float X[M * CPL];
    BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
      float s = c.H[a * NS] * Pc[cc][0];
      BF_UNROLL for (int i = 1; i < NS; ++i) s = fmaf(c.H[a * NS + i], Pc[cc][i], s);
      X[a * CPL + cc] = s;
    }
    // S = H_r R H_r^T + (H_x P) H_x^T
    float S[M * M];
    BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int bb = 0; bb < M; ++bb) {
      float s = X[a * CPL] * Hcol[0][bb];
      BF_UNROLL for (int cc = 1; cc < CPL; ++cc) s = fmaf(X[a * CPL + cc], Hcol[cc][bb], s);
      S[a * M + bb] = DRD[a * M + bb] + group_sum<NL>(s);
    }
    // K[col][:] = column col of solve(S + 1e-6, H_x P)       X[a][cc] = K[col(cc)][a]
    psd_solve<M, CPL>(S, X);
    // (K S)[col][:]
    float KS[CPL][M];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int bb = 0; bb < M; ++bb) {
      float s = X[cc] * S[bb];
      BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * CPL + cc], S[a * M + bb], s);
      KS[cc][bb] = s;
    }
    // P+[i][col] = P[i][col] - sum_b (K S)[i][b] K[col][b]   ((K S)[i][:] lives in lane i / CPL)
    static_for<0, NS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      float ks_i[M];
      BF_UNROLL for (int bb = 0; bb < M; ++bb) ks_i[bb] = group_bcast<NL, i / CPL>(KS[i % CPL][bb]);
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        float s = ks_i[0] * X[cc];
        BF_UNROLL for (int bb = 1; bb < M; ++bb) s = fmaf(ks_i[bb], X[bb * CPL + cc], s);
        Pc[cc][i] -= s;
      }
    });
    // m+[col] = m[col] + K[col][:] . v
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
      float s = X[cc] * v[0];
      BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * CPL + cc], v[a], s);
      mj[cc] += s;
    }
    const float ll = mvn_logpdf_chol<M>(S, v);

    // ================= reweight, K = 1 (inference.py:347-350) =================
    w = reweight_single(ll, w);

    if constexpr (MODE == EMIT_STAGED) {
      if (out.m.p) BF_UNROLL for (int cc = 0; cc < CPL; ++cc)
          if (Cfg::FULL_COLS || col_ok[cc]) lds[oM + putM + int(t % TM::TS) * NS + cc] = mj[cc];
      if (out.P.p) {
        const int o = oP + putP + int(t % TP::TS) * EP;
        BF_UNROLL for (int i = 0; i < NS; ++i) BF_UNROLL for (int cc = 0; cc < CPL; ++cc)
            if (Cfg::FULL_COLS || col_ok[cc]) lds[o + i * NS + cc] = Pc[cc][i];
      }
      if (jl == 0) {
        if (wscalar) {  // T is not a multiple of 4: the rows of the scalar streams are not 16-byte aligned, they go out one by one
          if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
          if (out.ll.p) out.ll.p[b * out.ll.sB + t * out.ll.sT] = ll;
        } else {
          if (out.w.p) lds[oW + putW + int(t % TW::TS)] = w;
          if (out.ll.p) lds[oL + putW + int(t % TW::TS)] = ll;
        }
      }
    } else if (chain_ok) {
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) if (col_ok[cc]) {
        const int col = jl * CPL + cc;
        if (out.m.p) out.m.p[b * out.m.sB + t * out.m.sT + col * out.m.sE] = mj[cc];
        if (out.P.p) BF_UNROLL for (int i = 0; i < NS; ++i)
            out.P.p[b * out.P.sB + t * out.P.sT + (i * NS + col) * out.P.sE] = Pc[cc][i];
      }
      if (jl == 0) {
        if (out.w.p) out.w.p[b * out.w.sB + t * out.w.sT] = w;
        if (out.ll.p) out.ll.p[b * out.ll.sB + t * out.ll.sT] = ll;
      }
    }

    // ================= _predict (inference.py:51-70) =================
    // owned columns of F_x P+
    float APc[CPL][NS];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i) {
      float s = c.A[i * NS] * Pc[cc][0];
      BF_UNROLL for (int k = 1; k < NS; ++k) s = fmaf(c.A[i * NS + k], Pc[cc][k], s);
      APc[cc][i] = s;
    }
    // P-[i][col] = sum_l (F_x P+)[i][l] F_x[col][l] + (F_q Q F_q^T)[i][col]
    BF_UNROLL for (int i = 0; i < NS; ++i) {
      float acc[CPL];
      static_for<0, NS>([&](auto L) {
        constexpr int l = decltype(L)::value;
        const float ap_l = group_bcast<NL, l / CPL>(APc[l % CPL][i]);
        BF_UNROLL for (int cc = 0; cc < CPL; ++cc) acc[cc] = (l == 0) ? ap_l * Arow[cc][0] : fmaf(ap_l, Arow[cc][l], acc[cc]);
      });
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        if constexpr (TV) Pc[cc][i] = acc[cc] + gqv[cc][i];
        else Pc[cc][i] = acc[cc] + Gcol[cc][i];
      }
    }
    // m-[col] = sum_k F_x[col][k] m+[k] + (F_q q0)[col]
    {
      float acc[CPL];
      static_for<0, NS>([&](auto Kk) {
        constexpr int k = decltype(Kk)::value;
        const float mk = group_bcast<NL, k / CPL>(mj[k % CPL]);
        BF_UNROLL for (int cc = 0; cc < CPL; ++cc) acc[cc] = (k == 0) ? Arow[cc][0] * mk : fmaf(Arow[cc][k], mk, acc[cc]);
      });
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) mj[cc] = acc[cc] + gq0[cc];
    }

    if constexpr (MODE == EMIT_STAGED) {
      if (out.pm.p) BF_UNROLL for (int cc = 0; cc < CPL; ++cc)
          if (Cfg::FULL_COLS || col_ok[cc]) lds[opM + putM + int(t % TM::TS) * NS + cc] = mj[cc];
      if (out.pP.p) {
        const int o = opP + putP + int(t % TP::TS) * EP;
        BF_UNROLL for (int i = 0; i < NS; ++i) BF_UNROLL for (int cc = 0; cc < CPL; ++cc)
            if (Cfg::FULL_COLS || col_ok[cc]) lds[o + i * NS + cc] = Pc[cc][i];
      }
    } else if (chain_ok) {
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) if (col_ok[cc]) {
        const int col = jl * CPL + cc;
        if (out.pm.p) out.pm.p[b * out.pm.sB + t * out.pm.sT + col * out.pm.sE] = mj[cc];
        if (out.pP.p) BF_UNROLL for (int i = 0; i < NS; ++i)
            out.pP.p[b * out.pP.sB + t * out.pP.sT + (i * NS + col) * out.pP.sE] = Pc[cc][i];
      }
    }
  };

  // flush the tiles whose rows completed at step t1 - 1: all LDS reads of an event are issued
  // before the first store so that one LDS round trip covers the whole event.  `last` (after the
  // final step) also flushes the incomplete rows, chunk-limited.  Both tiles of a pair are
  // read unconditionally so the staging registers never become a conditionally-initialised
  // array (which the compiler would demote to scratch).
  auto flush_all = [&](long long t1, bool last) __attribute__((always_inline)) {
    if constexpr (MODE == EMIT_STAGED) {
      const int remP = (int)(t1 % TP::TS), remM = (int)(t1 % TM::TS), remW = (int)(t1 % TW::TS);
      if (remP == 0 || last) {
        wave_lds_sync();
        const long long t0 = remP == 0 ? t1 - TP::TS : t1 - remP;
        const int lim = remP == 0 ? TP::CH : (remP * EP) / 4;
        float4 va[TP::ITER], vb[TP::ITER];
        TP::read(lds + oP, lane, va);
        TP::read(lds + opP, lane, vb);
        if (out.P.p) TP::write(va, lane, reinterpret_cast<char*>(out.P.p + b0w * out.P.sB + t0 * EP), offP, out.P.sB, lim);
        if (out.pP.p) TP::write(vb, lane, reinterpret_cast<char*>(out.pP.p + b0w * out.pP.sB + t0 * EP), offP, out.pP.sB, lim);
      }
      if (remM == 0 || last) {
        wave_lds_sync();
        const long long t0 = remM == 0 ? t1 - TM::TS : t1 - remM;
        const int lim = remM == 0 ? TM::CH : (remM * NS) / 4;
        float4 va[TM::ITER], vb[TM::ITER];
        TM::read(lds + oM, lane, va);
        TM::read(lds + opM, lane, vb);
        if (out.m.p) TM::write(va, lane, reinterpret_cast<char*>(out.m.p + b0w * out.m.sB + t0 * NS), offM, out.m.sB, lim);
        if (out.pm.p) TM::write(vb, lane, reinterpret_cast<char*>(out.pm.p + b0w * out.pm.sB + t0 * NS), offM, out.pm.sB, lim);
      }
      if (!wscalar && (remW == 0 || last)) {
        wave_lds_sync();
        const long long t0 = remW == 0 ? t1 - TW::TS : t1 - remW;
        const int lim = remW == 0 ? TW::CH : remW / 4;
        float4 va[TW::ITER], vb[TW::ITER];
        TW::read(lds + oW, lane, va);
        TW::read(lds + oL, lane, vb);
        if (out.w.p) TW::write(va, lane, reinterpret_cast<char*>(out.w.p + b0w * out.w.sB + t0), offW, out.w.sB, lim);
        if (out.ll.p) TW::write(vb, lane, reinterpret_cast<char*>(out.ll.p + b0w * out.ll.sB + t0), offW, out.ll.sB, lim);
      }
      wave_lds_sync();
    }
  };

  // ---- time loop.  Block structure: at the top of block k the DMA of block k (issued one
  // block earlier) is awaited with a counted vmcnt that leaves this wave's newer stores in
  // flight, then the DMA of block k+1 is issued.
  y_fetch(0, 0);
  const long long nblk = (T + YS - 1) / YS;
  for (long long kb = 0; kb < nblk; ++kb) {
    const long long tb = kb * YS;
    // outstanding, youngest first: [stores of the previous block (>= vm_younger of them)] [DMA of this block]
    wait_vm(kb == 0 ? 0 : vm_younger);
    if (tb + YS < T) y_fetch(tb + YS, (int)((kb + 1) & 1));
    wave_lds_sync();
    if (tb + YS <= T) {
      // whole block, unrolled: with t = kb * YS + I the tile slots (t % TS) fold to constants and
      // the compiler can overlap the tail of one step with the head of the next
      static_for<0, YS>([&](auto I) __attribute__((always_inline)) {
        const long long t = kb * YS + decltype(I)::value;
        step(t);
        if (t + 1 < T) flush_all(t + 1, false);
      });
    } else {
      for (long long t = tb; t < T; ++t) {
        step(t);
        if (t + 1 < T) flush_all(t + 1, false);
      }
    }
  }
  flush_all(T, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (chain_ok) {
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) if (col_ok[cc]) {
      const int col = jl * CPL + cc;
      if (carry.m_out) carry.m_out[b * NS + col] = mj[cc];
      if (carry.P_out) BF_UNROLL for (int i = 0; i < NS; ++i) carry.P_out[b * EP + i * NS + col] = Pc[cc][i];
    }
    if (carry.w_out && jl == 0) carry.w_out[b] = w;
  }
}

// ---------------------------------------------------------------------------------------
template <int N, int M>
static inline void fill_const(const bf_lgssm* p, KFConst<N, M>& c, const float* Qt, const float* Rt) {
  const int dq = p->dq, dr = p->dr;
  auto Gat = [&](int i, int k) { return p->G ? p->G[i * dq + k] : (i == k ? 1.f : 0.f); };
  auto Dat = [&](int i, int k) { return p->D ? p->D[i * dr + k] : (i == k ? 1.f : 0.f); };
  for (int i = 0; i < N * N; ++i) c.A[i] = p->A[i];
  for (int i = 0; i < M * N; ++i) c.H[i] = p->H[i];
  // (G @ Q) @ G^T and (D @ R) @ D^T in fp32, association as written in inference.py:69,:100
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int l = 0; l < dq; ++l) {
        float gq = 0.f;
        for (int k = 0; k < dq; ++k) gq = fmaf(Gat(i, k), Qt[k * dq + l], gq);
        s = fmaf(gq, Gat(j, l), s);
      }
      c.GQG[i * N + j] = s;
    }
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < M; ++j) {
      float s = 0.f;
      for (int l = 0; l < dr; ++l) {
        float dq_ = 0.f;
        for (int k = 0; k < dr; ++k) dq_ = fmaf(Dat(i, k), Rt[k * dr + l], dq_);
        s = fmaf(dq_, Dat(j, l), s);
      }
      c.DRD[i * M + j] = s;
    }
  for (int i = 0; i < N; ++i) {
    float s = 0.f;
    for (int k = 0; k < dq; ++k) s = fmaf(Gat(i, k), p->q0 ? p->q0[k] : 0.f, s);
    c.Gq0[i] = s;
  }
  for (int i = 0; i < M; ++i) {
    float s = 0.f;
    for (int k = 0; k < dr; ++k) s = fmaf(Dat(i, k), p->r0 ? p->r0[k] : 0.f, s);
    c.Dr0[i] = s;
  }
}

static inline bool stream_is_reference(const bf_stream& s, long long E, long long T) {
  return s.ptr == nullptr ||
         (s.sE == 1 && s.sT == E && s.sB == T * E && (reinterpret_cast<uintptr_t>(s.ptr) % 16 == 0));
}

template <int N, int M, int NL>
static inline int launch_nml(const bf_lgssm* p, const bf_cstream* y, long long B, long long T, const bf_carry* carry,
                      const bf_out_desc* out, hipStream_t stream, int force_mode) {
  using Cfg = GroupCfg<N, M, NL>;
  KFConst<N, M> c;
  fill_const<N, M>(p, c, p->Q, p->R);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};

  const bool ref_layout = stream_is_reference(out->weights, 1, T) && stream_is_reference(out->loglik, 1, T) &&
                          stream_is_reference(out->means, N, T) && stream_is_reference(out->pred_means, N, T) &&
                          stream_is_reference(out->covs, N * N, T) && stream_is_reference(out->pred_covs, N * N, T);
  // float4 stores need every enabled stream's rows (T*E floats apart) to stay 16-byte aligned
  auto row_ok = [&](const bf_stream& st, long long E) { return st.ptr == nullptr || (T * E) % 4 == 0; };
  // (the two scalar streams may fall back to dword stores on their own: T not a multiple of 4 is common)
  const bool wscalar = !(row_ok(out->weights, 1) && row_ok(out->loglik, 1));
  const bool rows_aligned = row_ok(out->means, N) && row_ok(out->pred_means, N) && row_ok(out->covs, N * N) &&
                            row_ok(out->pred_covs, N * N);
  // the flush addresses the rows of one wave through 32-bit byte offsets from a uniform base
  const bool off32_ok = (double)T * N * N * 4.0 * (Cfg::CPW + 1) < 4.0e9;
  const bool staged_ok = Cfg::STAGED_OK && ref_layout && rows_aligned && off32_ok;
  // the staging tiles of one workgroup (4 waves) must fit the 160 KiB LDS
  bool lds_ok = true;
  if constexpr (Cfg::STAGED_OK) {
    const int nP = (out->covs.ptr ? 1 : 0) + (out->pred_covs.ptr ? 1 : 0);
    const int nM = (out->means.ptr ? 1 : 0) + (out->pred_means.ptr ? 1 : 0);
    const int nW = wscalar ? 0 : (out->weights.ptr ? 1 : 0) + (out->loglik.ptr ? 1 : 0);
    const size_t per_wave = 2 * Cfg::YBUF + nP * Cfg::TP::FLOATS + nM * Cfg::TM::FLOATS + nW * Cfg::TW::FLOATS;
    lds_ok = per_wave * 4 * sizeof(float) <= 160 * 1024;
  }
  int mode = (staged_ok && lds_ok) ? EMIT_STAGED : EMIT_SCALAR;
  if (force_mode == EMIT_SCALAR || force_mode == 1) mode = EMIT_SCALAR;
  if (force_mode == EMIT_STAGED && staged_ok && !lds_ok)
    return set_error(BF_EINVAL, "staged emitter: the enabled streams need more than 160 KiB of LDS at %d lanes per trajectory", NL);
  if (force_mode == EMIT_STAGED && !staged_ok)
    return set_error(BF_EINVAL, "staged emitter needs the contiguous reference layout and 16-byte aligned rows");

  // time-varying covariances: per-step G Q_t G^T / D R_t D^T tables on the device
  // (stream-ordered upload through the constant cache: no host synchronisation, found again by content on the next call)
  const float* d_gqg = nullptr;
  const float* d_drd = nullptr;
  const bool tv = (p->Q_steps > 1) || (p->R_steps > 1);
  if (tv) {
    if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
      return set_error(BF_EINVAL, "time-varying Q/R need exactly T=%lld matrices", T);
    if (p->Q_steps > 1) {
      std::vector<float> h((size_t)T * N * N);
      for (long long t = 0; t < T; ++t) {
        KFConst<N, M> ct;
        fill_const<N, M>(p, ct, p->Q + t * p->dq * p->dq, p->R);
        for (int i = 0; i < N * N; ++i) h[t * N * N + i] = ct.GQG[i];
      }
      const void* dv = nullptr;
      const int rc = device_constants(h.data(), sizeof(float) * h.size(), stream, &dv);
      if (rc != BF_OK) return rc;
      d_gqg = static_cast<const float*>(dv);
    }
    if (p->R_steps > 1) {
      std::vector<float> h((size_t)T * M * M);
      for (long long t = 0; t < T; ++t) {
        KFConst<N, M> ct;
        fill_const<N, M>(p, ct, p->Q, p->R + t * p->dr * p->dr);
        for (int i = 0; i < M * M; ++i) h[t * M * M + i] = ct.DRD[i];
      }
      const void* dv = nullptr;
      const int rc = device_constants(h.data(), sizeof(float) * h.size(), stream, &dv);
      if (rc != BF_OK) return rc;
      d_drd = static_cast<const float*>(dv);
    }
  }

  // One launch over trajectories [b_begin, b_begin + b_count) with the given emit mode.
  auto launch = [&](int mode_, long long b_begin, long long b_count) {
    const int nP = (out->covs.ptr ? 1 : 0) + (out->pred_covs.ptr ? 1 : 0);
    const int nM = (out->means.ptr ? 1 : 0) + (out->pred_means.ptr ? 1 : 0);
    const int nW = (out->weights.ptr ? 1 : 0) + (out->loglik.ptr ? 1 : 0);
    int lds_per_wave = 2 * Cfg::YBUF;
    // VMEM operations a wave issues per observation block besides the DMA itself: the lower
    // bound the counted vmcnt of the kernel relies on
    int vm_younger;
    if (mode_ == EMIT_STAGED) {
      const int nWs = wscalar ? 0 : nW;  // scalar streams staged only when their rows are 16-byte aligned
      lds_per_wave += nP * Cfg::TP::FLOATS + nM * Cfg::TM::FLOATS + nWs * Cfg::TW::FLOATS;
      vm_younger = (Cfg::YS / Cfg::TP::TS) * nP * Cfg::TP::ITER + (Cfg::YS / Cfg::TM::TS) * nM * Cfg::TM::ITER +
                   (Cfg::YS / Cfg::TW::TS) * nWs * Cfg::TW::ITER;
      // (masked-tail tiles: no lower bound is claimed, the observation block is awaited with vmcnt(0))
      if (!(Cfg::TP::POW2 && Cfg::TM::POW2 && Cfg::TW::POW2)) vm_younger = 0;
    } else {
      vm_younger = Cfg::YS * (nM * Cfg::CPL + nP * N * Cfg::CPL + nW);
    }
    if (vm_younger > 40) vm_younger = 40;
    const size_t lds_bytes = sizeof(float) * (size_t)lds_per_wave * 4;
    auto shift = [&](SView v) { if (v.p) v.p += b_begin * v.sB; return v; };
    CView yv2{yv.p + b_begin * yv.sB, yv.sB, yv.sT, yv.sE};
    CarryView cv2{cv.w_in ? cv.w_in + b_begin : nullptr, cv.m_in + b_begin * N, cv.P_in + b_begin * N * N,
                  cv.w_out ? cv.w_out + b_begin : nullptr, cv.m_out ? cv.m_out + b_begin * N : nullptr,
                  cv.P_out ? cv.P_out + b_begin * N * N : nullptr};
    OutViews ov2{shift(ov.w), shift(ov.m), shift(ov.P), shift(ov.pm), shift(ov.pP), shift(ov.ll)};
    const long long waves = (b_count + Cfg::CPW - 1) / Cfg::CPW;
    dim3 block(256);
    dim3 grid((unsigned)((waves + 3) / 4));
#define BF_LAUNCH(MODE_, TV_)                                                                                  \
  hipLaunchKernelGGL((kf_scan_group_kernel<N, M, NL, MODE_, TV_>), grid, block, lds_bytes, stream, c, d_gqg,   \
                     d_drd, yv2, cv2, ov2, b_count, T, lds_per_wave, vm_younger, (int)wscalar)
    if (tv) {
      if (mode_ == EMIT_SCALAR) BF_LAUNCH(EMIT_SCALAR, true);
      else if constexpr (Cfg::STAGED_OK) BF_LAUNCH(EMIT_STAGED, true);
    } else {
      if (mode_ == EMIT_SCALAR) BF_LAUNCH(EMIT_SCALAR, false);
      else if constexpr (Cfg::STAGED_OK) BF_LAUNCH(EMIT_STAGED, false);
    }
#undef BF_LAUNCH
  };
  if (mode == EMIT_STAGED) {
    // the staged kernel takes whole waves only; a ragged remainder goes through the strided kernel
    const long long b_main = (B / Cfg::CPW) * Cfg::CPW;
    if (b_main > 0) launch(EMIT_STAGED, 0, b_main);
    if (b_main < B) launch(EMIT_SCALAR, b_main, B - b_main);
  } else {
    launch(EMIT_SCALAR, 0, B);
  }
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
