// gsf_scan: batched Gaussian-sum filter = bank of K extended Kalman filters + weight update.
//
// Replaces the lax.scan body of gaussian_sum_filter (gaussfiltax/inference.py:333-371) for
// the general case (K >= 1 components, nonlinear registry f / h):
//   vmap(_condition_on) over components (:345 -> :72-105), reweight (:347-350),
//   vmap(_predict) (:353 -> :51-70), five emitted streams (:357-363).
//
// Mapping (gfx950).  One chain = one (trajectory, component) pair, advanced by NL lanes that own
// CPL = n / NL covariance columns each (same lane-group scheme as kf_scan_group.hip; the
// cross-lane products use DPP / ds_swizzle broadcasts, lane_group.hpp).  The components of one
// trajectory occupy KP * NL consecutive lanes (KP = K rounded up to a power of two; padding
// components carry zero weight and never store), so the reweight -- max and sum over K -- is a
// segmented xor-butterfly across lanes, continued through LDS when a trajectory spans more than
// one wave (KP * NL <= 256 = one workgroup).  The butterfly adds adjacent components first:
// the adjacent-pair tree the oracle's sum uses, so weights agree to the last bit of the tree.
// Each lane gathers the chain's full mean (n broadcasts), evaluates f, F_x, h, H_x of the
// registry model redundantly (models.hpp: O(n) work against O(n^3 / NL) for the covariance
// algebra), and runs the same update/predict algebra as the Kalman kernel with the Jacobians in
// VGPRs instead of kernel-argument constants.
// SPEC_L96_PICK: structure-aware instance for the Lorenz-96 dynamics with the even-state-picking
// emission (gaussfiltax/nonlinearities.py:37-50), the model of BASELINE config 3.  F_x is a
// circulant band (4 entries per row) and H_x a selection, so the dense products shrink to 4-term
// rows and register selects.  Every lane keeps its covariance columns in coordinates relative to
// its own first column (slot i = row (base + i) mod n): the band pattern is then the same
// compile-time pattern in every lane, and what a lane needs from its neighbours arrives through
// DPP group rotations (lane_group.hpp: group_rot) instead of broadcast + lane-dependent selects.
// Skipped terms are exact zeros of the dense product, so results differ from the generic
// instance only in summation order.
// Stores: EMIT_STAGED (contiguous reference layout [B][K][T][E], K a power of two) goes through
// the per-wave LDS time-transpose tiles of scan_common.hpp; everything else through strided
// dword stores.
#pragma once
#include <cstring>
#include <vector>
#include "bf_common.hpp"
#include "kf_math.hpp"
#include "lane_group.hpp"
#include "scan_common.hpp"
#include "models.hpp"

namespace bf {

template <int NS, int M, int NL>
struct GsfCfg {
  static_assert(NS % NL == 0, "lanes per chain must divide the state dimension");
  static constexpr int CPL = NS / NL;
  static constexpr int CPW = 64 / NL;  // chains per wave
  static constexpr int EP = NS * NS;
  static constexpr int WMIN = 4 * NL;
  static constexpr int WP = (EP >= 32 ? EP : 32) > WMIN ? (EP >= 32 ? EP : 32) : WMIN;
  // 64-byte rows for the mean / weight streams.  128-byte rows write ~15 % faster per byte (scripts/store_pattern_bench.hip)
  // but the five-stream tiles of a wave then come to 26 KB, ONE 256-thread workgroup per CU (a wave per SIMD), and the stores
  // of a step overlap nobody's algebra: cfg3 (n = 8, K = 32) ran compute (6.1 ms per 200 steps) plus stores (6 ms) back to
  // back.  With 64-byte rows (17.9 KB per wave, two workgroups per CU): cfg3 12.06 -> 10.4 ms (5.0 -> 5.85 TB/s); the
  // manoeuvring-target model (n = 4, K = 128 / 32 / 4 / 1, scripts/gsf_k_probe.py) 4.99 / 4.46 / 3.90 / 3.66 ->
  // 3.41 / 2.87 / 2.70 / 2.84 ms.  (The K = 1 linear Kalman kernel of the headline, whose algebra is light, keeps 128-byte
  // rows: 18.8 against 19.7 ms, kf_scan_group.hpp.)
#ifdef BF_GSF_WSM
  static constexpr int WSM = BF_GSF_WSM;
#else
  static constexpr int WSM = 16;
#endif
  static constexpr int WM = (NS >= WSM ? NS : WSM) > WMIN ? (NS >= WSM ? NS : WSM) : WMIN;
  static constexpr int WW = WSM > WMIN ? WSM : WMIN;
  using TP = Tile<EP, WP, CPW, 4>;
  using TM = Tile<NS, WM, CPW, 4>;
  using TW = Tile<1, WW, CPW, 0>;
  static constexpr bool STAGED_OK = TP::OK && TM::OK && TW::OK;
};

struct UView {
  const float* p;  // NULL: inputs = zeros((T, 1)) as inference.py:23
  long long sB, sT;
};

enum { SPEC_GENERIC = 0, SPEC_L96_PICK = 1 };
constexpr int GSF_HDR = 32;  // LDS floats ahead of the tiles: cross-wave reduction scratch [0..7] scalar, [8..23] per lane-in-group

template <int N>
constexpr int wrap_mod(int i) {
  return ((i % N) + N) % N;
}

// EXT = true adds the optional features -- COLLAPSED outputs and per-step (time-varying) noise covariances --
// as their own instances (strided stores only), so that the plain instances keep their register budget.
template <int NS, int M, int NL, int MODE, int SPEC, bool EXT = false>
#ifndef BF_GSF_NONE_WAVES
#define BF_GSF_NONE_WAVES 2
#endif
__global__ void __launch_bounds__(256, (MODE == EMIT_NONE ? BF_GSF_NONE_WAVES : 2))
gsf_scan_kernel(EkfModel<NS, M> mdl, CView y, UView uin, CarryView carry, OutViews out, long long B, long long T, int K,
                int KP, int lds_per_wave, const float* __restrict__ tv_gqg, const float* __restrict__ tv_drd, int wscalar) {
  using Cfg = GsfCfg<NS, M, NL>;
  constexpr int CPL = Cfg::CPL, CPW = Cfg::CPW, EP = Cfg::EP;
  using TP = typename Cfg::TP;
  using TM = typename Cfg::TM;
  using TW = typename Cfg::TW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_in_blk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = KP * NL;                 // lanes per trajectory (power of two, <= 256)
  const int tpb = 256 / seg;               // trajectories per workgroup
  const int slot = tid / NL;               // chain slot in the workgroup
  const int jl = tid % NL;
  const int g = lane / NL;                 // chain slot in the wave
  const int k = slot % KP;
  const long long b_raw = (long long)blockIdx.x * tpb + slot / KP;
  const bool traj_ok = b_raw < B;
  const bool comp_ok = k < K;
  const bool chain_ok = traj_ok && comp_ok;
  const long long b = traj_ok ? b_raw : B - 1;
  const int kc = comp_ok ? k : 0;          // padding components shadow component 0 (never stored)
  const long long chain = b * K + kc;      // row index of contiguous [B][K] arrays
  // first chain of this wave in the contiguous [B][K] order (staged mode: K == KP, whole waves)
  const long long chain0w = ((long long)blockIdx.x * 4 + wave_in_blk) * CPW;

  constexpr bool L96 = SPEC == SPEC_L96_PICK;
  static_assert(!L96 || (NS >= 4 && 2 * M == NS && NL <= 4), "Lorenz-96 instance: n >= 4, m = n / 2, at most 4 lanes");
  const int base = jl * CPL;
  // row of P held in slot i: the row itself, or (base + i) mod n in the relative coordinates of L96
  auto rowabs = [&](int i) __attribute__((always_inline)) {
    if constexpr (L96) {
      const int r = base + i;
      return r >= NS ? r - NS : r;
    } else {
      return i;
    }
  };
  constexpr auto wrapn = wrap_mod<NS>;

  // ---- state: Pc[cc][i] = P[rowabs(i)][jl*CPL + cc]
  float Pc[CPL][NS], mj[CPL], w;
  BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
    const int col = jl * CPL + cc;
    BF_UNROLL for (int i = 0; i < NS; ++i) Pc[cc][i] = carry.P_in[chain * EP + rowabs(i) * NS + col];
    mj[cc] = carry.m_in[chain * NS + col];
  }
  w = comp_ok ? (carry.w_in ? carry.w_in[chain] : 1.0f / (float)K) : 0.f;

  // ---- LDS: staging tiles (per wave) + reduction scratch (per workgroup)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds;  // per-wave partials of max / sum (GSF_HDR floats)
  constexpr int GQ = L96 ? EP : 0;  // L96: F_q Q F_q^T, read back by every lane in its own coordinates
  if constexpr (L96) {
    if (tid == 0) BF_UNROLL for (int i = 0; i < EP; ++i) lds[GSF_HDR + i] = mdl.GQG[i];
    lds_barrier();
  }
  int q = GSF_HDR + GQ + wave_in_blk * lds_per_wave;
  int oP = q, opP = q, oM = q, opM = q, oW = q, oL = q;
  if constexpr (MODE == EMIT_STAGED) {
    // a covariance row completes every step when TS == 1 (n*n >= 32): the filtered and the predicted
    // stream then take turns in ONE tile (filtered rows are flushed before the predict algebra)
    if (out.P.p) { oP = q; q += TP::FLOATS; }
    if (out.pP.p) {
      if (TP::TS == 1 && out.P.p) opP = oP;
      else { opP = q; q += TP::FLOATS; }
    }
    if (out.m.p) { oM = q; q += TM::FLOATS; }
    if (out.pm.p) { opM = q; q += TM::FLOATS; }
    if (out.w.p && !wscalar) { oW = q; q += TW::FLOATS; }
    if (out.ll.p && !wscalar) { oL = q; q += TW::FLOATS; }
  }
  const unsigned offP = TP::lane_off(lane, T * EP);
  const unsigned offM = TM::lane_off(lane, T * NS);
  const unsigned offW = TW::lane_off(lane, T);
  const int putP = g * TP::PITCH + jl * CPL;
  const int putM = g * TM::PITCH + jl * CPL;
  const int putW = g * TW::PITCH;

  // reduction over the K components of a trajectory; every lane of the trajectory gets the result
  auto reduce_k = [&](float v, auto op) {
    const int lim = seg < 64 ? seg : 64;
    for (int off = NL; off < lim; off <<= 1) v = op(v, __shfl_xor(v, off, 64));
    if (seg > 64) {  // uniform: the trajectory spans seg / 64 waves of this workgroup
      lds_barrier();
      if (lane == 0) red[wave_in_blk] = v;
      lds_barrier();
      const int wpt = seg / 64;
      const int w0 = (wave_in_blk / wpt) * wpt;
      // adjacent-pair tree over the waves of the trajectory
      if (wpt == 2) v = op(red[w0], red[w0 + 1]);
      else v = op(op(red[w0], red[w0 + 1]), op(red[w0 + 2], red[w0 + 3]));
    }
    return v;
  };

  // the same sum for values that differ between the NL lanes of a chain (collapsed moments): the
  // cross-wave step keeps one partial per lane-in-group
  auto reduce_k_lane = [&](float v) {
    const int lim = seg < 64 ? seg : 64;
    int off = NL;
    if (seg >= 16) {  // a row of 16 lanes lies inside one trajectory: DPP row rotations (one fused add each)
      if constexpr (NL <= 1) v += dpp_mov<0x121>(v);  // row_ror:1
      if constexpr (NL <= 2) v += dpp_mov<0x122>(v);  // row_ror:2
      if constexpr (NL <= 4) v += dpp_mov<0x124>(v);  // row_ror:4
      v += dpp_mov<0x128>(v);                         // row_ror:8
      off = 16;
    }
    for (; off < lim && off < 16; off <<= 1) v += __shfl_xor(v, off, 64);
    auto add = [](float a, float b2) { return a + b2; };
    if (lim > 16) v = xor16_combine(v, add);
    if (lim > 32) v = xor32_combine(v, add);
    if (seg > 64) {
      lds_barrier();
      if (lane < NL) red[8 + wave_in_blk * 4 + lane] = v;
      lds_barrier();
      const int wpt = seg / 64;
      const int w0 = (wave_in_blk / wpt) * wpt;
      if (wpt == 2) v = red[8 + w0 * 4 + jl] + red[8 + (w0 + 1) * 4 + jl];
      else v = (red[8 + w0 * 4 + jl] + red[8 + (w0 + 1) * 4 + jl]) + (red[8 + (w0 + 2) * 4 + jl] + red[8 + (w0 + 3) * 4 + jl]);
    }
    return v;
  };

  auto gather = [&](const float* own, float* full) {  // full[jl*CPL + cc] <- own[cc] of lane jl
    static_for<0, NS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      full[i] = group_bcast<NL, i / CPL>(own[i % CPL]);
    });
  };

  // _predict (inference.py:51-70), linearised at the current (filtered) mean; also run once before the
  // scan for the legacy classes' predict -> update order (gaussfilt.py:113-121)
  // tv_gqg / tv_drd: per-step F_q Q_t F_q^T [T][n*n] and H_r R_t H_r^T [T][m*m] when the covariances vary in
  // time (_get_params(..., 2, t), inference.py:21,:337-340); NULL = the constants in mdl
  auto predict = [&](float u0, long long tq) __attribute__((always_inline)) {
    if constexpr (L96) {
      (void)u0;
      const float alpha = mdl.dth[0], beta = mdl.dth[1], gamma = mdl.dth[2], dt = mdl.dth[3];
      const bool mp = mdl.dth[4] != 0.f;
      // xr[l] = x[(base + l) mod n]
      float xr[NS];
      static_for<0, NS>([&](auto L) {
        constexpr int l = decltype(L)::value;
        xr[l] = group_rot<NL, l / CPL>(mj[l % CPL]);
      });
      // row i of F_x (relative): d0 on the diagonal, cm1[i] at i-1, cp1[i] at i+1, -cp1[i] at i-2
      const float d0 = 1.0f - dt * beta;
      float cm1[NS], cp1[NS];
      BF_UNROLL for (int i = 0; i < NS; ++i) {
        const float bx = mp ? (xr[wrapn(i + 1)] - xr[wrapn(i - 2)]) : 0.f;
        cm1[i] = mp ? dt * alpha * bx : 0.f;
        cp1[i] = mp ? dt * alpha * xr[wrapn(i - 1)] : 0.f;
      }
      float APc[CPL][NS];
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i) {
        float s = d0 * Pc[cc][i];
        s = fmaf(-cp1[i], Pc[cc][wrapn(i - 2)], s);
        s = fmaf(cm1[i], Pc[cc][wrapn(i - 1)], s);
        s = fmaf(cp1[i], Pc[cc][wrapn(i + 1)], s);
        APc[cc][i] = s;
      }
      // column l (relative) of F_x P, row slot i in THIS lane's coordinates
      auto ap_col = [&](auto Lr, auto I) __attribute__((always_inline)) {
        constexpr int l = wrapn(decltype(Lr)::value), i = decltype(I)::value;
        constexpr int r = l / CPL;
        return group_rot<NL, r>(APc[l % CPL][wrapn(i - r * CPL)]);
      };
      static_for<0, NS>([&](auto I) {
        constexpr int i = decltype(I)::value;
        float gq[CPL];
        {
          const int o = GSF_HDR + rowabs(i) * NS + base;
          BF_UNROLL for (int cc = 0; cc < CPL; ++cc) gq[cc] = lds[o + cc];
        }
        static_for<0, CPL>([&](auto C) {
          constexpr int cc = decltype(C)::value;
          float s = d0 * APc[cc][i];
          s = fmaf(-cp1[cc], ap_col(std::integral_constant<int, cc - 2 + NS>{}, I), s);
          s = fmaf(cm1[cc], ap_col(std::integral_constant<int, cc - 1 + NS>{}, I), s);
          s = fmaf(cp1[cc], ap_col(std::integral_constant<int, cc + 1>{}, I), s);
          Pc[cc][i] = s + gq[cc];
        });
      });
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        const float ax = xr[wrapn(cc - 1)];
        const float bx = mp ? (xr[wrapn(cc + 1)] - xr[wrapn(cc - 2)]) : 0.f;
        // x + dt (alpha ax bx - beta x + gamma) without fused multiply-adds (as NumPy evaluates it), so that
        // every copy of the loop body the compiler makes (peeled, versioned) rounds it the same way:
        // chunked scans stay bit-identical
        {
#pragma clang fp contract(off)
          mj[cc] = xr[cc] + dt * (alpha * (ax * bx) - beta * xr[cc] + gamma) + pick<NL>(mdl.Gq0, NS, jl, CPL, cc);
        }
      }
      return;
    }
    float xf[NS];
    float F[NS * NS], fx[NS];
    gather(mj, xf);
    dyn_linearize<NS, M>(mdl, xf, u0, F, fx);
    float APc[CPL][NS];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i) {
      float s = F[i * NS] * Pc[cc][0];
      BF_UNROLL for (int kk = 1; kk < NS; ++kk) s = fmaf(F[i * NS + kk], Pc[cc][kk], s);
      APc[cc][i] = s;
    }
    float Pc0[CPL][NS];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int i = 0; i < NS; ++i) Pc0[cc][i] = Pc[cc][i];
    float Frow[CPL][NS];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int l = 0; l < NS; ++l)
        Frow[cc][l] = pick<NL>(F, NS * NS, jl, CPL * NS, cc * NS + l);
    BF_UNROLL for (int i = 0; i < NS; ++i) {
      float acc[CPL];
      static_for<0, NS>([&](auto L) {
        constexpr int l = decltype(L)::value;
        const float ap_l = group_bcast<NL, l / CPL>(APc[l % CPL][i]);
        BF_UNROLL for (int cc = 0; cc < CPL; ++cc) acc[cc] = (l == 0) ? ap_l * Frow[cc][0] : fmaf(ap_l, Frow[cc][l], acc[cc]);
      });
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc)
          Pc[cc][i] = acc[cc] + (mdl.cov_quirk ? Pc0[cc][i]
                                     : (EXT && tv_gqg) ? tv_gqg[tq * EP + i * NS + jl * CPL + cc]
                                                : pick<NL>(mdl.GQG, NS * NS, jl, CPL, i * NS + cc));
    }
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) mj[cc] = pick<NL>(fx, NS, jl, CPL, cc);

  };

  if (mdl.predict_first) predict(uin.p ? uin.p[b * uin.sB] : 0.f, 0);

  for (long long t = 0; t < T; ++t) {
    float yv[M];
    BF_UNROLL for (int a = 0; a < M; ++a) yv[a] = y.p[b * y.sB + t * y.sT + a * y.sE];
    const float u0 = uin.p ? uin.p[b * uin.sB + t * uin.sT] : 0.f;

    // ================= _condition_on (inference.py:72-105), linearised at the predicted mean
    float v[M], X[M * CPL], S[M * M];
    if constexpr (L96) {
      // h picks the even states: H_x P and H_x P H_x^T are rows / entries of P
      static_for<0, M>([&](auto A) {
        constexpr int a = decltype(A)::value;
        v[a] = yv[a] - (group_bcast<NL, (2 * a) / CPL>(mj[(2 * a) % CPL]) + mdl.Dr0[a]);
        static_for<0, CPL>([&](auto C) {
          constexpr int cc = decltype(C)::value;
          float x = Pc[cc][wrapn(2 * a)];
          static_for<1, NL>([&](auto Q) {
            constexpr int ql = decltype(Q)::value;
            x = (jl == ql) ? Pc[cc][wrapn(2 * a - ql * CPL)] : x;
          });
          X[a * CPL + cc] = x;
        });
        static_for<0, M>([&](auto Bb) {
          constexpr int bb = decltype(Bb)::value;
          constexpr int own = (2 * bb) / CPL;
          S[a * M + bb] = mdl.DRD[a * M + bb] + group_bcast<NL, own>(Pc[(2 * bb) % CPL][wrapn(2 * a - own * CPL)]);
        });
      });
    } else {
      float xf[NS], H[M * NS], hx[M], HrRHr[M * M];
      gather(mj, xf);
      emi_linearize<NS, M>(mdl, xf, u0, H, hx, HrRHr);
      if constexpr (EXT) if (tv_drd) BF_UNROLL for (int i = 0; i < M * M; ++i) HrRHr[i] = tv_drd[t * (M * M) + i];
      float Hcol[CPL][M];
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int a = 0; a < M; ++a)
          Hcol[cc][a] = pick<NL>(H, M * NS, jl, CPL, a * NS + cc);
      BF_UNROLL for (int a = 0; a < M; ++a) v[a] = yv[a] - hx[a];
      BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        float s = H[a * NS] * Pc[cc][0];
        BF_UNROLL for (int i = 1; i < NS; ++i) s = fmaf(H[a * NS + i], Pc[cc][i], s);
        X[a * CPL + cc] = s;
      }
      BF_UNROLL for (int a = 0; a < M; ++a) BF_UNROLL for (int bb = 0; bb < M; ++bb) {
        float s = X[a * CPL] * Hcol[0][bb];
        BF_UNROLL for (int cc = 1; cc < CPL; ++cc) s = fmaf(X[a * CPL + cc], Hcol[cc][bb], s);
        S[a * M + bb] = HrRHr[a * M + bb] + group_sum<NL>(s);
      }
    }
    psd_solve<M, CPL>(S, X, mdl.jitter);
    float KS[CPL][M];
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) BF_UNROLL for (int bb = 0; bb < M; ++bb) {
      float s = X[cc] * S[bb];
      BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * CPL + cc], S[a * M + bb], s);
      KS[cc][bb] = s;
    }
    static_for<0, NS>([&](auto I) {
      constexpr int i = decltype(I)::value;
      float ks_i[M];
      // K S of the column that is row i of this lane's slots
      BF_UNROLL for (int bb = 0; bb < M; ++bb) {
        if constexpr (L96) ks_i[bb] = group_rot<NL, i / CPL>(KS[i % CPL][bb]);
        else ks_i[bb] = group_bcast<NL, i / CPL>(KS[i % CPL][bb]);
      }
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        float s = ks_i[0] * X[cc];
        BF_UNROLL for (int bb = 1; bb < M; ++bb) s = fmaf(ks_i[bb], X[bb * CPL + cc], s);
        Pc[cc][i] -= s;
      }
    });
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
      float s = X[cc] * v[0];
      BF_UNROLL for (int a = 1; a < M; ++a) s = fmaf(X[a * CPL + cc], v[a], s);
      mj[cc] += s;
    }
    const float ll = mvn_logpdf_chol<M>(S, v);

    // ================= reweight (inference.py:347-350): lls -= max; w = exp(lls) * w; w /= sum(w)
    {
      const float llm = comp_ok ? ll : -__builtin_inff();
      // jnp.max propagates NaN: a NaN log-likelihood poisons every weight of the trajectory
      const float mx = reduce_k(llm, [](float a, float b2) { return (a != a || b2 != b2) ? __builtin_nanf("") : fmaxf(a, b2); });
      const float e = comp_ok ? expf(ll - mx) * w : 0.f;
      const float tot = reduce_k(e, [](float a, float b2) { return a + b2; });
      w = comp_ok ? e / tot : 0.f;
    }

    // ================= COLLAPSED mode: moment-matched Gaussian of the filtered mixture (utils.py:10-18):
    //   mu = sum_k w_k m_k,  Sigma = sum_k w_k (P_k + (m_k - mu)(m_k - mu)^T)   (sums in the reweight's tree order)
    if constexpr (EXT) if (out.cm.p || out.cP.p) {
      float mu[CPL], dj[CPL];
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        mu[cc] = reduce_k_lane(w * mj[cc]);
        dj[cc] = mj[cc] - mu[cc];
      }
      // Every lane of the trajectory holds every sum.  With at least n^2 / NL component slots the lane of component q keeps
      // the q-th of its group's NS * CPL entries and the whole matrix goes out in ONE store instruction per step (64 lanes x 4
      // bytes, contiguous for a contiguous stream); otherwise the lanes of component 0 store them one by one.
      const bool spread = KP >= NS * CPL;
      const bool writer = traj_ok && k == 0;
      if (out.cm.p) {
        if (spread) {
          float keep = mu[0];
          BF_UNROLL for (int cc = 1; cc < CPL; ++cc) keep = (k == cc) ? mu[cc] : keep;
          if (traj_ok && k < CPL) out.cm.p[b * out.cm.sB + t * out.cm.sT + (jl * CPL + k) * out.cm.sE] = keep;
        } else if (writer) {
          BF_UNROLL for (int cc = 0; cc < CPL; ++cc) out.cm.p[b * out.cm.sB + t * out.cm.sT + (jl * CPL + cc) * out.cm.sE] = mu[cc];
        }
      }
      // 32 entries per lane group on one wave of 32 components x 2 lanes (BASELINE configs[2]): a reduce-scatter instead of 32
      // all-reduces.  Rows of 16 lanes sum by rotation as in reduce_k_lane; then entries j, 8 + j, 16 + j, 24 + j share the two
      // cross-row steps (pair16 / pair32_reduce_scatter: the four totals land in the four rows) and the lane of component
      // q = 8 (row) + j keeps its own.  Same operands in the same order as the all-reduce: the same bits, a third of the
      // instructions.
      bool scattered = false;
      if constexpr (NS * CPL == 32 && NL == 2) if (out.cP.p && seg == 64) {
        scattered = true;
        float di[NS];
        static_for<0, NS>([&](auto I) {
          constexpr int i = decltype(I)::value;
          if constexpr (L96) di[i] = group_rot<NL, i / CPL>(dj[i % CPL]);
          else di[i] = group_bcast<NL, i / CPL>(dj[i % CPL]);
        });
        float keep = 0.f;
        static_for<0, 8>([&](auto J) {
          constexpr int j = decltype(J)::value;
          float t[4];
          static_for<0, 4>([&](auto Mq) {
            constexpr int e = 8 * decltype(Mq)::value + j, i = e / CPL, cc = e % CPL;
            float v = w * fmaf(di[i], dj[cc], Pc[cc][i]);
            v += dpp_mov<0x122>(v);  // row_ror:2
            v += dpp_mov<0x124>(v);  // row_ror:4
            v += dpp_mov<0x128>(v);  // row_ror:8
            t[decltype(Mq)::value] = v;
          });
          const float r = pair32_reduce_scatter(pair16_reduce_scatter(t[0], t[1]), pair16_reduce_scatter(t[2], t[3]));
          keep = ((k & 7) == j) ? r : keep;
        });
        if (traj_ok) out.cP.p[b * out.cP.sB + t * out.cP.sT + (rowabs(k / CPL) * NS + jl * CPL + k % CPL) * out.cP.sE] = keep;
      }
      if (out.cP.p && !scattered) {
        float keep = 0.f;
        static_for<0, NS>([&](auto I) {
          constexpr int i = decltype(I)::value;
          // deviation of the row held in slot i (relative coordinates in the Lorenz-96 instance)
          float di;
          if constexpr (L96) di = group_rot<NL, i / CPL>(dj[i % CPL]);
          else di = group_bcast<NL, i / CPL>(dj[i % CPL]);
          BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
            const float s = reduce_k_lane(w * fmaf(di, dj[cc], Pc[cc][i]));
            if (spread) keep = (k == i * CPL + cc) ? s : keep;
            else if (writer) out.cP.p[b * out.cP.sB + t * out.cP.sT + (rowabs(i) * NS + jl * CPL + cc) * out.cP.sE] = s;
          }
        });
        if (spread && traj_ok && k < NS * CPL)
          out.cP.p[b * out.cP.sB + t * out.cP.sT + (rowabs(k / CPL) * NS + jl * CPL + k % CPL) * out.cP.sE] = keep;
      }
    }

    // ---- emit filtered streams
    if constexpr (MODE == EMIT_STAGED) {
      if (out.m.p) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) lds[oM + putM + int(t % TM::TS) * NS + cc] = mj[cc];
      if (out.P.p) {
        const int o = oP + putP + int(t % TP::TS) * EP;
        BF_UNROLL for (int i = 0; i < NS; ++i) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) lds[o + rowabs(i) * NS + cc] = Pc[cc][i];
      }
      if (jl == 0) {
        if (wscalar) {  // T is not a multiple of 4: the rows of the scalar streams are not 16-byte aligned, they go out one by one
          if (out.w.p) out.w.p[b * out.w.sB + k * out.w.sK + t * out.w.sT] = w;
          if (out.ll.p) out.ll.p[b * out.ll.sB + k * out.ll.sK + t * out.ll.sT] = ll;
        } else {
          if (out.w.p) lds[oW + putW + int(t % TW::TS)] = w;
          if (out.ll.p) lds[oL + putW + int(t % TW::TS)] = ll;
        }
      }
      if constexpr (TP::TS == 1) {
        if (out.P.p) {
          wave_lds_sync();
          float4 va[TP::ITER];
          TP::read(lds + oP, lane, va);
          TP::write(va, lane, reinterpret_cast<char*>(out.P.p + chain0w * out.P.sK + t * EP), offP, out.P.sK, TP::CH);
          wave_lds_sync();
        }
      }
    } else if (MODE == EMIT_SCALAR && chain_ok) {
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        const int col = jl * CPL + cc;
        if (out.m.p) out.m.p[b * out.m.sB + k * out.m.sK + t * out.m.sT + col * out.m.sE] = mj[cc];
        if (out.P.p) BF_UNROLL for (int i = 0; i < NS; ++i)
            out.P.p[b * out.P.sB + k * out.P.sK + t * out.P.sT + (rowabs(i) * NS + col) * out.P.sE] = Pc[cc][i];
      }
      if (jl == 0) {
        if (out.w.p) out.w.p[b * out.w.sB + k * out.w.sK + t * out.w.sT] = w;
        if (out.ll.p) out.ll.p[b * out.ll.sB + k * out.ll.sK + t * out.ll.sT] = ll;
      }
    }

    // ================= _predict
    predict(u0, t);

    // ---- emit predicted streams, flush completed rows
    if constexpr (MODE == EMIT_STAGED) {
      if (out.pm.p) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) lds[opM + putM + int(t % TM::TS) * NS + cc] = mj[cc];
      if (out.pP.p) {
        const int o = opP + putP + int(t % TP::TS) * EP;
        BF_UNROLL for (int i = 0; i < NS; ++i) BF_UNROLL for (int cc = 0; cc < CPL; ++cc) lds[o + rowabs(i) * NS + cc] = Pc[cc][i];
      }
      const long long t1 = t + 1;
      const bool last = t1 == T;
      const int remP = (int)(t1 % TP::TS), remM = (int)(t1 % TM::TS), remW = (int)(t1 % TW::TS);
      if (remP == 0 || last) {
        wave_lds_sync();
        const long long t0 = remP == 0 ? t1 - TP::TS : t1 - remP;
        const int lim = remP == 0 ? TP::CH : (remP * EP) / 4;
        float4 va[TP::ITER], vb[TP::ITER];
        TP::read(lds + oP, lane, va);
        TP::read(lds + opP, lane, vb);
        if (TP::TS != 1 && out.P.p) TP::write(va, lane, reinterpret_cast<char*>(out.P.p + chain0w * out.P.sK + t0 * EP), offP, out.P.sK, lim);
        if (out.pP.p) TP::write(vb, lane, reinterpret_cast<char*>(out.pP.p + chain0w * out.pP.sK + t0 * EP), offP, out.pP.sK, lim);
      }
      if (remM == 0 || last) {
        wave_lds_sync();
        const long long t0 = remM == 0 ? t1 - TM::TS : t1 - remM;
        const int lim = remM == 0 ? TM::CH : (remM * NS) / 4;
        float4 va[TM::ITER], vb[TM::ITER];
        TM::read(lds + oM, lane, va);
        TM::read(lds + opM, lane, vb);
        if (out.m.p) TM::write(va, lane, reinterpret_cast<char*>(out.m.p + chain0w * out.m.sK + t0 * NS), offM, out.m.sK, lim);
        if (out.pm.p) TM::write(vb, lane, reinterpret_cast<char*>(out.pm.p + chain0w * out.pm.sK + t0 * NS), offM, out.pm.sK, lim);
      }
      if (!wscalar && (remW == 0 || last)) {
        wave_lds_sync();
        const long long t0 = remW == 0 ? t1 - TW::TS : t1 - remW;
        const int lim = remW == 0 ? TW::CH : remW / 4;
        float4 va[TW::ITER], vb[TW::ITER];
        TW::read(lds + oW, lane, va);
        TW::read(lds + oL, lane, vb);
        if (out.w.p) TW::write(va, lane, reinterpret_cast<char*>(out.w.p + chain0w * out.w.sK + t0), offW, out.w.sK, lim);
        if (out.ll.p) TW::write(vb, lane, reinterpret_cast<char*>(out.ll.p + chain0w * out.ll.sK + t0), offW, out.ll.sK, lim);
      }
      wave_lds_sync();
    } else if (MODE == EMIT_SCALAR && chain_ok) {
      BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
        const int col = jl * CPL + cc;
        if (out.pm.p) out.pm.p[b * out.pm.sB + k * out.pm.sK + t * out.pm.sT + col * out.pm.sE] = mj[cc];
        if (out.pP.p) BF_UNROLL for (int i = 0; i < NS; ++i)
            out.pP.p[b * out.pP.sB + k * out.pP.sK + t * out.pP.sT + (rowabs(i) * NS + col) * out.pP.sE] = Pc[cc][i];
      }
    }
  }

  if (chain_ok) {
    BF_UNROLL for (int cc = 0; cc < CPL; ++cc) {
      const int col = jl * CPL + cc;
      if (carry.m_out) carry.m_out[chain * NS + col] = mj[cc];
      if (carry.P_out) BF_UNROLL for (int i = 0; i < NS; ++i) carry.P_out[chain * EP + rowabs(i) * NS + col] = Pc[cc][i];
    }
    if (carry.w_out && jl == 0) carry.w_out[chain] = w;
  }
}

// ---------------------------------------------------------------------------------------
static inline int next_pow2_i(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// tvq / tvr: filled with the per-step F_q Q_t F_q^T / H_r R_t H_r^T when p->Q_steps / p->R_steps > 1
template <int N, int M>
static inline int fill_model(const bf_model* p, EkfModel<N, M>& e, std::vector<float>* tvq = nullptr,
                             std::vector<float>* tvr = nullptr) {
  std::memset(&e, 0, sizeof(e));
  e.dyn_id = p->dyn_id;
  e.emi_id = p->emi_id;
  e.jitter = (p->flags & BF_MODEL_NO_JITTER) ? 0.0f : 1e-6f;
  e.predict_first = (p->flags & BF_MODEL_PREDICT_FIRST) ? 1 : 0;
  e.cov_quirk = (p->flags & BF_MODEL_LEGACY_GSF_COV) ? 1 : 0;
  const int dq = p->dq, dr = p->dr;
  // F_q / H_r of every registry function are constant matrices (identity unless stated)
  float G[N * 64] = {0}, D[M * 64] = {0};
  if (dq > 64 || dr > 64) return set_error(BF_EUNSUPPORTED, "noise dimension > 64");
  for (int i = 0; i < N; ++i)
    for (int kq = 0; kq < dq; ++kq) G[i * dq + kq] = (i == kq) ? 1.f : 0.f;
  for (int i = 0; i < M; ++i)
    for (int kr = 0; kr < dr; ++kr) D[i * dr + kr] = (i == kr) ? 1.f : 0.f;
  const float* th = p->dyn_theta;
  switch (p->dyn_id) {
    case DYN_LINEAR:
      if (p->n_dyn_theta != N * N + N * dq) return set_error(BF_EINVAL, "linear dynamics: theta must hold A and G");
      for (int i = 0; i < N * N; ++i) e.A[i] = th[i];
      for (int i = 0; i < N * dq; ++i) G[i] = th[N * N + i];
      break;
    case DYN_LORENZ96:
      if (p->n_dyn_theta != 5 || dq != N) return set_error(BF_EINVAL, "lorenz96: theta = (alpha, beta, gamma, dt, mode), dq = n");
      for (int i = 0; i < 5; ++i) e.dth[i] = th[i];
      break;
    case DYN_LORENZ63:
      if (N != 3 || p->n_dyn_theta != 4 || dq != 3) return set_error(BF_EINVAL, "lorenz63: n = dq = 3, theta = (sigma, rho, beta, dt)");
      for (int i = 0; i < 4; ++i) e.dth[i] = th[i];
      break;
    case DYN_MANEUVER_BOT: {
      if (N != 4 || p->n_dyn_theta != 2 || dq != 2) return set_error(BF_EINVAL, "maneuver_bot: n = 4, dq = 2, theta = (dt, acc)");
      e.dth[0] = th[0];
      e.dth[1] = th[1];
      const float Gb[8] = {0.5f, 0, 1, 0, 0, 0.5f, 0, 1};
      for (int i = 0; i < 8; ++i) G[i] = Gb[i];
    } break;
    case DYN_SINE:
      if (p->n_dyn_theta != 1 || dq != N) return set_error(BF_EINVAL, "sine: theta = (w0), dq = n");
      e.dth[0] = th[0];
      break;
    case DYN_GROWTH:
      if (N != 1 || dq != 1) return set_error(BF_EINVAL, "growth: n = dq = 1");
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown dynamics function id %d", p->dyn_id);
  }
  th = p->emi_theta;
  switch (p->emi_id) {
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
      if (M != 1 || dr != 1 || p->n_emi_theta != 1) return set_error(BF_EINVAL, "quadratic: m = dr = 1, theta = (c)");
      e.eth[0] = th[0];
      break;
    case EMI_STOCH_VOL:
      if (M != N || dr != N || p->n_emi_theta != 3) return set_error(BF_EINVAL, "stoch_vol: m = dr = n, theta = (sigma, beta, c)");
      for (int i = 0; i < 3; ++i) e.eth[i] = th[i];
      break;
    default: return set_error(BF_EUNSUPPORTED, "unknown emission function id %d", p->emi_id);
  }
  // (F_q Q) F_q^T, (H_r R) H_r^T, F_q q0, H_r r0 in fp32 with the association of inference.py:69,:100
  auto gqg_of = [&](const float* Q, float* out) {
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        float s = 0.f;
        for (int l = 0; l < dq; ++l) {
          float gq = 0.f;
          for (int kq = 0; kq < dq; ++kq) gq = fmaf(G[i * dq + kq], Q[kq * dq + l], gq);
          s = fmaf(gq, G[j * dq + l], s);
        }
        out[i * N + j] = s;
      }
  };
  auto drd_of = [&](const float* R, float* out) {
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < M; ++j) {
        float s = 0.f;
        for (int l = 0; l < dr; ++l) {
          float d1 = 0.f;
          for (int kr = 0; kr < dr; ++kr) d1 = fmaf(D[i * dr + kr], R[kr * dr + l], d1);
          s = fmaf(d1, D[j * dr + l], s);
        }
        out[i * M + j] = s;
      }
  };
  gqg_of(p->Q, e.GQG);
  drd_of(p->R, e.DRD);
  if (p->Q_steps > 1) {
    if (!tvq) return set_error(BF_EUNSUPPORTED, "time-varying Q is not supported on this path");
    tvq->resize((size_t)p->Q_steps * N * N);
    for (int t = 0; t < p->Q_steps; ++t) gqg_of(p->Q + (size_t)t * dq * dq, tvq->data() + (size_t)t * N * N);
  }
  if (p->R_steps > 1) {
    if (!tvr) return set_error(BF_EUNSUPPORTED, "time-varying R is not supported on this path");
    if (p->emi_id == EMI_STOCH_VOL)
      return set_error(BF_EUNSUPPORTED, "time-varying R needs an emission with a constant noise Jacobian H_r");
    tvr->resize((size_t)p->R_steps * M * M);
    for (int t = 0; t < p->R_steps; ++t) drd_of(p->R + (size_t)t * dr * dr, tvr->data() + (size_t)t * M * M);
  }
  for (int i = 0; i < N; ++i) {
    float s = 0.f;
    for (int kq = 0; kq < dq; ++kq) s = fmaf(G[i * dq + kq], p->q0 ? p->q0[kq] : 0.f, s);
    e.Gq0[i] = s;
  }
  for (int i = 0; i < M; ++i) {
    float s = 0.f;
    for (int kr = 0; kr < dr; ++kr) s = fmaf(D[i * dr + kr], p->r0 ? p->r0[kr] : 0.f, s);
    e.Dr0[i] = s;
  }
  if (p->emi_id == EMI_STOCH_VOL) {
    for (int i = 0; i < M * M; ++i) e.R[i] = p->R[i];
    for (int i = 0; i < M; ++i) e.r0[i] = p->r0 ? p->r0[i] : 0.f;
  }
  return BF_OK;
}

static inline bool gsf_stream_is_reference(const bf_stream& s, long long E, long long T, long long K) {
  return s.ptr == nullptr || (s.sE == 1 && s.sT == E && s.sK == T * E && s.sB == K * T * E &&
                              (reinterpret_cast<uintptr_t>(s.ptr) % 16 == 0));
}

// Lorenz-96 dynamics (F_q = I) observed through the matrix that picks the even states with H_r = I:
// the model SPEC_L96_PICK is written for
static inline bool gsf_is_l96_pick(const bf_model* p) {
  const int n = p->n, m = p->m;
  if (p->dyn_id != DYN_LORENZ96 || p->emi_id != EMI_LINEAR || n < 4 || 2 * m != n || p->dq != n || p->dr != m) return false;
  if (p->flags & BF_MODEL_LEGACY_GSF_COV) return false;
  if (p->Q_steps > 1 || p->R_steps > 1) return false;
  if (p->n_emi_theta != m * n + m * m) return false;
  for (int a = 0; a < m; ++a)
    for (int i = 0; i < n; ++i)
      if (p->emi_theta[a * n + i] != ((i == 2 * a) ? 1.0f : 0.0f)) return false;
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b)
      if (p->emi_theta[m * n + a * m + b] != ((a == b) ? 1.0f : 0.0f)) return false;
  return true;
}

template <int N, int M, int NL, int SPEC = SPEC_GENERIC>
static inline int launch_gsf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K,
                      const bf_carry* carry, const bf_out_desc* out, hipStream_t stream, int force_mode) {
  using Cfg = GsfCfg<N, M, NL>;
  EkfModel<N, M> e;
  std::vector<float> tvq, tvr;
  int rc = fill_model<N, M>(p, e, &tvq, &tvr);
  if (rc != BF_OK) return rc;
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  if (SPEC != SPEC_GENERIC && (!tvq.empty() || !tvr.empty()))
    return set_error(BF_EUNSUPPORTED, "structure-aware instances take constant covariances");
  const int KP = next_pow2_i(K);
  if (KP * NL > 256)
    return set_error(BF_EUNSUPPORTED, "gaussian-sum filter: %d components x %d lanes exceed one workgroup (256 lanes)", K, NL);
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  UView uv{u && u->ptr ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik),
              make_sview(out->coll_mean), make_sview(out->coll_cov)};
  const bool ref_layout = gsf_stream_is_reference(out->weights, 1, T, K) && gsf_stream_is_reference(out->loglik, 1, T, K) &&
                          gsf_stream_is_reference(out->means, N, T, K) && gsf_stream_is_reference(out->pred_means, N, T, K) &&
                          gsf_stream_is_reference(out->covs, N * N, T, K) && gsf_stream_is_reference(out->pred_covs, N * N, T, K);
  auto row_ok = [&](const bf_stream& st, long long E) { return st.ptr == nullptr || (T * E) % 4 == 0; };
  // (the two scalar streams may fall back to dword stores on their own: T not a multiple of 4 is common)
  const bool wscalar = !(row_ok(out->weights, 1) && row_ok(out->loglik, 1));
  const bool rows_aligned = row_ok(out->means, N) && row_ok(out->pred_means, N) && row_ok(out->covs, N * N) &&
                            row_ok(out->pred_covs, N * N);
  const bool off32_ok = (double)T * N * N * 4.0 * (Cfg::CPW + 1) < 4.0e9;
  const int tpb = 256 / (KP * NL);
  // staged stores need every wave to hold CPW valid, consecutive chains of the [B][K] order
  const bool staged_ok = Cfg::STAGED_OK && ref_layout && rows_aligned && off32_ok && (K == KP) && (B % tpb == 0) &&
                         ((B * K) % Cfg::CPW == 0);
  const bool ext = out->coll_mean.ptr || out->coll_cov.ptr || !tvq.empty() || !tvr.empty();
  int mode = (staged_ok && !ext) ? EMIT_STAGED : EMIT_SCALAR;
  if (force_mode == EMIT_SCALAR || force_mode == 1) mode = EMIT_SCALAR;
  if (force_mode == EMIT_STAGED && (!staged_ok || ext))
    return set_error(BF_EINVAL, "staged emitter needs the contiguous reference layout, K a power of two, whole waves, constant "
                                "covariances and no collapsed streams");

  const int nP = (out->covs.ptr ? 1 : 0) + (out->pred_covs.ptr ? 1 : 0);
  const int nM = (out->means.ptr ? 1 : 0) + (out->pred_means.ptr ? 1 : 0);
  const int nW = wscalar ? 0 : (out->weights.ptr ? 1 : 0) + (out->loglik.ptr ? 1 : 0);
  int lds_per_wave = 0;
  if constexpr (Cfg::STAGED_OK)
    if (mode == EMIT_STAGED)
      lds_per_wave = ((Cfg::TP::TS == 1 && nP == 2) ? 1 : nP) * Cfg::TP::FLOATS + nM * Cfg::TM::FLOATS + nW * Cfg::TW::FLOATS;
  lds_per_wave = (lds_per_wave + 3) & ~3;
  const size_t lds_bytes = sizeof(float) * (GSF_HDR + (SPEC == SPEC_L96_PICK ? N * N : 0) + (size_t)lds_per_wave * 4);
  if (lds_bytes > 160 * 1024) return set_error(BF_EUNSUPPORTED, "staging tiles exceed the 160 KiB LDS");
  // per-step covariance products: device copies for the duration of the launch
  // (stream-ordered upload through the constant cache, const_cache.hip: no host synchronisation)
  const float *d_tvq = nullptr, *d_tvr = nullptr;
  if (!tvq.empty()) {
    const void* dv = nullptr;
    const int rc = device_constants(tvq.data(), sizeof(float) * tvq.size(), stream, &dv);
    if (rc != BF_OK) return rc;
    d_tvq = static_cast<const float*>(dv);
  }
  if (!tvr.empty()) {
    const void* dv = nullptr;
    const int rc = device_constants(tvr.data(), sizeof(float) * tvr.size(), stream, &dv);
    if (rc != BF_OK) return rc;
    d_tvr = static_cast<const float*>(dv);
  }
  dim3 block(256);
  dim3 grid((unsigned)((B + tpb - 1) / tpb));
  const bool no_streams = !out->weights.ptr && !out->means.ptr && !out->covs.ptr && !out->pred_means.ptr && !out->pred_covs.ptr &&
                          !out->loglik.ptr;
  if (ext && no_streams) {  // collapsed-only: the per-component store code is not even compiled in
    hipLaunchKernelGGL((gsf_scan_kernel<N, M, NL, EMIT_NONE, SPEC, true>), grid, block, lds_bytes, stream, e, yv, uv, cv, ov, B,
                       T, K, KP, lds_per_wave, d_tvq, d_tvr, (int)wscalar);
  } else if (ext) {
    hipLaunchKernelGGL((gsf_scan_kernel<N, M, NL, EMIT_SCALAR, SPEC, true>), grid, block, lds_bytes, stream, e, yv, uv, cv, ov, B,
                       T, K, KP, lds_per_wave, d_tvq, d_tvr, (int)wscalar);
  } else if (mode == EMIT_SCALAR) {
    hipLaunchKernelGGL((gsf_scan_kernel<N, M, NL, EMIT_SCALAR, SPEC>), grid, block, lds_bytes, stream, e, yv, uv, cv, ov, B, T, K,
                       KP, lds_per_wave, d_tvq, d_tvr, (int)wscalar);
  } else {
    if constexpr (Cfg::STAGED_OK)
      hipLaunchKernelGGL((gsf_scan_kernel<N, M, NL, EMIT_STAGED, SPEC>), grid, block, lds_bytes, stream, e, yv, uv, cv, ov, B, T, K,
                         KP, lds_per_wave, d_tvq, d_tvr, (int)wscalar);
  }
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
