// collapse: moment-matched single Gaussian of a mixture, per (trajectory, step).
//   mu = sum_k w_k m_k ;  Sigma = sum_k w_k (P_k + (m_k - mu)(m_k - mu)^T)
// gaussfiltax/utils.py:10-18 (a Python loop over components in the reference) and the point
// estimate sum_k w_k m_k of docs/experiments/BOT_Experiment_script.py:101.  Reads the strided
// posterior streams the filters emit; one thread per output matrix element, loop over K.
#include "bf_common.hpp"

namespace bf {

__global__ void __launch_bounds__(256)
collapse_kernel(SView w, SView m, SView P, float* __restrict__ mean_out, float* __restrict__ cov_out, long long B,
                long long T, int K, int n) {
  const long long total = B * T * n * n;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % n);
    const int i = (int)((idx / n) % n);
    const long long t = (idx / ((long long)n * n)) % T;
    const long long b = idx / ((long long)n * n * T);
    float mui = 0.f, muj = 0.f;
    for (int k = 0; k < K; ++k) {
      const float wk = w.p[b * w.sB + k * w.sK + t * w.sT];
      mui = fmaf(wk, m.p[b * m.sB + k * m.sK + t * m.sT + i * m.sE], mui);
      muj = fmaf(wk, m.p[b * m.sB + k * m.sK + t * m.sT + j * m.sE], muj);
    }
    if (cov_out) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) {
        const float wk = w.p[b * w.sB + k * w.sK + t * w.sT];
        const float di = m.p[b * m.sB + k * m.sK + t * m.sT + i * m.sE] - mui;
        const float dj = m.p[b * m.sB + k * m.sK + t * m.sT + j * m.sE] - muj;
        const float pk = P.p ? P.p[b * P.sB + k * P.sK + t * P.sT + (i * n + j) * P.sE] : 0.f;
        s = fmaf(wk, pk + di * dj, s);
      }
      cov_out[((b * T + t) * n + i) * n + j] = s;
    }
    if (mean_out && j == 0) mean_out[(b * T + t) * n + i] = mui;
  }
}

int launch_collapse(const bf_stream* w, const bf_stream* m, const bf_stream* P, long long B, long long T, int K, int n,
                    float* mean_out, float* cov_out, hipStream_t stream) {
  const long long total = B * T * n * n;
  const long long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
  SView pv = P ? make_sview(*P) : SView{nullptr, 0, 0, 0, 0};
  hipLaunchKernelGGL(collapse_kernel, dim3(grid), dim3(256), 0, stream, make_sview(*w), make_sview(*m), pv, mean_out,
                     cov_out, B, T, K, n);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
