// Instances and dispatch of the augmented Gaussian-sum filter kernel (agsf_scan.hpp) over the compiled (n, m) table.
#include "agsf_scan.hpp"

namespace bf {

int launch_agsf_user_impl(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                          const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out,
                          int* d_leaf_idx, int variant, hipStream_t stream);  // user_model.hip

const bf_user_model* registry_jit_handle(const bf_model* p, bool hw_arith);   // user_model.hip

int launch_agsf_ekf(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, const int32_t nc[3],
                    const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out, int* d_leaf_idx,
                    int variant, hipStream_t stream) {
  if (p->user) return launch_agsf_user_impl(p, nullptr, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
#define BF_CASE(N_, M_) \
  if (p->n == N_ && p->m == M_) return launch_agsf<N_, M_>(p, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
  BF_CASE(1, 1);
  BF_CASE(2, 1);
  BF_CASE(2, 2);
  BF_CASE(3, 1);
  BF_CASE(3, 3);
  BF_CASE(4, 1);
  BF_CASE(4, 2);
  BF_CASE(8, 4);
#undef BF_CASE
  // no compiled instance for these dimensions: a linear model runs on the kernel compiled now (its Jacobians are its matrices;
  // the registry's nonlinear functions have their analytic Jacobians in the compiled instances only)
  if (p->dyn_id == DYN_LINEAR && p->emi_id == EMI_LINEAR) {
    bf_model jit = *p;
    jit.user = registry_jit_handle(p, false);
    if (!jit.user) return set_error(BF_ENOGPU, "no current device");
    return launch_agsf_user_impl(&jit, nullptr, y, u, B, T, nc, key, opt, carry, out, d_leaf_idx, variant, stream);
  }
  return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: (n=%d, m=%d) is not compiled in", p->n, p->m);
}

// stand-alone utils.optimal_resampling (utils.py:216-244): one trajectory per MP-lane segment
__global__ void __launch_bounds__(256)
optimal_resample_kernel(const float* __restrict__ w, long long B, int M, int MP, int N, uint32_t k0, uint32_t k1,
                        int* __restrict__ idx, float* __restrict__ wout) {
  const int tid = threadIdx.x;
  const int l = tid % MP;
  const long long b_raw = (long long)blockIdx.x * (256 / MP) + tid / MP;
  const long long b = b_raw < B ? b_raw : B - 1;
  const float wl = l < M ? w[b * M + l] : 0.f;
  int io;
  float wo;
  optimal_resampling_lanes(wl, l, MP, M, N, k0, k1, io, wo);
  if (b_raw < B && l < N) {
    idx[b * N + l] = io;
    wout[b * N + l] = wo;
  }
}

// more than 64 weights: one trajectory per workgroup of NW waves (optimal_resampling_block)
template <int NW>
__global__ void __launch_bounds__(64 * NW)
optimal_resample_block_kernel(const float* __restrict__ w, long long B, int M, int N, uint32_t k0, uint32_t k1, int* __restrict__ idx,
                              float* __restrict__ wout) {
  __shared__ float scratch[4 * 64 * NW];
  __shared__ float red[64];
  const int l = threadIdx.x;
  const long long b = blockIdx.x;
  const float wl = l < M ? w[b * M + l] : 0.f;
  int io;
  float wo;
  optimal_resampling_block<NW>(wl, M, N, k0, k1, scratch, red, io, wo);
  if (l < N) {
    idx[b * N + l] = io;
    wout[b * N + l] = wo;
  }
}

int launch_optimal_resample(const float* d_w, const uint32_t key[2], long long B, int M, int N, int* d_idx, float* d_wout,
                            hipStream_t stream) {
  int MP = 1;
  while (MP < M) MP <<= 1;
  if (MP > 1024) return set_error(BF_EUNSUPPORTED, "optimal resampling: %d weights exceed one workgroup (1024)", M);
  if (N < 1 || N > M) return set_error(BF_EINVAL, "optimal resampling: need 1 <= N <= M");
  if (MP > 64) {
    if (MP <= 256)
      hipLaunchKernelGGL(optimal_resample_block_kernel<4>, dim3((unsigned)B), dim3(256), 0, stream, d_w, B, M, N, key[0], key[1], d_idx,
                         d_wout);
    else
      hipLaunchKernelGGL(optimal_resample_block_kernel<16>, dim3((unsigned)B), dim3(1024), 0, stream, d_w, B, M, N, key[0], key[1],
                         d_idx, d_wout);
    BF_HIP_CHECK(hipGetLastError());
    return BF_OK;
  }
  const int tpb = 256 / MP;
  hipLaunchKernelGGL(optimal_resample_kernel, dim3((unsigned)((B + tpb - 1) / tpb)), dim3(256), 0, stream, d_w, B, M, MP, N, key[0],
                     key[1], d_idx, d_wout);
  BF_HIP_CHECK(hipGetLastError());
  return BF_OK;
}

}  // namespace bf
