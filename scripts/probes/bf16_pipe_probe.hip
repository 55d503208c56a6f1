// Ad-hoc: v_mfma_f32_32x32x16_bf16 next to fp32 vector instructions of the SIMD partner: overlap or sum?  And its rate
// against v_mfma_f32_32x32x2_f32 for the same product (K = 16: one bf16 instruction x 6 split terms | eight fp32 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) probe(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 a0 = {0}, a1 = {0};
      bf16x8 x, y;
      for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 1e-3f + i); y[i] = (__bf16)(1.0f + i * 0.25f); }
      for (int i = 0; i < iters; ++i) {   // 6 bf16 MFMAs per trip = the six split terms of one K = 16 chunk
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      }
      r = a0[0] + a1[1];
    }
  } else if (mode & 2) {
    float c[8];
    for (int k = 0; k < 8; ++k) c[k] = threadIdx.x * 1e-3f + k;
    const float m = 1.0f + threadIdx.x * 1e-6f, a = 1e-3f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = __builtin_fmaf(c[k], m, a);
    for (int k = 0; k < 8; ++k) r += c[k];
  }
  if (r == 123.456f) out[0] = r;
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 1; mode <= 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, iters, mode);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("mode %d (%s): %.3f ms  (%.1f ns per trip: 6 bf16 MFMA 32x32x16 / 64 v_fma)\n", mode,
           mode == 1 ? "bf16 MFMA waves only" : mode == 2 ? "vector waves only" : "both", best, best * 1e6 / iters);
  }
  return 0;
}
