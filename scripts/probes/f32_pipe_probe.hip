// Ad-hoc: do v_mfma_f32_32x32x2_f32 and plain f32 vector instructions of two waves on the SAME SIMD overlap, or do they
// share one fp32 datapath?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (their SIMD
// partners) issue v_fma_f32 / v_pk_fma_f32 chains.  Times: MFMA waves alone, vector waves alone, both.
// hipcc --offload-arch=gfx950 -O2 f32_pipe_probe.hip -o f32_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(512) probe(float* out, int iters, int mode, int packed) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else if (mode & 2) {
    if (!packed) {
      float c[8];
      for (int k = 0; k < 8; ++k) c[k] = threadIdx.x * 1e-3f + k;
      const float m = 1.0f + threadIdx.x * 1e-6f, a = 1e-3f;
      for (int i = 0; i < iters; ++i)    // 64 v_fma per trip = the issue cycles of 4 MFMAs (4 x 64)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 8; ++k) c[k] = __builtin_fmaf(c[k], m, a);
      for (int k = 0; k < 8; ++k) r += c[k];
    } else {
      f32x2 c[8];
      for (int k = 0; k < 8; ++k) c[k] = f32x2{threadIdx.x * 1e-3f + k, 1.0f * k};
      const f32x2 m = {1.0f + threadIdx.x * 1e-6f, 1.0f - threadIdx.x * 1e-6f}, a = {1e-3f, 2e-3f};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 8; ++k) c[k] = __builtin_elementwise_fma(c[k], m, a);
      for (int k = 0; k < 8; ++k) r += c[k].x + c[k].y;
    }
  }
  if (r == 123.456f) out[0] = r;
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int packed = 0; packed < 2; ++packed)
    for (int mode = 1; mode <= 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, iters, mode, packed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%s  mode %d (%s): %.3f ms  (%.1f ns per trip: 4 MFMA / 64 %s)\n", packed ? "packed" : "scalar", mode,
             mode == 1 ? "MFMA waves only" : mode == 2 ? "vector waves only" : "both", best, best * 1e6 / iters, packed ? "v_pk_fma" : "v_fma");
    }
  return 0;
}
