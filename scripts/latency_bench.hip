// Single-wave latency microbenchmarks on gfx950 (dependent VALU chains, v_readlane -> VALU, LDS round trips).
// build: hipcc -O3 --offload-arch=gfx950 scripts/latency_bench.hip -o scripts/latency_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
template <int MODE>
__global__ void k(float* out, long long* ticks, int iters, float seed) {
  __shared__ float buf[256];
  float a = seed + threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-4f;
  buf[threadIdx.x] = a;
  __syncthreads();
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) a = fmaf(a, b, c);                                   // dependent FMA
      if (MODE == 1) a = fmaf(a, rdlane(a, u), c);                        // readlane of the chain value -> FMA
      if (MODE == 2) a = __builtin_amdgcn_rsqf(a) + 1.5f;                 // rsq + add
      if (MODE == 3) { buf[threadIdx.x] = a; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); a = buf[(threadIdx.x + 1) & 63] + c; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }  // LDS round trip
      if (MODE == 4) { a = fmaf(a, b, c); b = fmaf(b, b, c) ; c = fmaf(c, b, 1e-9f)*0.5f; }  // 3-4 independent-ish ops
      if (MODE == 5) a = fmaf(a, __shfl(a, u, 64), c);                    // ds_bpermute broadcast -> FMA
      if (MODE == 6) a = fmaf(a, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x140, 0xF, 0xF, true)), c);  // DPP row_mirror -> FMA
    }
  }
  long long t1 = wall_clock64();
  out[threadIdx.x] = a + b + c;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* d_out, long long* d_t) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d_out, d_t, 10, 1.0f);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d_out, d_t, iters, 1.0f);
  long long t;
  hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
  printf("%-40s %8.1f ns per op  (%6.1f cycles at 2.4 GHz)\n", name, t * 10.0 / (iters * 32.0), t * 10.0 / (iters * 32.0) * 2.4);
}
int main() {
  float* d_out; long long* d_t;
  hipMalloc(&d_out, 1024); hipMalloc(&d_t, 64);
  run<0>("dependent v_fma", d_out, d_t);
  run<1>("v_readlane(chain) -> v_fma", d_out, d_t);
  run<2>("v_rsq + v_add", d_out, d_t);
  run<3>("ds_write + ds_read + add (round trip)", d_out, d_t);
  run<4>("4 loosely dependent VALU (per group)", d_out, d_t);
  run<5>("ds_bpermute(chain) -> v_fma", d_out, d_t);
  run<6>("DPP row_mirror(chain) -> v_fma", d_out, d_t);
  return 0;
}
