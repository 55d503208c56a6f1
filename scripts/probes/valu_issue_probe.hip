// Ad-hoc: the VALU issue rate of one gfx950 SIMD as a function of resident waves and instruction kind -- the denominator of
// the particle filter's "VALU issue" roofline.  One workgroup per CU (grid = 256), 64 * 4 * W threads = W waves per SIMD;
// every wave runs `iters` trips of 64 instructions of one kind, either one dependent chain (each instruction reads the
// previous result) or 8 independent chains.  Prints cycles per wave-instruction per SIMD at the measured shader clock
// (s_memtime) and at 2.4 GHz wall time.
// hipcc --offload-arch=gfx950 -O2 valu_issue_probe.hip -o valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int CHAINS>
__global__ void __launch_bounds__(1024) probe(uint32_t* out, long long* clk, int iters) {
  uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u;
  float f[8];
  f32x2 g[8];
  for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x + k * 77u; f[k] = 1.0f + 1e-3f * (threadIdx.x + k); g[k] = f32x2{f[k], 0.5f * f[k]}; }
  const float m = 1.0f + 1e-7f * threadIdx.x, c = 1e-3f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / CHAINS; ++u)
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) {
        if constexpr (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if constexpr (KIND == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if constexpr (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[k]));
        if constexpr (KIND == 3) asm volatile("v_add3_u32 %0, %0, %1, 3" : "+v"(a[k]) : "v"(b));
        if constexpr (KIND == 4) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b));
        if constexpr (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(m), "v"(c));
        if constexpr (KIND == 6) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(f[k]) : "v"(m));
        if constexpr (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(g[k]) : "v"(m2), "v"(c2));
        if constexpr (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(m));
        if constexpr (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  uint32_t r = 0;
  for (int k = 0; k < 8; ++k) r += a[k] + (uint32_t)f[k] + (uint32_t)g[k].x;
  if (r == 0x12345678u) out[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND, int CHAINS>
void run(const char* name, uint32_t* d, long long* dclk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  printf("%-16s chains=%d :", name, CHAINS);
  for (int W : {1, 2, 3, 4}) {
    float best = 1e9f;
    long long clk = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((probe<KIND, CHAINS>), dim3(256), dim3(256 * W), 0, 0, d, dclk, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) { best = ms; hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost); }
    }
    const double inst = (double)iters * 64 * W;      // wave-instructions per SIMD
    printf("  W=%d %5.2f cyc/inst @2.4GHz (%.2f by s_memtime x%lld)", W, best * 1e-3 * 2.4e9 / inst, (double)clk / inst, 1LL);
  }
  printf("\n");
}

int main() {
  uint32_t* d; long long* dclk;
  hipMalloc(&d, 4); hipMalloc(&dclk, 8);
  run<0, 1>("v_add_u32", d, dclk);    run<0, 8>("v_add_u32", d, dclk);
  run<1, 1>("v_xor_b32", d, dclk);    run<2, 1>("v_alignbit_b32", d, dclk);  run<2, 8>("v_alignbit_b32", d, dclk);
  run<3, 1>("v_add3_u32", d, dclk);   run<4, 1>("v_xad_u32", d, dclk);       run<4, 8>("v_xad_u32", d, dclk);
  run<5, 1>("v_fma_f32", d, dclk);    run<5, 8>("v_fma_f32", d, dclk);
  run<6, 1>("v_fmaak_f32", d, dclk);  run<6, 8>("v_fmaak_f32", d, dclk);
  run<7, 1>("v_pk_fma_f32", d, dclk); run<7, 8>("v_pk_fma_f32", d, dclk);
  run<8, 8>("v_mul_f32", d, dclk);    run<9, 8>("v_cndmask_b32", d, dclk);
  return 0;
}
