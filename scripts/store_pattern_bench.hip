// Micro-benchmark: HBM write ceiling for the reference layout's store pattern.
// B chains, each owns a contiguous row of ROWBYTES; at every "event" each chain appends RUN
// bytes to its row (dwordx4 stores, RUN/16 lanes per chain).  Pure stores, no arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int RUN>
__global__ void __launch_bounds__(256) k(float4* out, long long rowbytes, int events) {
  constexpr int LPC = RUN / 16;           // lanes per chain
  constexpr int CPW = 64 / LPC;           // chains per wave
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long chain = wave * CPW + lane / LPC;
  char* p = (char*)out + chain * rowbytes + (lane % LPC) * 16;
  float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
  for (int e = 0; e < events; ++e) {
    *(float4*)(p + (long long)e * RUN) = v;
    v.x += 1.f;
    __builtin_amdgcn_s_sleep(8);
  }
}

int main(int argc, char** argv) {
  const long long B = 65536;
  const long long total = (argc > 1 ? atoll(argv[1]) : 20LL) << 30;  // GiB
  const long long rowbytes = total / B;  // 320 KiB per chain
  float4* d;
  CK(hipMalloc(&d, total));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](auto tag, const char* name) {
    constexpr int RUN = decltype(tag)::value;
    const int events = (int)(rowbytes / RUN);
    const long long waves = B / (64 / (RUN / 16));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k<RUN>, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, d, rowbytes, events);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("%s total=%lld GiB run=%4d B  waves=%6lld  %.3f ms  %.1f GB/s\n", name, total >> 30, RUN, waves, ms, total / ms / 1e6);
    }
  };
  run(std::integral_constant<int, 64>{}, "rows");
  run(std::integral_constant<int, 128>{}, "rows");
  run(std::integral_constant<int, 256>{}, "rows");
  run(std::integral_constant<int, 512>{}, "rows");
  run(std::integral_constant<int, 1024>{}, "rows");
  return 0;
}
