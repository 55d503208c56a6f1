// Ad-hoc: where does the dispatcher put the waves of 256-thread workgroups that fit two per CU?  Prints, for the first
// workgroups, (XCC, SE, CU) and per wave (SIMD, wave slot).  hipcc --offload-arch=gfx950 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wave) * 2 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    out[(blockIdx.x * 4 + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  lds[threadIdx.x] = x;
  __syncthreads();
  if (lds[(threadIdx.x + 1) & 255] == 123.f) out[0] = 1;
}
int main() {
  const int B = 1024;
  unsigned* d;
  hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipLaunchKernelGGL(probe, dim3(B), dim3(256), 72 * 1024, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(B * 8);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int b = 0; b < B; ++b) {
    if (!(b < 24 || (b >= 256 && b < 280) || (b >= 512 && b < 530))) continue;
    const unsigned hw = h[b * 8], xcc = h[b * 8 + 1];
    printf("wg %4d xcc %u se %u sh %u cu %2u |", b, xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
    for (int w = 0; w < 4; ++w) printf(" w%d: simd %u slot %u", w, (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15);
    printf("\n");
  }
  return 0;
}
