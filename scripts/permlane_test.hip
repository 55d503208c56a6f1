// Semantics check of v_permlane16_swap / v_permlane32_swap on gfx950 (prints the two results for v = lane id).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  int a = threadIdx.x, a2 = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, a2, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, a2, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  int* d; hipMalloc(&d, 1024); int h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; i += 4) printf(" %d", h[j * 64 + i]); printf("\n"); }
  return 0;
}
