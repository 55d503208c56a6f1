// Ad-hoc: layout and accuracy check of a 32x32 (K = 64) product on v_mfma_f32_32x32x16_bf16 with three-way bf16 splits of
// both operands (6 of the 9 cross terms) against the fp32 MFMA and a float64 reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short bf16_rn(float x) {  // round to nearest even, as v_cvt_pk_bf16_f32
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__global__ void __launch_bounds__(64) check(const float* X, const float* Yt, float* C3, float* C32) {
  __shared__ unsigned short xs[3][32][72], ys[3][32][72];   // pitch 72 halves = 144 B (16-byte aligned rows)
  const int lane = threadIdx.x, lr = lane & 31, lk = lane >> 5;
  for (int e = lane; e < 32 * 64; e += 64) {
    const int i = e / 64, k = e % 64;
    float x = X[e], y = Yt[e];
    const unsigned short xh = bf16_rn(x); x -= bf16_f(xh);
    const unsigned short xm = bf16_rn(x); x -= bf16_f(xm);
    const unsigned short xl = bf16_rn(x);
    const unsigned short yh = bf16_rn(y); y -= bf16_f(yh);
    const unsigned short ym = bf16_rn(y); y -= bf16_f(ym);
    const unsigned short yl = bf16_rn(y);
    xs[0][i][k] = xh; xs[1][i][k] = xm; xs[2][i][k] = xl;
    ys[0][i][k] = yh; ys[1][i][k] = ym; ys[2][i][k] = yl;
  }
  __syncthreads();
  f32x16 acc = {0};
  for (int c = 0; c < 4; ++c) {
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p) {
      a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&xs[p][lr][16 * c + 8 * lk]));
      b[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(&ys[p][lr][16 * c + 8 * lk]));
    }
    // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  f32x16 ref = {0};
  for (int s = 0; s < 32; ++s) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(X[lr * 64 + 2 * s + lk], Yt[lr * 64 + 2 * s + lk], ref, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    C3[row * 32 + lr] = acc[r];
    C32[row * 32 + lr] = ref[r];
  }
}
int main() {
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> X(2048), Yt(2048), c3(1024), c32(1024);
  for (auto& v : X) v = nd(g);
  for (auto& v : Yt) v = nd(g) * 0.3f + 1.0f;   // a common offset: cancellation-free large sums too
  float *dX, *dY, *d3, *d32;
  hipMalloc(&dX, 8192); hipMalloc(&dY, 8192); hipMalloc(&d3, 4096); hipMalloc(&d32, 4096);
  hipMemcpy(dX, X.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dY, Yt.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dX, dY, d3, d32);
  hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost); hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
  double e3 = 0, e32 = 0, scale = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0, sa = 0;
      for (int k = 0; k < 64; ++k) { s += (double)X[m * 64 + k] * Yt[n * 64 + k]; sa += std::fabs((double)X[m * 64 + k] * Yt[n * 64 + k]); }
      e3 = std::fmax(e3, std::fabs(c3[m * 32 + n] - s) / sa);
      e32 = std::fmax(e32, std::fabs(c32[m * 32 + n] - s) / sa);
      scale = std::fmax(scale, std::fabs(s));
    }
  printf("max |err| / sum|terms|:  bf16x3 (6 terms) %.3e   fp32 MFMA %.3e   (max |C| %.2f)\n", e3, e32, scale);
  return 0;
}
