/* A C program (not C++, no Python) that drives the engine through include/bayesfilt.h alone:
 * hipMalloc'd buffers, bf_abi_check, bf_kalman_filter_f32 on the constant-velocity model of
 * BASELINE configs[0] (n = 4, m = 2), all five posterior streams in the reference layout.
 * Prints the observations it used and every output as text; tests/test_c_abi_program.py
 * compiles it with gcc, runs it and compares the numbers with the oracle.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c/kalman_from_c.c \
 *       -L bayesianfiltering_amd -lbayesfilt_hip -L /opt/rocm/lib -lamdhip64 -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "bayesfilt.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 3, T = argc > 2 ? atoi(argv[2]) : 24;
  enum { n = 4, m = 2 };
  int rc = bf_abi_check(BF_VERSION, sizeof(bf_out_desc), sizeof(bf_lgssm), sizeof(bf_model), sizeof(bf_bpf_model), sizeof(bf_bpf_out));
  if (rc != BF_OK) { fprintf(stderr, "abi: %s\n", bf_last_error()); return 2; }
  /* a wrong size must be refused */
  if (bf_abi_check(BF_VERSION, sizeof(bf_out_desc) - sizeof(bf_stream), sizeof(bf_lgssm), sizeof(bf_model), sizeof(bf_bpf_model),
                   sizeof(bf_bpf_out)) != BF_EINVAL) { fprintf(stderr, "abi check accepted a short bf_out_desc\n"); return 2; }
  printf("version %d devices %d\n", bf_version(), bf_device_count());
  if (bf_device_count() < 1) { printf("no-gpu\n"); return 0; }

  const float dt = 0.5f;
  const float A[n * n] = {1, dt, 0, 0, 0, 1, 0, 0, 0, 0, 1, dt, 0, 0, 0, 1};
  const float G[n * 2] = {0.5f, 0, 1, 0, 0, 0.5f, 0, 1};
  const float H[m * n] = {1, 0, 0, 0, 0, 0, 1, 0};
  const float Q[4] = {1e-2f, 0, 0, 1e-2f}, R[4] = {1e-1f, 0, 0, 1e-1f};
  bf_lgssm mdl = {n, 2, m, m, A, G, H, NULL, NULL, NULL, Q, R, 1, 1};

  /* deterministic pseudo-observations (a slow spiral plus a small LCG jitter) */
  float* y = (float*)malloc(sizeof(float) * B * T * m);
  unsigned s = 12345u;
  for (int i = 0; i < B * T * m; ++i) {
    s = s * 1664525u + 1013904223u;
    y[i] = 0.05f * (float)(i % (T * m)) + ((float)(s >> 8) / 16777216.0f - 0.5f);
  }
  float* m0 = (float*)calloc((size_t)B * n, sizeof(float));
  float* P0 = (float*)calloc((size_t)B * n * n, sizeof(float));
  for (int b = 0; b < B; ++b) for (int i = 0; i < n; ++i) P0[b * n * n + i * n + i] = 1.0f;

  float *d_y, *d_m0, *d_P0, *d_w, *d_m, *d_P, *d_pm, *d_pP, *d_ll;
  CHECK_HIP(hipMalloc((void**)&d_y, sizeof(float) * B * T * m));
  CHECK_HIP(hipMalloc((void**)&d_m0, sizeof(float) * B * n));
  CHECK_HIP(hipMalloc((void**)&d_P0, sizeof(float) * B * n * n));
  CHECK_HIP(hipMalloc((void**)&d_w, sizeof(float) * B * T));
  CHECK_HIP(hipMalloc((void**)&d_ll, sizeof(float) * B * T));
  CHECK_HIP(hipMalloc((void**)&d_m, sizeof(float) * B * T * n));
  CHECK_HIP(hipMalloc((void**)&d_pm, sizeof(float) * B * T * n));
  CHECK_HIP(hipMalloc((void**)&d_P, sizeof(float) * B * T * n * n));
  CHECK_HIP(hipMalloc((void**)&d_pP, sizeof(float) * B * T * n * n));
  CHECK_HIP(hipMemcpy(d_y, y, sizeof(float) * B * T * m, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_m0, m0, sizeof(float) * B * n, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_P0, P0, sizeof(float) * B * n * n, hipMemcpyHostToDevice));

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  bf_cstream yd = {d_y, (int64_t)T * m, 0, m, 1};
  bf_carry cr = {NULL, d_m0, d_P0, NULL, NULL, NULL};
  bf_out_desc od = {{0}};
  /* reference layout [B][K=1][T][E] */
  od.weights = (bf_stream){d_w, T, T, 1, 1};
  od.loglik = (bf_stream){d_ll, T, T, 1, 1};
  od.means = (bf_stream){d_m, (int64_t)T * n, (int64_t)T * n, n, 1};
  od.pred_means = (bf_stream){d_pm, (int64_t)T * n, (int64_t)T * n, n, 1};
  od.covs = (bf_stream){d_P, (int64_t)T * n * n, (int64_t)T * n * n, n * n, 1};
  od.pred_covs = (bf_stream){d_pP, (int64_t)T * n * n, (int64_t)T * n * n, n * n, 1};
  rc = bf_kalman_filter_f32(&mdl, &yd, B, T, &cr, &od, (void*)stream);
  if (rc != BF_OK) { fprintf(stderr, "bf_kalman_filter_f32: %d %s\n", rc, bf_last_error()); return 4; }
  CHECK_HIP(hipStreamSynchronize(stream));

  /* an unsupported request must come back as a status code with text, not as a crash */
  bf_lgssm bad = mdl;
  bad.n = 0;
  if (bf_kalman_filter_f32(&bad, &yd, B, T, &cr, &od, (void*)stream) != BF_EINVAL) { fprintf(stderr, "n = 0 accepted\n"); return 5; }

  float* buf = (float*)malloc(sizeof(float) * B * T * n * n);
  struct { const char* name; float* d; int E; } outs[] = {{"weights", d_w, 1}, {"loglik", d_ll, 1}, {"means", d_m, n}, {"predicted_means", d_pm, n},
                                                           {"covariances", d_P, n * n}, {"predicted_covariances", d_pP, n * n}};
  printf("B %d T %d n %d m %d\n", B, T, n, m);
  printf("emissions");
  for (int i = 0; i < B * T * m; ++i) printf(" %.9g", y[i]);
  printf("\n");
  for (int k = 0; k < 6; ++k) {
    CHECK_HIP(hipMemcpy(buf, outs[k].d, sizeof(float) * B * T * outs[k].E, hipMemcpyDeviceToHost));
    printf("%s", outs[k].name);
    for (int i = 0; i < B * T * outs[k].E; ++i) printf(" %.9g", buf[i]);
    printf("\n");
  }
  return 0;
}
