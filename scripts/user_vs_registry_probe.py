"""Ad-hoc probe: the same Gaussian-sum filter with registry functions (compile-time-dimension register kernel) and with the
functions given as source (run-time-compiled kernel): Lorenz-63 + quadratic emission, K components, B trajectories."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
F32 = np.float32
nl = bfa.nonlinearities
L63 = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  out[0] = th[3] * th[0] * (x[1] - x[0]) + x[0] + q[0];
  out[1] = th[3] * (x[0] * th[1] - x[1] - x[0] * x[2]) + x[1] + q[1];
  out[2] = th[3] * (x[0] * x[1] - th[2] * x[2]) + x[2] + q[2];
}"""
QUAD = """
template <class T> __device__ void emission(const T* x, const T* r, T u, const float* th, T* out) {
  out[0] = th[0] * (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) + r[0];
}"""
th = [10.0, 28.0, 2.667, 0.01]
B, T = int(os.environ.get("PB", 8192)), int(os.environ.get("PT", 500))
for K in (1, 8, 100):
    reg = bfa.ParamsNLSSM(np.array([0, 1, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(*th), np.zeros(3, F32), 0.1 * np.eye(3, dtype=F32),
                          nl.quadratic(3, 0.05), np.zeros(1, F32), np.eye(1, dtype=F32))
    usr = reg._replace(dynamics_function=nl.user_dynamics(L63, 3, theta=th), emission_function=nl.user_emission(QUAD, 3, 1, theta=[0.05]))
    y = torch.randn((B, T, 1), device="cuda")
    im = (np.array([0, 1, 1.05], F32) + 0.5 * np.random.default_rng(0).normal(size=(B, K, 3))).astype(F32)
    for name, p in (("registry", reg), ("source", usr)):
        best = 1e30
        for rep in range(3):
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            bfa.gaussian_sum_filter(p, y, K, 1, initial_means=im, fields=("weights", "means"))
            e0.record(); torch.cuda.synchronize()
            best = min(best, s0.elapsed_time(e0))
        print(f"K={K:4d} {name:9s}: {best:8.2f} ms  {B*T*K/best/1e3:9.1f} M component-steps/s", flush=True)
