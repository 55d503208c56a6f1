"""Ad-hoc: Gaussian-sum kernel, manoeuvring-target model (n=4, m=2), FULL5 outputs, for several K (staged vs strided path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
nl = bfa.nonlinearities
F32 = np.float32
mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
pp = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R)
T = 200
u = np.zeros(T, F32)
for K, B in ((128, 2048), (100, 2048), (32, 8192), (30, 8192), (5, 65536), (4, 65536), (1, 262144)):
    y = torch.randn((B, T, 2), device="cuda") * 0.1 + torch.tensor([0.9, 3.6], device="cuda")
    init = torch.as_tensor(mu0, device="cuda") + 0.05 * torch.randn((B, K, 4), device="cuda")
    post = bfa.gaussian_sum_filter(pp, y, K, 1, u, initial_means=init)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); bfa.gaussian_sum_filter(pp, y, K, 1, u, initial_means=init, out=post); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    nbytes = B * T * (8 + 4 * K * 41)
    print(f"K={K:4d} B={B:7d}: {best:8.3f} ms  {B*T*K/best/1e6:8.2f} G comp-steps/s  {nbytes/best/1e6:8.1f} GB/s", flush=True)
    del post
