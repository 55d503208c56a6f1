"""Ad-hoc: throughput of the augmented Gaussian-sum filter kernels at the reference experiment's shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
nl = bfa.nonlinearities
F32 = np.float32
mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
pp = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R)
T = 500
u = np.zeros(T, F32)
for nc in ((2, 2, 2), (100, 2, 2)):
    for B in (1, 256, 4096):
        y = torch.randn((B, T, 2), device="cuda") * 0.1 + torch.tensor([0.9, 3.6], device="cuda")
        for name, fn in (("AGSF ", lambda: bfa.speedy_augmented_gaussian_sum_filter(pp, y, nc, inputs=u)),
                         ("UAGSF", lambda: bfa.speedy_unscented_agsf(pp, bfa.ParamsUKF(1, 0, 0), y, nc, inputs=u))):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
            print(f"{name} nc={nc} B={B} T={T}: {el*1e3:9.2f} ms  {B*T/el:10.3e} steps/s  {B*T*nc[0]*nc[1]*nc[2]/el:10.3e} leaf-steps/s", flush=True)
