"""Ad-hoc: Gaussian-sum filter of a linear model at (n, m) on the matrix cores against the run-time-dimension kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from tests import common as cm
F32 = np.float32
for n, m, K, B, T, tv in ((32, 16, 32, 2048, 50, False), (32, 16, 4, 8192, 50, False), (32, 16, 4, 8192, 50, True), (16, 8, 32, 2048, 50, False),
                          (32, 16, 1, 16384, 50, True), (64, 32, 4, 2048, 40, False), (64, 32, 32, 512, 20, False), (64, 32, 1, 8192, 40, True)):
    a = cm.random_stable_lgssm(n, m, seed=n)
    p = cm.product_params(a)
    if tv:
        rng = np.random.default_rng(1)
        p = p._replace(dynamics_noise_covariance=np.stack([(0.6 + rng.random()) * a["Q"] for _ in range(T)]).astype(F32),
                       emission_noise_covariance=np.stack([(0.6 + rng.random()) * a["R"] for _ in range(T)]).astype(F32))
    y = cm.device_observations(cm.product_params(a), (n, n, m, m), B, T, seed=n)
    init = torch.as_tensor(a["m0"], device="cuda") + 0.3 * torch.randn((B, K, n), device="cuda")
    for force in (0, 1):
        fn = lambda: bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=("weights",), return_carry=True, options={"force_generic": force})
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        print(f"{'generic     ' if force else 'matrix cores'} n={n} m={m} K={K} B={B} T={T} tv={tv}: {ms:8.2f} ms  {B*T/ms/1e3:9.3f} Mstep/s  {B*T*K/ms/1e3:9.3f} M component-steps/s", flush=True)
# extended Kalman chains: Lorenz-96 with the even-state emission
nl = bfa.nonlinearities
for n, K, B, T in ((16, 32, 2048, 50), (32, 32, 2048, 50), (32, 4, 8192, 50)):
    m = n // 2
    p = bfa.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), (1e-2 * np.eye(n)).astype(F32),
                        nl.pick_even(n), np.zeros(m, F32), (1e-1 * np.eye(m)).astype(F32))
    y = cm.device_observations(p, (n, n, m, m), B, T, seed=n)
    init = 8.0 + torch.randn((B, K, n), device="cuda")
    for force in (0, 1):
        fn = lambda: bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=("weights",), return_carry=True, options={"force_generic": force})
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        print(f"{'generic     ' if force else 'matrix cores'} lorenz96 n={n} K={K} B={B} T={T}: {ms:8.2f} ms  {B*T/ms/1e3:9.3f} Mstep/s  {B*T*K/ms/1e3:9.3f} M component-steps/s", flush=True)
