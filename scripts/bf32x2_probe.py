"""Ad-hoc: kf_small_mode = 1 (one chain per wave) against 2 (two chains per wave) on the one-wave matrix-core kernel: same bits?
how fast?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from tests import common as cm
F32 = np.float32
nl = bfa.nonlinearities
def run(p, y, K, init, mode, fields=("weights", "means", "covariances", "predicted_means", "predicted_covariances")):
    fn = lambda f=fields: bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=f, return_carry=True, options={"kf_small_mode": mode})
    out = fn(); torch.cuda.synchronize()
    fast = lambda: fn(("weights",))
    fast(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fast(); e.record(); torch.cuda.synchronize()
    return out, s.elapsed_time(e)
def same(a, b):
    ok = True
    for k in ("weights", "means", "covariances", "predicted_means", "predicted_covariances"):
        x, z = getattr(a[0], k), getattr(b[0], k)
        ok &= bool(((x == z) | (torch.isnan(x) & torch.isnan(z))).all())
    for x, z in zip(a[1], b[1]):
        ok &= bool(((x == z) | (torch.isnan(x) & torch.isnan(z))).all())
    return ok
cases = []
for n, m, K, B, T, tv in ((32, 16, 1, 8191, 50, False), (32, 16, 4, 2048, 50, False), (20, 12, 3, 1001, 40, True), (16, 8, 32, 512, 50, False), (32, 32, 1, 4096, 50, True)):
    a = cm.random_stable_lgssm(n, m, seed=n)
    p = cm.product_params(a)
    if tv:
        rng = np.random.default_rng(1)
        p = p._replace(dynamics_noise_covariance=np.stack([(0.6 + rng.random()) * a["Q"] for _ in range(T)]).astype(F32),
                       emission_noise_covariance=np.stack([(0.6 + rng.random()) * a["R"] for _ in range(T)]).astype(F32))
    y = cm.device_observations(cm.product_params(a), (n, n, m, m), B, T, seed=n)
    init = torch.as_tensor(a["m0"], device="cuda") + 0.3 * torch.randn((B, K, n), device="cuda")
    cases.append((f"linear n={n} m={m} K={K} B={B} T={T} tv={tv}", p, y, K, init, B * T * K))
for n, K, B, T in ((16, 32, 512, 50), (32, 4, 4096, 50), (24, 5, 333, 30)):
    m = n // 2
    p = bfa.ParamsNLSSM(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), (1e-2 * np.eye(n)).astype(F32),
                        nl.pick_even(n), np.zeros(m, F32), (1e-1 * np.eye(m)).astype(F32))
    y = cm.device_observations(p, (n, n, m, m), B, T, seed=n)
    init = 8.0 + torch.randn((B, K, n), device="cuda")
    cases.append((f"lorenz96 n={n} K={K} B={B} T={T}", p, y, K, init, B * T * K))
for name, p, y, K, init, units in cases:
    o1, t1 = run(p, y, K, init, 1)
    o2, t2 = run(p, y, K, init, 2)
    print(f"{name:46s} one/wave {t1:8.2f} ms {units/t1/1e3:8.1f} M/s | two/wave {t2:8.2f} ms {units/t2/1e3:8.1f} M/s | x{t1/t2:.2f} | same bits: {same(o1, o2)}", flush=True)
