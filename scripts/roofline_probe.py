"""One launch (after a warm-up) of a kernel that had no stated roofline, at a fixed shape, for rocprofv3 PMC passes and HIP-event
timing: usage  python scripts/roofline_probe.py {bf32|ugsf|agsf|uagsf|gsf_collapsed}.  Prints one JSON line with the shape,
the HIP-event time of the launch and the units processed."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from tests import common as cm
nl = bfa.nonlinearities
F32 = np.float32
which = sys.argv[1]
mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
Qb, Rb = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
bot = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Qb, nl.bearing_range(), np.zeros(2, F32), Rb)

if which == "bf32":        # kf_scan_bf32_kernel: one wave per trajectory, (n, m) = (32, 16), no output streams (carry only)
    n, m, B, T = 32, 16, 16384, 100
    a = cm.random_stable_lgssm(n, m, seed=n)
    p = cm.product_params(a)
    y = cm.device_observations(p, (n, n, m, m), B, T, seed=32)
    init = torch.zeros((B, n), device="cuda")
    fn = lambda: bfa.kalman_filter(p, y, initial_means=init, fields=(), return_carry=True)
    info = {"kernel": "kf_scan_bf32_kernel", "n": n, "m": m, "B": B, "T": T, "units": B * T, "unit": "timesteps"}
elif which == "ugsf":      # ugsf_scan_kernel at the reference notebook's shape (BOTExperiment.ipynb: K = 100, n = 4, T = 500)
    B, T, K = 8192, 500, 100
    y = torch.randn((B, T, 2), device="cuda") * 0.1 + torch.tensor([0.9, 3.6], device="cuda")
    ib = torch.as_tensor(mu0, device="cuda") + 0.05 * torch.randn((B, K, 4), device="cuda")
    u = np.zeros(T, F32)
    fn = lambda: bfa.unscented_gaussian_sum_filter(bot, bfa.ParamsUKF(1, 0, 0), y, K, 1, u, initial_means=ib, fields=("weights", "means"))
    info = {"kernel": "ugsf_scan_kernel", "n": 4, "m": 2, "K": K, "B": B, "T": T, "units": B * T * K, "unit": "component-steps"}
elif which in ("agsf", "uagsf"):   # agsf_scan_kernel, tree [100, 2, 2] (400 leaves per trajectory), extended / unscented nodes
    B, T, nc = 4096, 500, (100, 2, 2)
    y = torch.randn((B, T, 2), device="cuda") * 0.1 + torch.tensor([0.9, 3.6], device="cuda")
    u = np.zeros(T, F32)
    if which == "agsf":
        fn = lambda: bfa.speedy_augmented_gaussian_sum_filter(bot, y, nc, inputs=u)
    else:
        fn = lambda: bfa.speedy_unscented_agsf(bot, bfa.ParamsUKF(1, 0, 0), y, nc, inputs=u)
    info = {"kernel": "agsf_scan_kernel", "nodes": "ukf" if which == "uagsf" else "ekf", "tree": list(nc), "B": B, "T": T,
            "units": B * T * nc[0] * nc[1] * nc[2], "unit": "leaf-steps"}
elif which == "gsf_collapsed":   # gsf_scan_kernel<8,4,2,EMIT_NONE,...>: cfg3 in COLLAPSED mode (in-scan moment matching), as bench.py runs it
    B, T, K, n, m = 16384, 1000, 32, 8, 4
    p = bfa.ParamsNLSSM(8 * np.ones(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8, mode="as_written"), np.zeros(8, F32),
                        1e-2 * np.eye(8, dtype=F32), nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    y = cm.device_observations(p, (8, 8, 4, 4), B, T, seed=2000)
    g = torch.Generator(device="cuda").manual_seed(20)
    init = 8.0 + torch.randn((B, K, n), device="cuda", generator=g)
    lanes = int(os.environ.get("PLANES", "0"))
    fn = lambda: bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=(), return_carry=True, return_collapsed=True,
                                         options={"kf_lanes": lanes} if lanes else None)
    info = {"kernel": "gsf_scan_kernel", "mode": "COLLAPSED", "lanes_per_chain": lanes or 2, "n": n, "m": m, "K": K, "B": B, "T": T, "units": B * T, "unit": "timesteps"}
else:
    raise SystemExit("unknown probe")
fn(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); fn(); e.record(); torch.cuda.synchronize()
info["ms"] = s.elapsed_time(e)
info["units_per_s"] = info["units"] / (info["ms"] * 1e-3)
print(json.dumps(info), flush=True)
