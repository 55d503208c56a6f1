"""Ad-hoc: where do the benchmark's GSF / PF runs leave the finite numbers?  (linear-domain weights, inference.py:347-350)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
import bench
F32 = np.float32
nl = bfa.nonlinearities
# ---- cfg3
B, T, K, n, m = 2048, 5000, 32, 8, 4
for p0, q in ((1.0, 1e-2), (0.1, 1e-2), (0.01, 1e-2), (1.0, 1e-1)):
    p = bfa.ParamsNLSSM(8 * np.ones(8, F32), p0 * np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32), q * np.eye(8, dtype=F32),
                        nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    y = bench.simulate_on_device(p, (8, 8, 4, 4), B, T, seed=7)
    g = torch.Generator(device="cuda").manual_seed(1)
    init = 8.0 + np.sqrt(p0) * torch.randn((B, K, n), device="cuda", generator=g)
    fin = []
    carry = None
    for t0 in range(0, T, 500):
        post, carry = bfa.gaussian_sum_filter(p, y[:, t0:t0 + 500], K, 1, initial_means=init, carry=carry, fields=("weights",), return_carry=True)
        fin.append(float(torch.isfinite(post.weights[:, :, -1]).all(dim=1).float().mean()))
    neff = 1.0 / (carry.weights ** 2).sum(dim=1)
    print(f"cfg3 P0={p0} Q={q}: y finite {float(torch.isfinite(y).all(dim=(1,2)).float().mean()):.3f}; finite weights after each 500 steps:", [round(f, 3) for f in fin], "median n_eff", float(neff[torch.isfinite(neff)].median()) if torch.isfinite(neff).any() else None, flush=True)
# ---- cfg4
B, T, N = 256, 2000, 4096
for q, r in ((1e-1, 0.5), (1e-2, 0.5), (1e-2, 0.1)):
    g = nl.pick_even(16); R = r * np.eye(8, dtype=F32)
    p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32), q * np.eye(16, dtype=F32), g, np.zeros(8, F32), R,
                      nl.gaussian_log_prob(g, R))
    y = bench.simulate_on_device(bfa.ParamsNLSSM(*p[:8]), (16, 16, 8, 8), B, T, seed=9)
    out = bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), output="summary")
    fin_t = torch.isfinite(out["mean"]).all(dim=2)
    first_bad = torch.where(fin_t.all(dim=1), T, (~fin_t).float().argmax(dim=1))
    print(f"cfg4 Q={q} R={r}: y finite {float(torch.isfinite(y).all(dim=(1,2)).float().mean()):.3f} max|y| {float(y[torch.isfinite(y)].abs().max()):.1f}; finite trajectories {float(fin_t.all(dim=1).float().mean()):.3f}; "
          f"first non-finite step: min {int(first_bad.min())} median {int(first_bad.median())}; resampled {float(out['resampled'].mean()):.2f}; min ESS {float(out['ess'][torch.isfinite(out['ess'])].min()):.1f}", flush=True)
