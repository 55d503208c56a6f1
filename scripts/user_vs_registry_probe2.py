"""Ad-hoc probe: particle / unscented / augmented filters with the BOT model from the registry and from source (run-time builds)."""
import os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
F32 = np.float32
nl = bfa.nonlinearities
t = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "test_user_model_gpu.py")).read()
DYN = re.search(r'BOT_DYN_SRC = """(.*?)"""', t, flags=re.S).group(1)
EMI = re.search(r'BOT_EMI_SRC = """(.*?)"""', t, flags=re.S).group(1)
mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32); S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32); r0 = np.zeros(2, F32)
B, T = int(os.environ.get("PB", 4096)), int(os.environ.get("PT", 100))
reg = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), r0, R)
f_usr = nl.user_dynamics(DYN, 4, noise_dim=2, theta=nl.maneuver_bot().theta); g_usr = nl.user_emission(EMI, 4, 2)
usr = reg._replace(dynamics_function=f_usr, emission_function=g_usr)
u = torch.tensor(np.array([1] * (T // 3) + [0] * (T // 3) + [2] * (T - 2 * (T // 3)), F32), device="cuda")
y = torch.tensor(np.array([0.98, 3.6], F32), device="cuda") + 0.05 * torch.randn((B, T, 2), device="cuda")
def timed(fn):
    best = 1e30
    for rep in range(3):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(); fn(); e0.record(); torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0))
    return best
for name, p in (("registry", reg), ("source", usr)):
    g = p.emission_function
    pb = bfa.ParamsBPF(*p, nl.gaussian_log_prob(g, R, r0))
    im = (mu0 + 0.05 * np.random.default_rng(0).normal(size=(B, 20, 4))).astype(F32)
    print(f"{name:9s} BPF N=1000  {timed(lambda: bfa.bootstrap_particle_filter(pb, y, 1000, np.array([0, 1], np.uint32), u, output='summary')):8.2f} ms", flush=True)
    print(f"{name:9s} UGSF K=20   {timed(lambda: bfa.unscented_gaussian_sum_filter(p, bfa.ParamsUKF(1, 0, 0), y, 20, 1, u, initial_means=im, fields=('means',))):8.2f} ms", flush=True)
    print(f"{name:9s} AGSF 20x2x2 {timed(lambda: bfa.speedy_augmented_gaussian_sum_filter(p, y, (20, 2, 2), None, 1, (0.1, 0.1), u, initial_means=im)):8.2f} ms", flush=True)
    print(f"{name:9s} GSF K=20    {timed(lambda: bfa.gaussian_sum_filter(p, y, 20, 1, u, initial_means=im, fields=('means',))):8.2f} ms", flush=True)
