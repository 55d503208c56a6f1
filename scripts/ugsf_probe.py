"""Ad-hoc: error levels and throughput of the unscented Gaussian-sum filter kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm
nl = bfa.nonlinearities
F32 = np.float32
rng = np.random.default_rng(0)
T, K, B = 24, 5, 3
mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32)
po = go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R)
pp = bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R)
ys = np.stack([go.sample_ssm(po, otf.PRNGKey(10 + b), T, inputs.reshape(T, 1))[1] for b in range(B)])
init = (mu0 + 0.05 * rng.normal(size=(B, K, 4))).astype(F32)
post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, K, 1, inputs, initial_means=init)
for b in range(B):
    ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(1, 0, 0), ys[b], K, initial_means=init[b], inputs=inputs.reshape(T, 1))
    print("BOT b", b, {k: float(cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k))) for k in ("means", "covariances", "predicted_covariances")})
# throughput at the reference notebook's shape (BOTExperiment.ipynb: K = 100, n = 4, T = 500), batched
Tb, Kb = 500, 100
for Bb in (1, 1024, 8192):
    yb = torch.randn((Bb, Tb, 2), device="cuda") * 0.1 + torch.tensor([0.9, 3.6], device="cuda")
    ib = torch.as_tensor(mu0, device="cuda") + 0.05 * torch.randn((Bb, Kb, 4), device="cuda")
    ub = np.zeros(Tb, F32)
    for name, fn in (("UGSF", lambda: bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), yb, Kb, 1, ub, initial_means=ib, fields=("weights", "means"))),
                     ("GSF ", lambda: bfa.gaussian_sum_filter(pp, yb, Kb, 1, ub, initial_means=ib, fields=("weights", "means")))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"{name} B={Bb} K={Kb} T={Tb}: {el*1e3:8.2f} ms  {Bb*Tb/el:10.3e} steps/s  {Bb*Tb*Kb/el:10.3e} component-steps/s", flush=True)
