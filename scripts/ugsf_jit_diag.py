import sys, os, re
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bayesianfiltering_amd as bfa
from oracle import gaussfilt_oracle as go, models as om, threefry as otf
nl = bfa.nonlinearities
F32 = np.float32
src = open("/root/repo/tests/test_user_model_gpu.py").read()
ns = {}
for name in ("L63_SRC", "QUAD_FMA_SRC"):
    exec(re.search(name + r' = """.*?"""', src, re.S).group(0), ns)
T, B, K = 6, 1, 1
Q, R = 0.1 * np.eye(3, dtype=F32), 1.0 * np.eye(1, dtype=F32)
m0, P0 = np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32)
th = [10.0, 28.0, 2.667, 0.01]
reg = bfa.ParamsNLSSM(m0, P0, nl.lorenz63(*th), np.zeros(3, F32), Q, nl.quadratic(3, 0.05), np.zeros(1, F32), R)
po = go.ParamsNLSSM(m0, P0, om.Lorenz63(), np.zeros(3, F32), Q, om.Quadratic(3, 0.05), np.zeros(1, F32), R)
ys = np.stack([go.sample_ssm(po, otf.PRNGKey(b), T)[1] for b in range(B)])
im = (m0 + 0.5 * np.random.default_rng(K).normal(size=(B, K, 3))).astype(F32)
up = bfa.ParamsUKF(1, 0, 0)
f_usr = nl.user_dynamics(ns["L63_SRC"], 3, theta=th)
h_usr = nl.user_emission(ns["QUAD_FMA_SRC"], 3, 1, theta=[0.05])
a = bfa.unscented_gaussian_sum_filter(reg, up, ys, K, 1, initial_means=im)
bits = lambda t: np.ascontiguousarray(t.cpu().numpy(), F32).view(np.int32).astype(np.int64)
for name, usr in (("dyn", reg._replace(dynamics_function=f_usr)), ("emi", reg._replace(emission_function=h_usr))):
    b_ = bfa.unscented_gaussian_sum_filter(usr, up, ys, K, 1, initial_means=im)
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        d = np.abs(bits(getattr(a, k)) - bits(getattr(b_, k)))
        print(name, k, "max ulp diff per step:", d.reshape(T, -1).max(axis=1))
# the same pair of dynamics in the particle filter: ahead-of-time registry kernel against the run-time build with L63 from source
g = nl.quadratic(3, 0.05)
rb = bfa.ParamsBPF(m0, P0, nl.lorenz63(*th), np.zeros(3, F32), Q, g, np.zeros(1, F32), R, nl.gaussian_log_prob(g, R))
ub = rb._replace(dynamics_function=f_usr)
key = np.array([0, 3], np.uint32)
o1 = bfa.bootstrap_particle_filter(rb, ys, 256, key, output="both")
o2 = bfa.bootstrap_particle_filter(ub, ys, 256, key, output="both")
print("bpf particles max ulp diff per step:", np.abs(bits(o1["particles"]) - bits(o2["particles"])).reshape(B, 256, T, 3).max(axis=(0, 1, 3)))
