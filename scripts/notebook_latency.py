"""Single-trajectory wall times at the shapes of the reference's BOTExperiment.ipynb (cell 6/7: T = 500, GSF / UGSF with 100
components, AGSF / UAGSF with [2, 2, 2], BPF with 10 000 particles and ess threshold 1.0) and Experiment_TSP_2023.ipynb
(T = 100, BPF with 5e5 particles) -- the timings those notebooks print are the reference's only performance figures."""
import time
import numpy as np
import torch
import bayesianfiltering_amd as gf
from bayesianfiltering_amd import ParamsNLSSM, ParamsBPF, ParamsUKF, NonlinearSSM, nonlinearities as nl

F32 = np.float32


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def bot():
    T = 500
    mu0 = np.array([-0.05, 0.001, 0.7, -0.05], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-5 * np.eye(2, dtype=F32), 1e-4 * np.eye(2, dtype=F32)
    f, g = nl.maneuver_bot(dt=0.5, acc=0.5), nl.bearing_range()
    z2 = np.zeros(2, F32)
    inputs = np.array([1] * 200 + [0] * 100 + [2] * 200, F32)
    p = ParamsNLSSM(mu0, S0, f, z2, Q, g, z2, R)
    pb = ParamsBPF(mu0, S0, f, z2, Q, g, z2, R, nl.gaussian_log_prob(g, R))
    _, ys = NonlinearSSM(4, 2, 2, 2).sample(p, gf.PRNGKey(1), T, inputs=inputs)
    key = gf.PRNGKey(2)
    rows = [
        ("GSF, 100 components", 0.91, lambda: gf.gaussian_sum_filter(p, ys, 100, 1, inputs)),
        ("UGSF, 100 components", 1.12, lambda: gf.unscented_gaussian_sum_filter(p, ParamsUKF(1, 0, 0), ys, 100, 1, inputs)),
        ("AGSF [2,2,2]", 2.21, lambda: gf.speedy_augmented_gaussian_sum_filter(p, ys, [2, 2, 2], key, 1, (0.8, 0.8), inputs)),
        ("UAGSF [2,2,2]", 2.44, lambda: gf.speedy_unscented_agsf(p, ParamsUKF(1, 0, 0), ys, [2, 2, 2], key, 1, (0.8, 0.8), inputs)),
        ("BPF, 1e4 particles, ess 1.0", 1.68, lambda: gf.bootstrap_particle_filter(pb, ys, 10000, key, inputs, 1.0)),
    ]
    for name, ref_s, fn in rows:
        ms = timed(fn)
        print(f"BOT T=500  {name:32s} reference {ref_s:5.2f} s   engine {ms:8.2f} ms   x{ref_s * 1e3 / ms:7.0f}", flush=True)


def tsp():
    eye3 = np.eye(3, dtype=F32)
    f, g = nl.lorenz63(), nl.quadratic(3, 0.001)
    R = F32(0.1) * np.eye(1, dtype=F32)
    p = ParamsNLSSM(np.zeros(3, F32), eye3, f, np.zeros(3, F32), F32(20.0) * eye3, g, np.zeros(1, F32), R)
    pb = ParamsBPF(np.zeros(3, F32), eye3, f, np.zeros(3, F32), F32(20.0) * eye3, g, np.zeros(1, F32), R, nl.gaussian_log_prob(g, R))
    _, ys = NonlinearSSM(3, 3, 1, 1).sample(p, gf.PRNGKey(0), 100)
    rows = [
        ("GSF, 2 components", 0.21, lambda: gf.gaussian_sum_filter(p, ys, 2, 1)),
        ("UGSF, 2 components", 0.48, lambda: gf.unscented_gaussian_sum_filter(p, ParamsUKF(1, 0, 0), ys, 2, 1)),
        ("BPF, 5e5 particles", 3.84, lambda: gf.bootstrap_particle_filter(pb, ys, 500000, gf.PRNGKey(0), output="summary")),
        ("BPF, 5e5, full history out", 3.84, lambda: gf.bootstrap_particle_filter(pb, ys, 500000, gf.PRNGKey(0))),
    ]
    for name, ref_s, fn in rows:
        ms = timed(fn)
        print(f"TSP T=100  {name:32s} reference {ref_s:5.2f} s   engine {ms:8.2f} ms   x{ref_s * 1e3 / ms:7.0f}", flush=True)


if __name__ == "__main__":
    bot()
    tsp()
