#!/usr/bin/env python3
"""Bearings-only-tracking Monte-Carlo experiment on the HIP engine.

The experiment of the reference's docs/experiments/BOT_Experiment_script.py:20-245 -- manoeuvring target
(fManBOT :40), bearing + range emission (gBOT2 :43), 500 steps with the manoeuvre inputs of :46 -- with its
`for i in range(Nsim)` loop (:89) turned into the batch axis of the engine: all Nsim trajectories are
simulated and filtered in one launch per filter.  Prints the RMSE table of :238-245 (position components
0 and 2, mean +- std over the runs) and the wall time per filter for the whole batch.

Differences from the reference script: the functions come from the device registry by default (--python-functions runs the
reference's own lambdas, recorded and compiled at first use: same results to rounding, 1-2x the registry's times); the particle filter runs 4096 particles per trajectory by default (the reference: 50 000, which
--particles 50000 reproduces through the particles-in-HBM kernels); emission noise R = 1e-4 I instead of
25e-6 I, where the reference's own GSF / UGSF return NaN (BOTExperiment.ipynb cell 7).

    python examples/bot_experiment.py [--nsim 100] [--steps 500] [--components 100]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsim", type=int, default=100)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--components", type=int, default=100)
    ap.add_argument("--particles", type=int, default=4096)
    ap.add_argument("--python-functions", action="store_true",
                    help="the model as the reference writes it -- Python lambdas (:31-45) with jax.numpy / tfp's MVN replaced by "
                         "bayesianfiltering_amd.jnp / distributions.MVN -- recorded and compiled at first use, instead of the device registry")
    args = ap.parse_args()

    import torch
    import bayesianfiltering_amd as gf
    from bayesianfiltering_amd import ParamsNLSSM, ParamsBPF, ParamsUKF, NonlinearSSM, nonlinearities as nl

    F32 = np.float32
    T, M = args.steps, args.components
    mu0 = np.array([-0.05, 0.001, 0.7, -0.05], F32)                                   # :24
    Sigma0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)                             # :27
    Q = 1e-5 * np.eye(2, dtype=F32)                                                   # :28
    R = 1e-4 * np.eye(2, dtype=F32)
    q0, r0 = np.zeros(2, F32), np.zeros(2, F32)
    f, g = nl.maneuver_bot(dt=0.5, acc=0.5), nl.bearing_range()                       # :40, :43
    inputs = np.array([1] * (2 * T // 5) + [0] * (T // 5) + [2] * (T - 2 * T // 5 - T // 5), F32)   # :46
    glp = nl.gaussian_log_prob(g, R)
    if args.python_functions:
        import bayesianfiltering_amd.jnp as jnp                      # was: import jax.numpy as jnp
        from bayesianfiltering_amd.distributions import MVN          # was: tfd.MultivariateNormalFullCovariance as MVN
        dt, acc = 0.5, 0.5
        FCV = jnp.array([[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1.0]])                                        # :31

        def FCT(x, a):                                                                                                   # :33-39
            omega = 0.1 * a / jnp.sqrt(x[1] ** 2 + x[3] ** 2)
            return jnp.array([[1, jnp.sin(dt * omega) / omega, 0, -(1 - jnp.cos(dt * omega)) / omega],
                              [0, jnp.cos(dt * omega), 0, -jnp.sin(dt * omega)],
                              [0, (1 - jnp.cos(dt * omega)) / omega, 1, jnp.sin(dt * omega) / omega],
                              [0, jnp.sin(dt * omega), 0, jnp.cos(dt * omega)]])
        G = jnp.array([[0.5, 0], [1, 0], [0, 0.5], [0, 1.0]])
        f = lambda x, q, u: (0.5 * (u - 1) * (u - 2) * FCV - u * (u - 2) * FCT(x, acc) + 0.5 * u * (u - 1) * FCT(x, -acc)) @ x + G @ q   # :42
        g = lambda x, r, u: jnp.array([jnp.arctan2(x[2], x[0]), jnp.sqrt(x[0] ** 2 + x[2] ** 2)]) + r                        # :43-44
        glp = lambda x, y, u: MVN(loc=g(x, r0, u), covariance_matrix=R).log_prob(y)                                       # :45
    params = ParamsNLSSM(mu0, Sigma0, f, q0, Q, g, r0, R)
    params_bpf = ParamsBPF(mu0, Sigma0, f, q0, Q, g, r0, R, glp)

    model = NonlinearSSM(4, 2, 2, 2)
    keys = np.stack([gf.PRNGKey(1000 + i) for i in range(args.nsim)])
    states, emissions = model.sample(params, keys, T, inputs=inputs)                  # (Nsim, T, 4), (Nsim, T, 2)

    def point_estimate(weights, means):   # jnp.sum(jnp.einsum('ijk,ij->ijk', means, weights), axis=0)      :101
        return (weights.unsqueeze(-1) * means).sum(dim=1)

    def timed(fn):
        fn()                                  # first call: code-object load, allocator warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    results = {}
    post, dt = timed(lambda: gf.gaussian_sum_filter(params, emissions, M, 1, inputs, fields=("weights", "means")))
    results["GSF"] = (point_estimate(post.weights, post.means), dt)
    post, dt = timed(lambda: gf.unscented_gaussian_sum_filter(params, ParamsUKF(1, 0, 0), emissions, M, 1, inputs,
                                                              fields=("weights", "means")))
    results["U-GSF"] = (point_estimate(post.weights, post.means), dt)
    nc = [M, 2, 2]                                                                    # :118
    (post, _), dt = timed(lambda: gf.speedy_augmented_gaussian_sum_filter(params, emissions, nc, gf.PRNGKey(2), 1, (0.9, 0.9),
                                                                          inputs))
    results["AGSF"] = (point_estimate(post.weights, post.means), dt)
    (post, _), dt = timed(lambda: gf.speedy_unscented_agsf(params, ParamsUKF(1, 0, 0), emissions, nc, gf.PRNGKey(2), 1,
                                                           (0.9, 0.9), inputs))
    results["U-AGSF"] = (point_estimate(post.weights, post.means), dt)
    out, dt = timed(lambda: gf.bootstrap_particle_filter(params_bpf, emissions, args.particles, gf.PRNGKey(3), inputs, 1.0,
                                                         output="summary"))
    results["BPF"] = (out["mean"], dt)

    pos = [0, 2]
    print(f"{args.nsim} Monte-Carlo runs, T = {T}, {M} components / {args.particles} particles")
    print(f"{'filter':8s} {'RMSE (mean +- std)':>24s} {'time for all runs':>20s} {'per run':>12s}")
    for name, (est, dt) in results.items():
        err = est[:, :, pos] - states[:, :, pos]
        rmse = torch.sqrt((err ** 2).sum(dim=(1, 2)) / T).cpu().numpy()              # utils.rmse :184-187, per run
        ok = np.isfinite(rmse)
        stat = f"{rmse[ok].mean():12.5f} +- {rmse[ok].std():8.5f}" if ok.any() else f"{'nan':>12s}    {'':8s}"
        note = "" if ok.all() else f"   ({(~ok).sum()} runs NaN: linear-domain weights underflow to 0/0, inference.py:347-350)"
        print(f"{name:8s} {stat} {dt * 1e3:17.1f} ms {dt / args.nsim * 1e3:9.3f} ms{note}")


if __name__ == "__main__":
    main()
