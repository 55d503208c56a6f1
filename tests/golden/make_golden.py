#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the NumPy oracle.

The reference (gaussfiltax) cannot run in the build container (no jax / tensorflow_probability)
and its own tests hold no vectors, so these fixtures are produced by oracle/gaussfilt_oracle.py
-- the fp32 restatement of gaussfiltax/inference.py:51-120,146-224,303-456,621-812,1302-1380 -- and pin the
HIP kernels (tests -m gpu) and the C port against it.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gaussfilt_oracle as go, models as om, threefry as otf  # noqa: E402
from tests import common as cm  # noqa: E402

F32 = np.float32
OUT = os.path.dirname(os.path.abspath(__file__))


def kalman_cv():
    """(i) LGSSM n=4, m=2 (constant-velocity model of SURVEY.md 8(d) cfg1/cfg2), B=3, T=64."""
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, 3, 64, seed=20231003)
    init = (np.tile(a["m0"], (3, 1)) + np.array([[0, 0, 0, 0], [1, -1, 0.5, 0.2], [-2, 0.3, 1, -0.4]], F32)).astype(F32)
    ref = cm.oracle_kalman_batch(a, ys, init)
    np.savez_compressed(os.path.join(OUT, "kalman_cv_n4_m2_T64.npz"), emissions=ys, initial_means=init,
                        **{k: v for k, v in a.items()}, **{"out_" + k: v for k, v in ref.items()})


def kalman_random():
    """LGSSM n=3, m=3 with biases, non-identity G/D and non-diagonal Q/R, B=2, T=40."""
    a = cm.random_stable_lgssm(3, 3, seed=7, dq=3, dr=3, bias=True)
    ys = cm.simulate_batch(a, 2, 40, seed=8)
    init = np.tile(a["m0"], (2, 1)).astype(F32)
    ref = cm.oracle_kalman_batch(a, ys, init)
    np.savez_compressed(os.path.join(OUT, "kalman_random_n3_m3_T40.npz"), emissions=ys, initial_means=init,
                        **{k: v for k, v in a.items()}, **{"out_" + k: v for k, v in ref.items()})


def gsf_models():
    """(iii) Gaussian-sum filter, K=4: Lorenz-63 (n=3, scalar quadratic emission) and Lorenz-96
    (n=8, m=4, both modes of gaussfiltax/nonlinearities.py:48), T=32."""
    rng = np.random.default_rng(63)
    # Lorenz-63, g(x) = 0.05 * x.x + r
    f63, g63 = om.Lorenz63(), om.Quadratic(3, 0.05)
    p = go.ParamsNLSSM(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), f63, np.zeros(3, F32),
                       0.1 * np.eye(3, dtype=F32), g63, np.zeros(1, F32), np.eye(1, dtype=F32))
    xs, ys = go.sample_ssm(p, otf.PRNGKey(63), 32)
    im = (p.initial_mean[None] + rng.normal(size=(4, 3))).astype(F32)
    post, ll = go.gaussian_sum_filter(p, ys, 4, initial_means=im, return_ll=True)
    np.savez_compressed(os.path.join(OUT, "gsf_lorenz63_K4_T32.npz"), emissions=ys, states=xs, initial_means=im,
                        loglik=ll, **post._asdict())
    for mode in ("matrix_power", "as_written"):
        f96, g96 = om.Lorenz96(8, mode=mode), om.PickEven(8)
        p = go.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), f96, np.zeros(8, F32), 1e-2 * np.eye(8, dtype=F32),
                           g96, np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
        xs, ys = go.sample_ssm(p, otf.PRNGKey(96), 32)
        im = (rng.normal(size=(4, 8))).astype(F32)
        post, ll = go.gaussian_sum_filter(p, ys, 4, initial_means=im, return_ll=True)
        np.savez_compressed(os.path.join(OUT, f"gsf_lorenz96_{mode}_n8_K4_T32.npz"), emissions=ys, states=xs,
                            initial_means=im, loglik=ll, **post._asdict())


def bpf_small():
    """(iv) bootstrap particle filter N=64, Lorenz-63 with a linear emission (n=3, m=3), T=16, threefry key (0, 7).
    Ancestors / resample flags are part of the fixture: index parity must be bit-exact."""
    f63, h = om.Lorenz63(), om.Linear(np.eye(3, dtype=F32))
    R = 0.5 * np.eye(3, dtype=F32)
    p = go.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), f63, np.zeros(3, F32),
                     0.1 * np.eye(3, dtype=F32), h, np.zeros(3, F32), R, go.GaussianEmissionLogProb(h, R))
    xs, ys = go.sample_ssm(go.ParamsNLSSM(*p[:8]), otf.PRNGKey(5), 16)
    key = np.array([0, 7], np.uint32)
    out, dbg = go.bootstrap_particle_filter(p, ys, 64, key=key, debug=True)
    np.savez_compressed(os.path.join(OUT, "bpf_lorenz63_N64_T16.npz"), emissions=ys, key=key, weights=out["weights"],
                        particles=out["particles"], resampled=dbg["resampled"], ancestors=dbg["ancestors"], ess=dbg["ess"])
    # the same run on the canonical fp32 arithmetic (oracle/fp32.py): the HIP engine must reproduce EVERY BIT of it
    out, dbg = go.bootstrap_particle_filter(p, ys, 64, key=key, debug=True, arith="canonical")
    np.savez_compressed(os.path.join(OUT, "bpf_lorenz63_N64_T16_canonical.npz"), emissions=ys, key=key, weights=out["weights"],
                        particles=out["particles"], resampled=dbg["resampled"], ancestors=dbg["ancestors"], ess=dbg["ess"],
                        pre_weights=dbg["pre_weights"])


def rng_vectors():
    """(v) PRNG restatement: Threefry block outputs and the derived split / uniform / normal draws."""
    k = otf.PRNGKey(0)
    np.savez_compressed(os.path.join(OUT, "threefry_vectors.npz"),
                        split_0_5=otf.split(k, 5), bits_0_9=otf.random_bits(k, 9), uniform_0_8=otf.uniform(k, 8),
                        normal_0_8=otf.normal(k, 8), normal_42_3=otf.normal(otf.PRNGKey(42), 3),
                        cumsum_in=np.linspace(0.1, 1.3, 13, dtype=F32), cumsum_out=otf.cumsum_assoc(np.linspace(0.1, 1.3, 13, dtype=F32)))


def _bot_model():
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    return go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R), mu0


def unscented_and_augmented():
    """(f-2, f-3) unscented GSF (ParamsUKF(1, 0, 0), K = 4) and speedy augmented GSF ((3, 2, 2) components) on the
    manoeuvring-target / bearing + range model of BOT_Experiment_script.py, T = 24 with inputs."""
    p, mu0 = _bot_model()
    T = 24
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32).reshape(T, 1)
    xs, ys = go.sample_ssm(p, otf.PRNGKey(42), T, inputs)
    im = (mu0 + 0.05 * np.random.default_rng(3).normal(size=(4, 4))).astype(F32)
    post = go.unscented_gaussian_sum_filter(p, go.ParamsUKF(1, 0, 0), ys, 4, inputs=inputs, initial_means=im)
    np.savez_compressed(os.path.join(OUT, "ugsf_bot_K4_T24.npz"), emissions=ys, states=xs, inputs=inputs, initial_means=im,
                        uparams=np.array([1, 0, 0], F32), **post._asdict())
    post, aux = go.speedy_augmented_gaussian_sum_filter(p, ys, (3, 2, 2), otf.PRNGKey(5), 1, (0.2, 0.3), inputs,
                                                        initial_means=im[:3], debug=True)
    np.savez_compressed(os.path.join(OUT, "agsf_bot_322_T24.npz"), emissions=ys, inputs=inputs, initial_means=im[:3],
                        key=otf.PRNGKey(5), opt_args=np.array([0.2, 0.3], F32), weights=post.weights, means=post.means,
                        covariances=post.covariances, pre_weights=aux["pre_weights"])


if __name__ == "__main__":
    kalman_cv(); kalman_random(); gsf_models(); bpf_small(); rng_vectors(); unscented_and_augmented()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
