"""The reference's own "unit" tests, docs/tests/test_inference.py:55-104, run against the HIP engine.

There the six tests build a 4-state manoeuvring-target model observed through its bearing
(:22-53), sample 30 steps (:72) and *return* the posterior of each filter without asserting anything.
Here the same model, sizes and calls (same argument order) must run, return the reference's shapes,
and -- the assertion the reference lacks -- agree with the NumPy oracle.  The reference's emission
noise R = 25e-6 makes S = H P H^T + R so small that psd_solve's +1e-6 jitter dominates it and
fp32 parity is lost within a few steps (the reference's own GSF returns NaN on this problem,
BOTExperiment.ipynb cell 7); the parity half of each test therefore uses R = 1e-2, the shape / finiteness
half uses the reference's value."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32
T = 30                                                                  # seq_length            :26
INPUTS = np.array([1] * 10 + [0] * 10 + [2] * 10, F32)                   # manoeuvre inputs      :48
MU0 = np.ones(4, F32)                                                    #                       :28
SIGMA0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)                    #                       :31
Q = np.eye(2, dtype=F32)                                                 #                       :32


def _params(R):
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    Rm = R * np.eye(1, dtype=F32)
    po = go.ParamsNLSSM(MU0, SIGMA0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.Bearing(), np.zeros(1, F32), Rm)
    pp = bfa.ParamsNLSSM(MU0, SIGMA0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing(), np.zeros(1, F32), Rm)
    return bfa, po, pp


def _data(bfa, pp):
    # model.sample(params, key, seq_length, inputs=inputs)                                        :72
    states, emissions = bfa.NonlinearSSM(4, 2, 1, 1).sample(pp, bfa.PRNGKey(0), T, inputs=INPUTS)
    return emissions


def _finite_shapes(post, K):
    assert tuple(post.means.shape) == (K, T, 4) and tuple(post.covariances.shape) == (K, T, 4, 4)
    assert tuple(post.weights.shape) == (K, T)


def test_gaussian_sum_filter():                                          # :74-77
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    post = bfa.gaussian_sum_filter(pp, ys, 5, 1, INPUTS)
    _finite_shapes(post, 5)
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    im = bfa.sample_initial_component_means(pp, 5)
    ref = go.gaussian_sum_filter(po, ys, 5, initial_means=im, inputs=INPUTS.reshape(T, 1))
    post = bfa.gaussian_sum_filter(pp, ys, 5, 1, INPUTS)                 # default PRNGKey(0) draw of :367
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, k
    assert np.max(np.abs(post.weights.cpu().numpy() - ref.weights)) < 5e-5


def test_speedy_agsf():                                                  # :79-82
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    post, aux = bfa.speedy_augmented_gaussian_sum_filter(pp, ys, [5, 3, 2], opt_args=(0.1, 0.1), inputs=INPUTS)
    _finite_shapes(post, 5)
    assert bool(np.isfinite(post.means.cpu().numpy()).all())
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    ref, raux = go.speedy_augmented_gaussian_sum_filter(po, ys, [5, 3, 2], opt_args=(0.1, 0.1), inputs=INPUTS.reshape(T, 1),
                                                        debug=True)
    post, aux = bfa.speedy_augmented_gaussian_sum_filter(pp, ys, [5, 3, 2], opt_args=(0.1, 0.1), inputs=INPUTS,
                                                         return_leaf_indices=True)
    idx = np.stack([np.minimum(otf.choice_indices(otf.cumsum_assoc(w), otf.uniform(otf.PRNGKey(0), 5)), 29)
                    for w in raux["pre_weights"]])
    assert np.array_equal(aux["leaf_indices"].cpu().numpy(), idx)
    for k in ("means", "covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, k


def test_augmented_gaussian_sum_filter():                                # :84-87
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    post, aux = bfa.augmented_gaussian_sum_filter(pp, ys, [2, 5, 5], opt_args=(0.1, 0.1), inputs=INPUTS)
    _finite_shapes(post, 2)
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    ref, raux = go.augmented_gaussian_sum_filter(po, ys, [2, 5, 5], opt_args=(0.1, 0.1), inputs=INPUTS.reshape(T, 1), debug=True)
    post, aux = bfa.augmented_gaussian_sum_filter(pp, ys, [2, 5, 5], opt_args=(0.1, 0.1), inputs=INPUTS, return_leaf_indices=True)
    idx = np.stack([np.minimum(otf.choice_indices(otf.cumsum_assoc(w), otf.uniform(otf.PRNGKey(0), 2)), 49)
                    for w in raux["pre_weights"]])
    assert np.array_equal(aux["leaf_indices"].cpu().numpy(), idx)
    for k in ("means", "covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, k


def test_augmented_gaussian_sum_filter_optimal():                        # :89-92
    """The reference's own tree, num_components = [5, 5, 5]: 125 leaves, one 128-thread workgroup per trajectory."""
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    post, aux = bfa.augmented_gaussian_sum_filter_optimal(pp, ys, [5, 5, 5], opt_args=(0.1, 0.1), inputs=INPUTS)
    _finite_shapes(post, 5)
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    for nc in ([5, 5, 5], [5, 2, 2]):
        ref, raux = go.augmented_gaussian_sum_filter_optimal(po, ys, nc, opt_args=(0.1, 0.1), inputs=INPUTS.reshape(T, 1), debug=True)
        post, aux = bfa.augmented_gaussian_sum_filter_optimal(pp, ys, nc, opt_args=(0.1, 0.1), inputs=INPUTS, return_leaf_indices=True)
        assert np.array_equal(aux["leaf_indices"].cpu().numpy(), raux["leaf_indices"]), nc
        for k in ("means", "covariances"):
            assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, (nc, k)


def test_unscented_gaussian_sum_filter():                                # :94-98
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    # ParamsUKF() = (1e-3, 2, 0): L + lambda = 5e-6, sigma-point weights of +-1e5..1e6 -- fp32 cancellation
    # leaves nothing to compare on either side; the call must run and keep the reference's shapes
    post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(), ys, 5, 1, INPUTS)
    _finite_shapes(post, 5)
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    im = bfa.sample_initial_component_means(pp, 5)
    ref = go.unscented_gaussian_sum_filter(po, go.ParamsUKF(1, 0, 0), ys, 5, initial_means=im, inputs=INPUTS.reshape(T, 1))
    post = bfa.unscented_gaussian_sum_filter(pp, bfa.ParamsUKF(1, 0, 0), ys, 5, 1, INPUTS)
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, k


def test_unscented_agsf():                                               # :100-104
    bfa, po, pp = _params(25e-6)
    ys = _data(bfa, pp)
    # ParamsUKF(alpha=1e-3, beta=2, kappa=0) as in the reference: must run and keep the shapes (see above)
    post, aux = bfa.speedy_unscented_agsf(pp, bfa.ParamsUKF(alpha=1e-3, beta=2.0, kappa=0.0), ys, [2, 5, 5],
                                          opt_args=(0.1, 0.1), inputs=INPUTS)
    _finite_shapes(post, 2)
    bfa, po, pp = _params(1e-2)
    ys = _data(bfa, pp).cpu().numpy()
    ref, raux = go.speedy_unscented_agsf(po, go.ParamsUKF(1, 0, 0), ys, [2, 5, 5], opt_args=(0.1, 0.1),
                                         inputs=INPUTS.reshape(T, 1), debug=True)
    post, aux = bfa.speedy_unscented_agsf(pp, bfa.ParamsUKF(1, 0, 0), ys, [2, 5, 5], opt_args=(0.1, 0.1), inputs=INPUTS,
                                          return_leaf_indices=True)
    idx = np.stack([np.minimum(otf.choice_indices(otf.cumsum_assoc(w), otf.uniform(otf.PRNGKey(0), 2)), 49)
                    for w in raux["pre_weights"]])
    assert np.array_equal(aux["leaf_indices"].cpu().numpy(), idx)
    for k in ("means", "covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), getattr(ref, k)) < 5e-5, k


def test_bootstrap_particle_filter_on_the_same_model():
    """Not in the reference's test file, but its experiment runs the particle filter on this model
    (BOT_Experiment_script.py:151): gBOTlp = MVN(gBOT(x, 0, u), R).log_prob(y) (:47)."""
    bfa, po, pp = _params(1e-2)
    nl = bfa.nonlinearities
    ys = _data(bfa, pp)
    bp = bfa.ParamsBPF(*pp, nl.gaussian_log_prob(pp.emission_function, 1e-2 * np.eye(1, dtype=F32)))
    out = bfa.bootstrap_particle_filter(bp, ys, 1000, bfa.PRNGKey(0), INPUTS)
    assert tuple(out["weights"].shape) == (1000, T) and tuple(out["particles"].shape) == (1000, T, 4)
    w = out["weights"].cpu().numpy()
    assert np.allclose(w.sum(axis=0), 1.0, atol=1e-4)


def test_bot_monte_carlo_example_runs():
    """examples/bot_experiment.py (the loop of BOT_Experiment_script.py:89-180 as one batch) at a small size."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "bot_experiment.py"), "--nsim", "8", "--steps", "50",
                          "--components", "8", "--particles", "256"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for name in ("GSF", "U-GSF", "AGSF", "U-AGSF", "BPF"):
        assert any(line.startswith(name) for line in out.stdout.splitlines()), out.stdout
