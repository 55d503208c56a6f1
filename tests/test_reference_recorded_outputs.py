"""The oracle against numbers the REFERENCE ITSELF produced: the stored cell outputs of its notebooks
(tests/golden/reference_notebook_outputs.json, extracted by tests/golden/extract_reference_outputs.py).

Each case re-runs, with the NumPy restatement, exactly what the notebook cell ran -- same seeds, same key chain, same
model, data drawn by the restated NonlinearSSM.sample -- and compares the RMSE the notebook printed.  A match needs the
whole chain to agree with the reference's JAX run: PRNGKey / split, normal draws (bits -> uniform -> erf_inv), the
MVN sampler layout, the data generator, the PRNGKey(0) draw of the initial component means, the EKF / UKF / particle
recursions with their quirks, the multinomial resampler and the point estimate.  Tolerance: 2e-5 relative on an RMSE
over 100 (30) steps; measured 1e-7 ... 3e-6."""
import json
import warnings

import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as tf

F32 = np.float32
TOL = 2e-5


def rmse(est, base):                        # gaussfiltax/utils.py:184-187
    return float(np.sqrt(np.sum((est - base) ** 2) / est.shape[0]))


def point_estimate(means, weights):         # jnp.sum(jnp.einsum('ijk,ij->ijk', means, weights), axis=0)
    return np.einsum("ktn,kt->tn", means, weights)


@pytest.fixture(scope="module")
def recorded(golden_dir):
    return json.load(open(golden_dir + "/reference_notebook_outputs.json"))


def tsp_model():
    eye3 = np.eye(3, dtype=F32)
    f, g = om.Lorenz63(), om.Quadratic(3, 0.001)
    R = F32(0.1) * np.eye(1, dtype=F32)
    p = go.ParamsNLSSM(np.zeros(3, F32), eye3, f, np.zeros(3, F32), F32(20.0) * eye3, g, np.zeros(1, F32), R)
    return p, go.ParamsBPF(*p, go.GaussianEmissionLogProb(g, R))


def single_run_model():
    mu0 = np.array([-0.05, 0.001, 0.7, -0.05], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = F32(1e-6) * np.eye(2, dtype=F32), F32(25 * 1e-6) * np.eye(2, dtype=F32)
    f, g = om.ManeuverBOT(), om.BearingRange()
    p = go.ParamsNLSSM(mu0, S0, f, np.zeros(2, F32), Q, g, np.zeros(2, F32), R)
    return p, go.ParamsBPF(*p, go.GaussianEmissionLogProb(g, R))


def test_experiment_tsp_gsf_and_ugsf(recorded):
    """Experiment_TSP_2023.ipynb cell 6, simulations 1 and 2: GSF 24.11498 / 35.695778, UGSF nan / 32.37542."""
    ref = recorded["tsp"]["rmse"]
    params, _ = tsp_model()
    next_key = tf.PRNGKey(0)
    for i in range(len(ref["GSF"])):
        key, next_key = tf.split(next_key, 2)
        states, emissions = go.sample_ssm(params, key, 100, None)
        post = go.gaussian_sum_filter(params, emissions, 2, 1, None)
        got = rmse(point_estimate(post.means, post.weights), states)
        assert abs(got - ref["GSF"][i]) <= TOL * ref["GSF"][i], (i, got, ref["GSF"][i])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            post = go.unscented_gaussian_sum_filter(params, go.ParamsUKF(1, 0, 0), emissions, 2, 1, None)
        got = rmse(point_estimate(post.means, post.weights), states)
        if np.isnan(ref["UGSF"][i]):
            assert np.isnan(got), (i, got)
        else:
            assert abs(got - ref["UGSF"][i]) <= TOL * ref["UGSF"][i], (i, got, ref["UGSF"][i])


def test_single_run_gsf_nan_and_particle_filter(recorded):
    """test_single_run.ipynb cell 6: GSF RMSE nan, BPF (100 particles) RMSE 0.7464309; cell 9: weights[:, 16] all 0.01."""
    rec = recorded["single_run"]
    params, params_bpf = single_run_model()
    inputs = np.array([1] * 10 + [0] * 10 + [2] * 10, F32)
    states, emissions = go.sample_ssm(params, np.array(rec["settings"]["sample_key"], np.uint32), 30, inputs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        post = go.gaussian_sum_filter(params, emissions, 5, 1, inputs)
    assert np.isnan(rec["rmse"]["GSF"]) and np.isnan(rmse(point_estimate(post.means, post.weights), states))
    key = tf.split(np.array(rec["settings"]["next_key"], np.uint32), 2)[0]
    out = go.bootstrap_particle_filter(params_bpf, emissions, 100, key=key, inputs=inputs, ess_threshold=0.5)
    got = rmse(np.einsum("ntd,nt->td", out["particles"], out["weights"]), states)
    assert abs(got - rec["rmse"]["BPF"]) <= TOL * rec["rmse"]["BPF"], (got, rec["rmse"]["BPF"])
    w16 = np.array(rec["bpf_weights_t16"], F32)
    assert np.array_equal(out["weights"][:len(w16), 16], w16)
    # the printed cloud at t = 16: every slot holds the reference's particle (ancestry of 17 resampling-gated steps)
    x16 = np.array(rec["bpf_particles_t16"]).reshape(100, 4)
    assert np.max(np.abs(out["particles"][:, 16] - x16)) < 1e-6
    assert len(np.unique(x16, axis=0)) == 2


def test_autocov_sims_matrix_pins_the_normal_draws(recorded):
    """autocov_sims.ipynb cell 2 printed the matrix it computes from ``jrandom.multivariate_normal(PRNGKey(0), ones(3),
    eye(3), (10,))``: replaying the cell on the oracle's 30 normal draws (bits -> uniform -> erf_inv, (10, 3) layout)
    reproduces all 8 printed digits."""
    from tests import common as cm
    X = cm.autocov_sims_replay(tf.normal(tf.PRNGKey(0), 30))
    assert np.max(np.abs(X - np.array(recorded["autocov"]["X"]))) < 5e-7, X
    # and the draws matter: another key is far from the recorded matrix
    assert np.max(np.abs(cm.autocov_sims_replay(tf.normal(tf.PRNGKey(1), 30)) - np.array(recorded["autocov"]["X"]))) > 1e-3
