"""The HIP engine against numbers the REFERENCE ITSELF produced (stored notebook outputs, see
tests/test_reference_recorded_outputs.py and tests/golden/extract_reference_outputs.py): every step -- data generation
included -- runs on the device through the C-ABI, with the notebook's seeds; nothing from the oracle is involved.
Tolerance 5e-5 relative on the printed RMSE (fp32 reassociation over 100 steps of a chaotic model).  The 100-particle
filter of test_single_run.ipynb is held to the same bound.  The 5e5-particle filter of Experiment_TSP_2023.ipynb is
not reproducible to that level by any fp32 implementation other than the very same XLA build: its CDF steps (2e-6) are
only ~30 ulps of the cumulative sum wide, so last-bit differences in exp / erf_inv move a few percent of the draws to
a neighbouring particle at every resampling and the clouds decorrelate.  What can be asserted is agreement within the
filter's own Monte-Carlo spread on this (bimodal: the emission is |x|^2) problem: the RMSE's standard deviation over
independent keys is 0.33 / 0.38 for the two simulations (scripts/bpf_mc_spread_probe.py; ESS drops to ~8e3), the engine's
values with the notebook's key are 27.229 / 32.454 against the recorded 27.188 / 32.112; bound 3 standard deviations
(3.5e-2 relative)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32
TOL = 5e-5


def rmse(est, base):                        # gaussfiltax/utils.py:184-187
    est, base = np.asarray(est, np.float64), np.asarray(base, np.float64)
    return float(np.sqrt(np.sum((est - base) ** 2) / est.shape[0]))


@pytest.fixture(scope="module")
def recorded(golden_dir):
    return json.load(open(golden_dir + "/reference_notebook_outputs.json"))


def _close(got, ref, what, tol=TOL):
    if np.isnan(ref):
        assert np.isnan(got), (what, got)
    else:
        assert abs(got - ref) <= tol * ref, (what, got, ref)


def test_experiment_tsp_notebook(recorded):
    """Experiment_TSP_2023.ipynb cell 6, simulations 1 and 2: GSF 24.11498 / 35.695778, UGSF nan / 32.37542,
    BPF with 5e5 particles 27.18829 / 32.11221."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import legacy
    nl = bfa.nonlinearities
    ref = recorded["tsp"]["rmse"]
    eye3 = np.eye(3, dtype=F32)
    f, g = nl.lorenz63(), nl.quadratic(3, 0.001)
    R = F32(0.1) * np.eye(1, dtype=F32)
    params = bfa.ParamsNLSSM(np.zeros(3, F32), eye3, f, np.zeros(3, F32), F32(20.0) * eye3, g, np.zeros(1, F32), R)
    params_bpf = bfa.ParamsBPF(np.zeros(3, F32), eye3, f, np.zeros(3, F32), F32(20.0) * eye3, g, np.zeros(1, F32), R,
                               nl.gaussian_log_prob(g, R))
    model = bfa.NonlinearSSM(3, 3, 1, 1)
    next_key = bfa.PRNGKey(0)
    for i in range(len(ref["GSF"])):
        key, next_key = legacy._split(next_key, 2)
        states, emissions = model.sample(params, key, 100)
        states = states.cpu().numpy()
        post = bfa.gaussian_sum_filter(params, emissions, 2, 1)
        pe = (post.weights.unsqueeze(-1) * post.means).sum(dim=0).cpu().numpy()
        _close(rmse(pe, states), ref["GSF"][i], ("GSF", i))
        post = bfa.unscented_gaussian_sum_filter(params, bfa.ParamsUKF(1, 0, 0), emissions, 2, 1)
        pe = (post.weights.unsqueeze(-1) * post.means).sum(dim=0).cpu().numpy()
        _close(rmse(pe, states), ref["UGSF"][i], ("UGSF", i))
        out = bfa.bootstrap_particle_filter(params_bpf, emissions, 500000, key, output="summary")
        _close(rmse(out["mean"].cpu().numpy(), states), ref["BPF"][i], ("BPF", i), tol=3.5e-2)


def test_single_run_notebook(recorded):
    """test_single_run.ipynb cell 6: GSF RMSE nan, BPF (100 particles, ess 0.5) RMSE 0.7464309; cell 9: weights[:, 16]."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import legacy
    nl = bfa.nonlinearities
    rec = recorded["single_run"]
    mu0 = np.array([-0.05, 0.001, 0.7, -0.05], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = F32(1e-6) * np.eye(2, dtype=F32), F32(25 * 1e-6) * np.eye(2, dtype=F32)
    f, g = nl.maneuver_bot(), nl.bearing_range()
    params = bfa.ParamsNLSSM(mu0, S0, f, np.zeros(2, F32), Q, g, np.zeros(2, F32), R)
    params_bpf = bfa.ParamsBPF(mu0, S0, f, np.zeros(2, F32), Q, g, np.zeros(2, F32), R, nl.gaussian_log_prob(g, R))
    inputs = np.array([1] * 10 + [0] * 10 + [2] * 10, F32)
    states, emissions = bfa.NonlinearSSM(4, 2, 2, 2).sample(params, np.array(rec["settings"]["sample_key"], np.uint32), 30,
                                                           inputs=inputs)
    states = states.cpu().numpy()
    post = bfa.gaussian_sum_filter(params, emissions, 5, 1, inputs)
    pe = (post.weights.unsqueeze(-1) * post.means).sum(dim=0).cpu().numpy()
    _close(rmse(pe, states), rec["rmse"]["GSF"], "GSF")
    key = legacy._split(np.array(rec["settings"]["next_key"], np.uint32), 2)[0]
    out = bfa.bootstrap_particle_filter(params_bpf, emissions, 100, key, inputs, 0.5)
    w, x = out["weights"].cpu().numpy(), out["particles"].cpu().numpy()
    _close(rmse(np.einsum("ntd,nt->td", x, w), states), rec["rmse"]["BPF"], "BPF")
    w16 = np.array(rec["bpf_weights_t16"], F32)
    assert np.array_equal(w[:len(w16), 16], w16)
    x16 = np.array(rec["bpf_particles_t16"]).reshape(100, 4)      # the printed cloud at t = 16, slot by slot
    assert np.max(np.abs(x[:, 16] - x16)) < 1e-6


def test_autocov_sims_matrix_from_library_normals(recorded):
    """autocov_sims.ipynb cell 2 (see the CPU test of the same name) replayed on the LIBRARY's normal draws
    (bf_random_normal_f32: the Threefry / erf_inv code the kernels inline)."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import inference
    from tests import common as cm
    X = cm.autocov_sims_replay(inference._random_normal(bfa.PRNGKey(0), 30))
    assert np.max(np.abs(X - np.array(recorded["autocov"]["X"]))) < 5e-7, X
