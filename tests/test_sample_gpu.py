"""GPU parity of the device data generator (bf_sample_ssm_f32) against the oracle's restatement of
NonlinearSSM.sample (gaussfiltax/models.py:240-289)."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32


def test_sample_matches_oracle_for_registry_models():
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T = 25
    cases = []
    a = cm.cv_model_arrays()
    cases.append((cm.oracle_params(a), cm.product_params(a), (4, 2, 2, 2), None))
    cases.append((go.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), om.Lorenz96(8), np.zeros(8, F32), 1e-2 * np.eye(8, dtype=F32),
                                 om.PickEven(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32)),
                  bfa.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32), 1e-2 * np.eye(8, dtype=F32),
                                  nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32)), (8, 8, 4, 4), None))
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 9, F32)
    mu0, S0 = np.array([2.0, 0.3, 3.0, -0.2], F32), np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    cases.append((go.ParamsNLSSM(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R),
                  bfa.ParamsNLSSM(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R),
                  (4, 2, 2, 2), inputs))
    r0 = np.array([0.1, -0.2, 0.05], F32)
    u2 = np.array([0] * 12 + [1] * 13, F32)
    cases.append((go.ParamsNLSSM(np.zeros(3, F32), np.eye(3, dtype=F32), om.Linear(0.8 * np.eye(3, dtype=F32)), np.zeros(3, F32),
                                 0.5 * np.eye(3, dtype=F32), om.StochVol(3), r0, 0.1 * np.eye(3, dtype=F32)),
                  bfa.ParamsNLSSM(np.zeros(3, F32), np.eye(3, dtype=F32), nl.linear_dynamics(0.8 * np.eye(3, dtype=F32)), np.zeros(3, F32),
                                  0.5 * np.eye(3, dtype=F32), nl.stoch_vol(3), r0, 0.1 * np.eye(3, dtype=F32)), (3, 3, 3, 3), u2))
    for po, pp, dims, u in cases:
        model = bfa.NonlinearSSM(*dims)
        keys = np.stack([otf.PRNGKey(11), otf.PRNGKey(12), otf.split(otf.PRNGKey(1), 3)[2]])
        xs, ys = model.sample(pp, keys, T, u)
        assert tuple(xs.shape) == (3, T, dims[0]) and tuple(ys.shape) == (3, T, dims[2])
        for b in range(3):
            xr, yr = go.sample_ssm(po, keys[b], T, None if u is None else u.reshape(T, 1))
            assert cm.rel_err(xs[b].cpu().numpy(), xr) < 2e-5, type(po.dynamics_function).__name__
            assert cm.rel_err(ys[b].cpu().numpy(), yr) < 2e-5, type(po.emission_function).__name__
        x1, y1 = model.sample(pp, keys[0], T, u)            # single key -> reference shapes (T, n), (T, m)
        assert tuple(x1.shape) == (T, dims[0]) and np.array_equal(x1.cpu().numpy(), xs[0].cpu().numpy())


def test_library_split_reproduces_keys_recorded_by_the_reference(golden_dir):
    """bf_random_split (the library's Threefry, the same function the kernels inline) against the 20 keys the
    reference's own notebook run printed: PRNGKey(1), then key0, key, next_key = split(next_key, 3), ten times."""
    import json
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import legacy
    d = json.load(open(golden_dir + "/reference_notebook_keys.json"))
    nk = bfa.PRNGKey(d["seed"])
    for k0_ref, k_ref in zip(d["key0"], d["key"]):
        k0, k, nk = legacy._split(nk, 3)
        assert k0.tolist() == k0_ref and k.tolist() == k_ref
