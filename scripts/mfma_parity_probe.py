"""Ad-hoc: errors of the (64, 32) Kalman kernel variants against the oracle's C port on configs[4]'s model, T = 2 000."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
from tests import common as cm
from oracle import c_oracle
F32 = np.float32
a = cm.random_stable_lgssm(64, 32, seed=64)
a["Q"] = (1e-2 * np.eye(64)).astype(F32); a["R"] = (1e-1 * np.eye(32)).astype(F32); a["m0"] = np.zeros(64, F32)
p = cm.product_params(a)
T = 2000
y = cm.device_observations(p, (64, 64, 32, 32), 2, T, seed=5)
init = np.zeros((2, 64), F32)
ref = c_oracle.kalman_filter(a, y.cpu().numpy(), init)
for v in (2, 3, 5):
    _lib.check(_lib.require_gpu().bf_set_option(b"kf_mfma_variant", v))
    post, ll = bfa.kalman_filter(p, y, initial_means=init, return_loglik=True)
    e = {k: (cm.rel_err(getattr(post, k).cpu().numpy()[:, :, :300], ref[k][:, :, :300]), cm.rel_err(getattr(post, k).cpu().numpy(), ref[k])) for k in bfa.FULL5}
    print("variant", v, " ".join(f"{k}: {e[k][0]:.2e}/{e[k][1]:.2e}" for k in e), f"loglik {cm.rel_err(ll.cpu().numpy(), ref['loglik']):.2e}", flush=True)
