"""Every compiled (state_dim, obs_dim) instance of the Kalman and Gaussian-sum kernels against the oracle, at ragged
sizes: odd batch, T not a multiple of the time tile, non-square noise maps, biases.  (The other test files go deep on a
few shapes; this one makes sure no instance of the dispatch tables is left unexercised.)  Tolerance 1e-5 relative."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32
TOL = 1e-5
FIELDS = ("means", "covariances", "predicted_means", "predicted_covariances")
DIMS = [(n, m) for n in range(1, 9) for m in range(1, 5) if m <= n and (n, m) not in ((1, 2),)]


@pytest.mark.parametrize("n,m", DIMS)
def test_kalman_instance(n, m):
    import bayesianfiltering_amd as bfa
    dq, dr = max(1, n - (n + m) % 3), max(1, m - (n % 2))
    a = cm.random_stable_lgssm(n, m, seed=100 * n + m, dq=dq, dr=dr, bias=True)
    B, T = 7 + n, 9 + 2 * m + (n % 4)
    ys = cm.simulate_batch(a, B, T, seed=n + m)
    init = np.tile(a["m0"], (B, 1)) + 0.1 * np.arange(B, dtype=F32)[:, None]
    ref = cm.oracle_kalman_batch(a, ys, init)
    for layout in ("reference", "batch_inner"):
        post, ll = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, layout=layout, return_loglik=True)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < TOL, (layout, k)
        assert cm.rel_err(ll.cpu().numpy().reshape(-1), np.asarray(ref["loglik"]).reshape(-1)) < 3e-5, layout


@pytest.mark.parametrize("n,m", DIMS)
def test_gaussian_sum_instance(n, m):
    import bayesianfiltering_amd as bfa
    K = 3 + (n + m) % 3
    a = cm.random_stable_lgssm(n, m, seed=200 * n + m, bias=True)
    B, T = 3, 7 + m + (n % 3)
    ys = cm.simulate_batch(a, B, T, seed=2 * n + m)
    rng = np.random.default_rng(n * 10 + m)
    im = (a["m0"] + 0.5 * rng.normal(size=(B, K, n))).astype(F32)
    po = cm.oracle_params(a)
    post = bfa.gaussian_sum_filter(cm.product_params(a), ys, K, initial_means=im)
    for b in range(B):
        ref = go.gaussian_sum_filter(po, ys[b], K, initial_means=im[b])
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k)[b].cpu().numpy(), getattr(ref, k)) < TOL, (b, k)
        assert np.max(np.abs(post.weights[b].cpu().numpy() - ref.weights)) < 2e-5, b
