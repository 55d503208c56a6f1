"""GPU parity of the batched Kalman kernel (through the C-ABI) against the NumPy oracle.

Tolerance: north_star asks filtered means/covariances within 1e-5 relative in fp32."""
import numpy as np
import pytest

from tests import common as cm

pytestmark = pytest.mark.gpu
TOL = 1e-5
FIELDS = ("weights", "means", "covariances", "predicted_means", "predicted_covariances")


def _run(a, ys, init, layout, mode=-1, lanes=0, **kw):
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_emit_mode", mode))
    _lib.check(lib.bf_set_option(b"kf_lanes", lanes))
    try:
        return bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, layout=layout, **kw)
    finally:
        lib.bf_set_option(b"kf_emit_mode", -1)
        lib.bf_set_option(b"kf_lanes", 0)


# lanes = lanes cooperating on one trajectory (0 = shipping default); mode 2 = LDS-staged stores,
# mode 0 = strided stores
@pytest.mark.parametrize("lanes", [0, 1, 2, 4])
@pytest.mark.parametrize("layout,mode", [("reference", 2), ("reference", 0), ("batch_inner", -1)])
def test_cv_model_matches_oracle(lanes, layout, mode):
    a = cm.cv_model_arrays()
    # ragged last wave (130 = 2*64 + 2); T a multiple of 4 (aligned rows) but not of the staging depth
    B, T = 130, 72
    ys = cm.simulate_batch(a, B, T, seed=1)
    init = np.tile(a["m0"], (B, 1)) + np.random.default_rng(2).normal(size=(B, 4)).astype(np.float32)
    ref = cm.oracle_kalman_batch(a, ys, init)
    post, ll = _run(a, ys, init, layout, mode, lanes, return_loglik=True)
    for k in FIELDS:
        got = getattr(post, k).cpu().numpy()
        assert got.shape == ref[k].shape
        assert cm.rel_err(got, ref[k]) < TOL, k
    assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 2e-5


@pytest.mark.parametrize("n,m,dq,dr", [(1, 1, 1, 1), (2, 1, 2, 1), (2, 2, 1, 2), (3, 1, 3, 1), (3, 3, 3, 3),
                                       (4, 1, 2, 1), (4, 2, 4, 2), (4, 4, 3, 4), (5, 2, 5, 2), (6, 3, 6, 3),
                                       (7, 4, 3, 4), (8, 1, 8, 1), (8, 4, 8, 4)])
def test_random_lgssm_dims(n, m, dq, dr):
    a = cm.random_stable_lgssm(n, m, seed=10 * n + m, dq=dq, dr=dr, bias=True)
    B, T = 70, 48
    ys = cm.simulate_batch(a, B, T, seed=3)
    init = np.tile(a["m0"], (B, 1))
    ref = cm.oracle_kalman_batch(a, ys, init)
    for layout in ("reference", "batch_inner"):
        post = _run(a, ys, init, layout)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < TOL, (layout, k)


def test_single_trajectory_shapes_like_reference():
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, 1, 32, seed=5)[0]
    post = _run(a, ys, a["m0"][None], "reference")
    assert tuple(post.means.shape) == (1, 32, 4) and tuple(post.covariances.shape) == (1, 32, 4, 4)
    assert tuple(post.weights.shape) == (1, 32)
    ref = cm.oracle_kalman_batch(a, ys[None], a["m0"][None])
    assert cm.rel_err(post.predicted_covariances.cpu().numpy(), ref["predicted_covariances"][0]) < TOL


def test_chunked_carry_equals_one_shot():
    import bayesianfiltering_amd as bfa
    a = cm.cv_model_arrays()
    B, T = 64, 96
    ys = cm.simulate_batch(a, B, T, seed=7)
    p = cm.product_params(a)
    full = bfa.kalman_filter(p, ys)
    first, carry = bfa.kalman_filter(p, ys[:, :32], return_carry=True)
    second = bfa.kalman_filter(p, ys[:, 32:], carry=carry)
    for k in FIELDS:
        whole = getattr(full, k).cpu().numpy()
        parts = np.concatenate([getattr(first, k).cpu().numpy(), getattr(second, k).cpu().numpy()], axis=2)
        assert np.array_equal(whole, parts), k     # same arithmetic, chunking must be bit-exact


def test_time_varying_covariances():
    a = cm.cv_model_arrays()
    B, T = 8, 32
    ys = cm.simulate_batch(a, B, T, seed=9)
    rng = np.random.default_rng(0)
    a = dict(a)
    a["Q"] = (a["Q"][None] * (1 + rng.random((T, 1, 1)))).astype(np.float32)
    a["R"] = (a["R"][None] * (1 + rng.random((T, 1, 1)))).astype(np.float32)
    init = np.tile(a["m0"], (B, 1))
    ref = cm.oracle_kalman_batch(a, ys, init)
    post = _run(a, ys, init, "reference")
    for k in FIELDS:
        assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < TOL, k


def test_errors_are_loud():
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    a = cm.random_stable_lgssm(9, 2, seed=1)
    ys = cm.simulate_batch(a, 2, 8, seed=1)
    bfa.kalman_filter(cm.product_params(a), ys)     # n = 9 has no compiled instance: the run-time-dimension kernel runs it
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.kalman_filter(cm.product_params(cm.random_stable_lgssm(200, 3, seed=1)), np.zeros((1, 4, 3), np.float32))
    assert e.value.code == _lib.BF_EUNSUPPORTED
    with pytest.raises(TypeError):
        bfa.kalman_filter(cm.product_params(a)._replace(dynamics_function=lambda x, q, u: x), ys)


def test_mfma_path_n64_m32():
    """cfg5 shape: dense random-stable LGSSM n = 64, m = 32 on the fp32 matrix-core kernel.  The
    C port of the oracle (validated against the NumPy oracle in tests/test_oracle_filters.py) is the
    checker; a few trajectories also run through the NumPy oracle itself."""
    import bayesianfiltering_amd as bfa
    from oracle import c_oracle
    a = cm.random_stable_lgssm(64, 32, seed=64, bias=True)
    a["Q"] = (1e-2 * np.eye(64)).astype(np.float32)
    a["R"] = (1e-1 * np.eye(32)).astype(np.float32)
    B, T = 6, 24
    ys = cm.simulate_batch(a, B, T, seed=9)
    init = np.tile(a["m0"], (B, 1))
    ref = c_oracle.kalman_filter(a, ys, init)
    post, ll, carry = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, return_loglik=True, return_carry=True)
    for k in FIELDS:
        got = getattr(post, k).cpu().numpy()
        assert got.shape == ref[k].shape
        assert cm.rel_err(got, ref[k]) < TOL, (k, cm.rel_err(got, ref[k]))
    assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 2e-5
    ref_np = cm.oracle_kalman_batch(a, ys[:1], init[:1])
    assert cm.rel_err(post.covariances[:1].cpu().numpy(), ref_np["covariances"]) < TOL
    # the carry continues the scan bit for bit
    p1, c1 = bfa.kalman_filter(cm.product_params(a), ys[:, :10], initial_means=init, return_carry=True)
    p2 = bfa.kalman_filter(cm.product_params(a), ys[:, 10:], carry=c1)
    whole = post.predicted_covariances.cpu().numpy()
    parts = np.concatenate([p1.predicted_covariances.cpu().numpy(), p2.predicted_covariances.cpu().numpy()], axis=2)
    assert np.array_equal(whole, parts)


def test_golden_kalman_fixtures(golden_dir):
    """The committed golden vectors (tests/golden/make_golden.py) through the C-ABI."""
    import bayesianfiltering_amd as bfa
    for name in ("kalman_cv_n4_m2_T64", "kalman_random_n3_m3_T40"):
        d = np.load(f"{golden_dir}/{name}.npz")
        a = {k: d[k] for k in ("A", "G", "H", "D", "Q", "R", "m0", "P0", "q0", "r0")}
        post, ll = bfa.kalman_filter(cm.product_params(a), d["emissions"], initial_means=d["initial_means"], return_loglik=True)
        for k in FIELDS:
            assert cm.rel_err(getattr(post, k).cpu().numpy(), d["out_" + k]) < TOL, (name, k)
        assert cm.rel_err(ll.cpu().numpy(), d["out_loglik"]) < 2e-5


@pytest.mark.parametrize("T", [50, 73 * 2, 21 * 4 + 2])
def test_staged_path_with_rows_of_the_scalar_streams_unaligned(T):
    """T not a multiple of 4: the weight / log-likelihood rows are not 16-byte aligned and go out as dword
    stores while means and covariances keep the LDS-staged path (forcing the staged emitter must succeed)."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    a = cm.cv_model_arrays()
    p = cm.product_params(a)
    B = 64
    ys = cm.simulate_batch(a, B, T, seed=T)
    init = np.tile(a["m0"], (B, 1)).astype(np.float32)
    ref = cm.oracle_kalman_batch(a, ys, init)
    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_emit_mode", 2))
    try:
        post, ll = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True)
    finally:
        _lib.check(lib.bf_set_option(b"kf_emit_mode", -1))
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < 1e-5, k
    assert np.allclose(post.weights.cpu().numpy(), 1.0)
    assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 3e-5


@pytest.mark.parametrize("n,m", [(3, 1), (3, 2), (5, 2), (6, 3), (7, 4), (7, 1)])
@pytest.mark.parametrize("T", [40, 37])
def test_staged_stores_for_non_power_of_two_state_dims(n, m, T):
    """The LDS time-transpose emitter (128-byte-plus rows, dwordx4 stores) for n = 3, 5, 6, 7: rows of 36 / 100 / 36 / 196
    floats (a whole number of steps that is also a multiple of 16 bytes), padding lanes masked.  Forced on
    (kf_emit_mode = 2 fails loudly if the layout is not eligible), compared bit for bit with the strided emitter and to
    1e-5 with the oracle; ragged batch (the tail of the batch goes through the strided kernel) and, for T = 37, rows that
    are not complete at the end."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    a = cm.random_stable_lgssm(n, m, seed=10 * n + m, bias=True)
    B = 150
    ys = cm.simulate_batch(a, B, T, seed=n)
    init = np.tile(a["m0"], (B, 1))
    p = cm.product_params(a)
    lib = _lib.load()
    try:
        if (T * n) % 4 == 0 and (T * n * n) % 4 == 0:
            _lib.check(lib.bf_set_option(b"kf_emit_mode", 2))
        staged, lls = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True)
        _lib.check(lib.bf_set_option(b"kf_emit_mode", 0))
        strided, llt = bfa.kalman_filter(p, ys, initial_means=init, return_loglik=True)
    finally:
        lib.bf_set_option(b"kf_emit_mode", -1)
    for k in bfa.FULL5:
        assert torch.equal(getattr(staged, k), getattr(strided, k)), k
    assert torch.equal(lls, llt)
    ref = cm.oracle_kalman_batch(a, ys, init)
    for k in FIELDS:
        assert cm.rel_err(getattr(staged, k).cpu().numpy(), ref[k]) < 1e-5, k


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
def test_every_variant_of_the_matrix_core_kernel(variant):
    """bf_set_option("kf_mfma_variant"): the (64, 32) kernel's selectable forms -- round 1's explicit inverse (1), the
    gain-free update on fp32 MFMAs with the factorization in registers (2), as rank-2 MFMAs (3), at three workgroups per
    CU (4), and the default with the products as three-term bf16 splits (5) -- all against the oracle at 1e-5."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    from oracle import c_oracle
    a = cm.random_stable_lgssm(64, 32, seed=640 + variant, bias=True)
    B, T = 3, 30
    ys = cm.simulate_batch(a, B, T, seed=variant)
    init = np.tile(a["m0"], (B, 1))
    ref = c_oracle.kalman_filter(a, ys, init)
    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_mfma_variant", variant))
    try:
        post, ll = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init, return_loglik=True)
    finally:
        _lib.check(lib.bf_set_option(b"kf_mfma_variant", 5))
    for k in ("means", "covariances", "predicted_means", "predicted_covariances"):
        assert cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]) < 1e-5, (variant, k, cm.rel_err(getattr(post, k).cpu().numpy(), ref[k]))
    assert cm.rel_err(ll.cpu().numpy(), ref["loglik"]) < 5e-5
