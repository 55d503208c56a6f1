"""GPU parity of the bootstrap particle filter kernel (bf_bpf_f32) and of the stand-alone
resampler (bf_resample_f32) against the NumPy oracle and the golden fixtures.

`north_star`: "bit-exact resampling indices for fixed RNG".  The weight path's arithmetic is DEFINED
(oracle/fp32.py <-> csrc/bf_canon_math.hpp: IEEE operations in a fixed order, no hardware transcendentals), so
for every model built from IEEE operations, exp and log the engine and the oracle (arith="canonical") agree
on EVERY BIT of every weight, particle and ancestor index at every step -- asserted with array_equal below for
N = 100 ... 70 000 and T up to 30.  Models with sin / cos / atan2 (manoeuvring target, bearings) keep libm's
functions on both sides and are compared to rounding."""
import numpy as np
import pytest

from oracle import gaussfilt_oracle as go, models as om, threefry as otf
from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32


def _bfa():
    import bayesianfiltering_amd as bfa
    return bfa, bfa.nonlinearities


@pytest.mark.parametrize("N", [1, 2, 13, 64, 100, 256, 1000, 1024, 4096])
@pytest.mark.parametrize("resampler", ["multinomial", "systematic"])
def test_resample_indices_bit_exact(N, resampler):
    bfa, nl = _bfa()
    rng = np.random.default_rng(N)
    B = 5
    w = rng.random((B, N)).astype(F32) ** 3
    w[1] = 0; w[1, N // 2] = 1.0            # degenerate
    w[2, : N // 2] = 0                      # zero head
    w = (w / w.sum(axis=1, keepdims=True)).astype(F32)
    keys = np.stack([otf.split(otf.PRNGKey(7), B)[b] for b in range(B)])
    got = bfa.resample_indices(w, keys, resampler).cpu().numpy()
    for b in range(B):
        if resampler == "multinomial":
            ref = np.minimum(otf.choice(keys[b], w[b]), N - 1)
        else:
            ref = go.systematic_indices(w[b], otf.uniform(keys[b], 1)[0])
        assert np.array_equal(got[b], ref), (b, np.flatnonzero(got[b] != ref)[:5])


def _l63_params(bfa, nl):
    R = 0.5 * np.eye(3, dtype=F32)
    h = nl.linear_emission(np.eye(3, dtype=F32))
    return bfa.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(), np.zeros(3, F32),
                         0.1 * np.eye(3, dtype=F32), h, np.zeros(3, F32), R, nl.gaussian_log_prob(h, R))


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, F32).view(np.uint32), np.ascontiguousarray(b, F32).view(np.uint32))


@pytest.mark.parametrize("op", [0, 1, 2])
def test_device_canonical_arithmetic_equals_oracle_bit_for_bit(op):
    """canon_log / canon_exp / the bits -> normal map evaluated ON THE DEVICE (bf_canon_eval_f32) against oracle/fp32.py."""
    import ctypes as C
    import torch
    from oracle import fp32
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    rng = np.random.default_rng(op)
    if op == 0:
        x = np.concatenate([np.array([0.0, 1.0, 2.0, 0.5, 0.70710677, 0.7071068, 1e-38, 1e-41, 3.4e38, np.inf], F32),
                            np.exp(rng.uniform(-87, 88, 1 << 20)).astype(F32), (1 + rng.uniform(-0.3, 0.42, 1 << 19)).astype(F32)])
        ref = fp32.canon_log(x)
    elif op == 1:
        x = np.concatenate([np.array([0.0, -0.0, -86.0, -86.00001, 88.0, 88.00001, -200.0, 200.0], F32),
                            rng.uniform(-90, 90, 1 << 20).astype(F32), rng.uniform(-1, 1, 1 << 19).astype(F32)])
        ref = fp32.canon_exp(x)
    else:
        bits = np.concatenate([np.array([0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xFFFFFE00, 0x1FF], np.uint32),
                               rng.integers(0, 2 ** 32, 1 << 21, dtype=np.uint64).astype(np.uint32)])
        x, ref = bits.view(F32), fp32.bits_to_normal(bits)
    xin = torch.as_tensor(x.view(np.int32), device="cuda")
    out = torch.empty(x.size, dtype=torch.float32, device="cuda")
    _lib.check(lib.bf_canon_eval_f32(op, C.c_void_p(xin.data_ptr()), x.size, C.c_void_p(out.data_ptr()), 1, None))
    assert _bits_equal(out.cpu().numpy(), ref)


def test_golden_lorenz63_fixture(golden_dir):
    """The committed canonical-arithmetic run (tests/golden/make_golden.py: bpf_small): every bit of it."""
    bfa, nl = _bfa()
    d = np.load(f"{golden_dir}/bpf_lorenz63_N64_T16_canonical.npz")
    out = bfa.bootstrap_particle_filter(_l63_params(bfa, nl), d["emissions"], 64, d["key"], return_ancestors=True,
                                        output="both")
    assert tuple(out["weights"].shape) == (64, 16) and tuple(out["particles"].shape) == (64, 16, 3)   # (N,T,..) :1378
    assert np.array_equal(out["resampled"].cpu().numpy() > 0.5, d["resampled"])
    assert np.array_equal(out["ancestors"].cpu().numpy().T, d["ancestors"])      # bit-exact indices, all 16 steps
    assert _bits_equal(out["particles"].cpu().numpy(), d["particles"])
    assert _bits_equal(out["weights"].cpu().numpy(), d["weights"])
    assert _bits_equal(out["ess"].cpu().numpy(), d["ess"])
    # the libm-arithmetic oracle run of the same model (the round-1 fixture) agrees to rounding
    d0 = np.load(f"{golden_dir}/bpf_lorenz63_N64_T16.npz")
    assert cm.rel_err(out["particles"].cpu().numpy(), d0["particles"]) < 1e-5
    assert np.max(np.abs(out["weights"].cpu().numpy() - d0["weights"])) < 1e-6
    # weighted mean summary == einsum over the emitted particles / weights (BOT_Experiment_script.py:152)
    mean = np.einsum("itd,it->td", out["particles"].cpu().numpy(), out["weights"].cpu().numpy())
    assert cm.rel_err(out["mean"].cpu().numpy(), mean) < 1e-5


@pytest.mark.parametrize("N,resampler,T", [(100, "multinomial", 30), (256, "systematic", 12), (1000, "multinomial", 30),
                                           (1024, "multinomial", 6), (4096, "multinomial", 30), (4096, "systematic", 6)])
def test_lorenz96_ancestry_bit_exact(N, resampler, T):
    """n = 8, m = 4 Lorenz-96 with the g96lp log-density (gaussfiltax/nonlinearities.py:37-52), B = 2: the WHOLE ancestry
    (inference.py:1350-1357, utils.py:207-214), every weight and every particle of every step equal to the oracle's bits."""
    bfa, nl = _bfa()
    R = 0.5 * np.eye(4, dtype=F32)
    Q = 1e-1 * np.eye(8, dtype=F32)
    po = go.ParamsBPF(8 * np.ones(8, F32), np.eye(8, dtype=F32), om.Lorenz96(8), np.zeros(8, F32), Q, om.PickEven(8),
                      np.zeros(4, F32), R, go.GaussianEmissionLogProb(om.PickEven(8), R))
    g96 = nl.pick_even(8)
    pp = bfa.ParamsBPF(8 * np.ones(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32), Q, g96,
                       np.zeros(4, F32), R, nl.gaussian_log_prob(g96, R))
    ys = np.stack([go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(b), T)[1] for b in range(2)])
    key = np.array([0, 11], np.uint32)
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, resampler=resampler, return_ancestors=True, output="both")
    for b in range(2):
        ref, dbg = go.bootstrap_particle_filter(po, ys[b], N, key=key, resampler=resampler, debug=True, arith="canonical")
        assert dbg["resampled"].any()
        assert np.array_equal(out["resampled"][b].cpu().numpy() > 0.5, dbg["resampled"])
        assert np.array_equal(out["ancestors"][b].cpu().numpy().T, dbg["ancestors"])
        assert _bits_equal(out["weights"][b].cpu().numpy(), ref["weights"])
        assert _bits_equal(out["particles"][b].cpu().numpy(), ref["particles"])
        assert _bits_equal(out["ess"][b].cpu().numpy(), dbg["ess"])
        # ... and the libm-arithmetic oracle (the reference leaves exp / log1p to XLA) agrees to rounding until a draw
        # of ITS run lands within an ulp of a CDF step
        ref0, dbg0 = go.bootstrap_particle_filter(po, ys[b], N, key=key, resampler=resampler, debug=True)
        # (with thousands of particles that can be the very first resampling: a handful of neighbouring-index flips)
        bad = np.flatnonzero((dbg["ancestors"] != dbg0["ancestors"]).any(axis=1))
        t_ok = T if bad.size == 0 else int(bad[0])
        assert t_ok >= (2 if N <= 1024 else 0)
        if t_ok < T:
            assert (dbg["ancestors"][t_ok] != dbg0["ancestors"][t_ok]).mean() < 2e-3
        assert np.max(np.abs(dbg["pre_weights"][0] - dbg0["pre_weights"][0])) < 1e-6
        if t_ok:
            assert cm.rel_err(out["particles"][b].cpu().numpy()[:, :t_ok], ref0["particles"][:, :t_ok]) < 1e-5
            assert np.max(np.abs(out["weights"][b].cpu().numpy()[:, :t_ok] - ref0["weights"][:, :t_ok])) < 1e-6


def test_full_covariances_and_biases_ancestry_bit_exact():
    """Linear dynamics with a non-square noise map, full Q / R / P0 (Cholesky factors with off-diagonal entries, the
    forward substitution of the log-density), noise biases: N = 300, T = 20, all bits."""
    bfa, nl = _bfa()
    a = cm.random_stable_lgssm(4, 2, seed=3, dq=2, dr=2, bias=True)
    N, T = 300, 20
    Rlp = (a["D"] @ a["R"] @ a["D"].T).astype(F32)
    fo, ho = om.Linear(a["A"], a["G"]), om.Linear(a["H"], a["D"])
    po = go.ParamsBPF(a["m0"], a["P0"], fo, a["q0"], a["Q"], ho, a["r0"], a["R"], go.GaussianEmissionLogProb(ho, Rlp, a["r0"]))
    f, h = nl.linear_dynamics(a["A"], a["G"]), nl.linear_emission(a["H"], a["D"])
    pp = bfa.ParamsBPF(a["m0"], a["P0"], f, a["q0"], a["Q"], h, a["r0"], a["R"], nl.gaussian_log_prob(h, Rlp, a["r0"]))
    ys = cm.simulate_batch(a, 1, T, seed=9)[0]
    key = otf.PRNGKey(21)
    ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, debug=True, arith="canonical")
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, return_ancestors=True, output="both")
    assert dbg["resampled"].any() and not dbg["resampled"].all()
    assert np.array_equal(out["ancestors"].cpu().numpy().T, dbg["ancestors"])
    assert _bits_equal(out["weights"].cpu().numpy(), ref["weights"]) and _bits_equal(out["particles"].cpu().numpy(), ref["particles"])


def test_bot_model_with_inputs_and_carry_chunks():
    bfa, nl = _bfa()
    T, N = 12, 64
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    inputs = np.array([1] * 4 + [0] * 4 + [2] * 4, F32)
    po = go.ParamsBPF(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R,
                      go.GaussianEmissionLogProb(om.BearingRange(), R))
    g = nl.bearing_range()
    pp = bfa.ParamsBPF(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, g, np.zeros(2, F32), R, nl.gaussian_log_prob(g, R))
    xs, ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(3), T, inputs.reshape(T, 1))
    key = np.array([0, 5], np.uint32)
    ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=inputs.reshape(T, 1), debug=True, arith="canonical")
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, inputs, return_ancestors=True)
    assert np.array_equal(out["ancestors"].cpu().numpy().T, dbg["ancestors"])
    assert _bits_equal(out["particles"].cpu().numpy(), ref["particles"])
    # chunked run through the carry == one shot, bit for bit
    o1, c1 = bfa.bootstrap_particle_filter(pp, ys[:5], N, key, inputs[:5], return_carry=True)
    o2 = bfa.bootstrap_particle_filter(pp, ys[5:], N, key, inputs[5:], carry=c1)
    whole = out["particles"].cpu().numpy()
    parts = np.concatenate([o1["particles"].cpu().numpy(), o2["particles"].cpu().numpy()], axis=1)
    assert np.array_equal(whole, parts)


@pytest.mark.parametrize("op", [3, 4, 5])
def test_device_canonical_trig_equals_oracle_bit_for_bit(op):
    """canon_sincos / canon_atan2 evaluated ON THE DEVICE against oracle/fp32.py (sin, cos: |x| up to 8192; atan2: every
    quadrant, zeros, huge and tiny ratios)."""
    import ctypes as C
    import torch
    from oracle import fp32
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    rng = np.random.default_rng(op)
    if op in (3, 4):
        x = np.concatenate([np.array([0.0, -0.0, 1e-30, 0.78539816, 0.7853982, 1.5707964, 3.1415927, -3.1415927, 8192.0], F32),
                            (rng.normal(size=1 << 20) * 20).astype(F32), (rng.normal(size=1 << 18) * 1e-3).astype(F32),
                            rng.uniform(-8192, 8192, 1 << 18).astype(F32)])
        ref = fp32.sincos(x)[op - 3]
        xin, n = x, x.size
    else:
        y = np.concatenate([np.array([0.0, 0.0, 1.0, -1.0, 0.0, 1e-30, 1e30], F32), rng.normal(size=1 << 20).astype(F32),
                            (rng.normal(size=1 << 18) * 1e3).astype(F32)])
        xx = np.concatenate([np.array([1.0, -1.0, 0.0, 0.0, 0.0, 1e30, 1e-30], F32), rng.normal(size=1 << 20).astype(F32),
                             rng.normal(size=1 << 18).astype(F32)])
        ref = fp32.atan2(y, xx)
        xin, n = np.concatenate([y, xx]), y.size
    din = torch.as_tensor(xin, device="cuda")
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(lib.bf_canon_eval_f32(op, C.c_void_p(din.data_ptr()), n, C.c_void_p(out.data_ptr()), 1, None))
    assert _bits_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("N,resampler", [(64, "multinomial"), (1000, "multinomial"), (4096, "systematic")])
def test_bot_model_ancestry_bit_exact(N, resampler):
    """The manoeuvring-target model of docs/experiments/BOT_Experiment_script.py:31-44 (sin / cos in the turn matrices,
    atan2 and a square root in the bearing-range emission) with inputs: on the canonical sin / cos / atan2 of
    csrc/bf_canon_math.hpp the whole ancestry, every weight and every particle equal the oracle's bits."""
    bfa, nl = _bfa()
    T = 24
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-3, 1e-2]).astype(F32)
    inputs = np.array([1] * 8 + [0] * 8 + [2] * 8, F32)
    po = go.ParamsBPF(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R,
                      go.GaussianEmissionLogProb(om.BearingRange(), R))
    g = nl.bearing_range()
    pp = bfa.ParamsBPF(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, g, np.zeros(2, F32), R, nl.gaussian_log_prob(g, R))
    xs, ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(3), T, inputs.reshape(T, 1))
    key = np.array([0, 5], np.uint32)
    ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=inputs.reshape(T, 1), resampler=resampler, debug=True,
                                            arith="canonical")
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, inputs, resampler=resampler, return_ancestors=True, output="both")
    assert dbg["resampled"].any()
    assert np.array_equal(out["resampled"].cpu().numpy() > 0.5, dbg["resampled"])
    assert np.array_equal(out["ancestors"].cpu().numpy().T, dbg["ancestors"])
    assert _bits_equal(out["weights"].cpu().numpy(), ref["weights"])
    assert _bits_equal(out["particles"].cpu().numpy(), ref["particles"])
    # the libm-arithmetic oracle agrees to rounding
    ref0 = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=inputs.reshape(T, 1), resampler=resampler)
    assert np.max(np.abs(out["weights"].cpu().numpy()[:, 0] - ref0["weights"][:, 0])) < 1e-6


def test_bpf_errors():
    bfa, nl = _bfa()
    from bayesianfiltering_amd import _lib
    p = _l63_params(bfa, nl)
    ys = np.zeros((4, 3), F32)
    with pytest.raises(_lib.BayesFiltError) as e:
        bfa.bootstrap_particle_filter(p, ys, (1 << 20) + 1)   # 2^20 particles per trajectory is the limit of the HBM path
    assert e.value.code == _lib.BF_EUNSUPPORTED
    with pytest.raises(TypeError):
        # (a Python log-density is recorded when it is written with numpy operations; one that leaves them cannot run on the device)
        bfa.bootstrap_particle_filter(p._replace(emission_distribution_log_prob=lambda x, y, u: float(x[0])), ys, 64)


def test_sixteen_thousand_particles_for_small_states():
    """N = 10 000 (the particle count of the reference's BOTExperiment notebook) on the manoeuvring-target
    model: 16 particles per thread; weights, particles and ancestry against the oracle."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T, N = 6, 10000
    mu0 = np.array([2.0, 0.3, 3.0, -0.2], F32)
    S0 = np.diag([0.1, 0.005, 0.1, 0.01]).astype(F32)
    Q, R = 1e-3 * np.eye(2, dtype=F32), np.diag([1e-2, 1e-1]).astype(F32)
    inputs = np.array([1, 1, 0, 0, 2, 2], F32)
    po = go.ParamsBPF(mu0, S0, om.ManeuverBOT(), np.zeros(2, F32), Q, om.BearingRange(), np.zeros(2, F32), R,
                      go.GaussianEmissionLogProb(om.BearingRange(), R))
    pp = bfa.ParamsBPF(mu0, S0, nl.maneuver_bot(), np.zeros(2, F32), Q, nl.bearing_range(), np.zeros(2, F32), R,
                       nl.gaussian_log_prob(nl.bearing_range(), R))
    ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(8), T, inputs.reshape(T, 1))[1]
    key = otf.PRNGKey(5)
    ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=inputs.reshape(T, 1), debug=True)
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, inputs, return_ancestors=True)
    assert tuple(out["weights"].shape) == (N, T) and tuple(out["particles"].shape) == (N, T, 4)
    anc, ranc = out["ancestors"].cpu().numpy(), np.asarray(dbg["ancestors"])
    if ranc.shape != anc.shape:
        ranc = ranc.T
    # a draw within an ulp of a CDF step may pick the neighbouring index; from that step on the two particle sets
    # differ in one member and the runs drift apart, so: (nearly) identical ancestry at the first step, and
    # bit-level agreement of everything up to the first such event
    assert (anc[:, 0] == ranc[:, 0]).mean() > 0.9995
    t_ok = 0
    while t_ok < T and np.array_equal(anc[:, t_ok], ranc[:, t_ok]):
        t_ok += 1
    same = anc[:, 0] == ranc[:, 0]
    # ... and a moved draw only crosses particles of (numerically) no weight: the CDF mass strictly between the two
    # answers is at rounding level, i.e. both indices are valid inverses of the same uniform under fp32 rounding
    cdf = np.cumsum(np.asarray(dbg["pre_weights"])[0].astype(np.float64))
    lo, hi = np.minimum(anc[~same, 0], ranc[~same, 0]), np.maximum(anc[~same, 0], ranc[~same, 0])
    if lo.size:
        assert np.max(cdf[hi - 1] - cdf[lo]) < 2e-6, np.max(cdf[hi - 1] - cdf[lo])
    assert cm.rel_err(out["particles"].cpu().numpy()[same, 0], ref["particles"][same, 0]) < 2e-5
    if t_ok >= 1:
        assert cm.rel_err(out["particles"].cpu().numpy()[:, :t_ok], ref["particles"][:, :t_ok]) < 2e-5
        assert np.max(np.abs(out["weights"].cpu().numpy()[:, :t_ok] - ref["weights"][:, :t_ok])) < 1e-6
    # against the oracle on the engine's own (canonical) arithmetic: every ancestor, weight and particle of all steps
    refc, dbgc = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=inputs.reshape(T, 1), debug=True, arith="canonical")
    assert np.array_equal(anc, np.asarray(dbgc["ancestors"]).T)
    assert _bits_equal(out["weights"].cpu().numpy(), refc["weights"]) and _bits_equal(out["particles"].cpu().numpy(), refc["particles"])


@pytest.mark.parametrize("N,resampler", [(20000, "multinomial"), (17000, "systematic"), (70000, "multinomial")])
def test_particles_in_hbm_path(N, resampler):
    """More particles than one workgroup's registers hold (the reference runs 5e4 / 5e5): the chunked kernel with the
    particles in HBM keeps the tree orders and the canonical arithmetic of the small kernel -- ancestry, weights and
    particles of every step equal to the oracle's bits."""
    import bayesianfiltering_amd as bfa
    nl = bfa.nonlinearities
    T = 3
    po = go.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), om.Lorenz63(), np.zeros(3, F32),
                      0.1 * np.eye(3, dtype=F32), om.Linear(np.eye(3, dtype=F32)), np.zeros(3, F32), 0.5 * np.eye(3, dtype=F32),
                      go.GaussianEmissionLogProb(om.Linear(np.eye(3, dtype=F32)), 0.5 * np.eye(3, dtype=F32)))
    h = nl.linear_emission(np.eye(3, dtype=F32))
    pp = bfa.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(), np.zeros(3, F32),
                       0.1 * np.eye(3, dtype=F32), h, np.zeros(3, F32), 0.5 * np.eye(3, dtype=F32),
                       nl.gaussian_log_prob(h, 0.5 * np.eye(3, dtype=F32)))
    ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(3), T)[1]
    key = otf.PRNGKey(11)
    ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, ess_threshold=1.1, resampler=resampler, debug=True, arith="canonical")
    out = bfa.bootstrap_particle_filter(pp, ys, N, key, None, 1.1, resampler=resampler, output="both", return_ancestors=True)
    anc, ranc = out["ancestors"].cpu().numpy(), np.asarray(dbg["ancestors"])
    if ranc.shape != anc.shape:
        ranc = ranc.T
    # CDF steps of ~5e-5 / N: only a defined arithmetic makes these reproducible -- every index of every step
    assert np.array_equal(anc, ranc)
    assert _bits_equal(out["weights"].cpu().numpy(), ref["weights"]) and _bits_equal(out["particles"].cpu().numpy(), ref["particles"])
    assert _bits_equal(out["ess"].cpu().numpy(), np.asarray(dbg["ess"]))
    assert bool(np.isfinite(out["mean"].cpu().numpy()).all())


@pytest.mark.parametrize("N,resampler,B", [(20000, "multinomial", 1), (17000, "systematic", 3), (70000, "multinomial", 2)])
def test_workgroup_per_chunk_equals_workgroup_per_trajectory(N, resampler, B):
    """The two particles-in-HBM kernels (bpf_big.hpp: a workgroup per trajectory; bpf_wide.hpp: a workgroup per
    1024-particle chunk, six launches per step) keep the same reduction / scan trees: identical weights, particles,
    ancestors, ESS and evidence over steps that do and do not resample, chunked runs through the carry included."""
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    nl = bfa.nonlinearities
    T = 6
    h = nl.linear_emission(np.eye(3, dtype=F32))
    pp = bfa.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(), np.zeros(3, F32),
                       0.1 * np.eye(3, dtype=F32), h, np.zeros(3, F32), 4.0 * np.eye(3, dtype=F32),
                       nl.gaussian_log_prob(h, 4.0 * np.eye(3, dtype=F32)))
    rng = np.random.default_rng(N)
    ys = (np.array([0.0, 1.0, 1.05], F32) + rng.normal(size=(B, T, 3))).astype(F32)
    if B == 1:
        ys = ys[0]
    lib = _lib.load()
    outs = []
    try:
        for mode in (1, 2):
            _lib.check(lib.bf_set_option(b"bpf_hbm_mode", mode))
            out = bfa.bootstrap_particle_filter(pp, ys, N, bfa.PRNGKey(5), None, 0.6, resampler=resampler, output="both",
                                                return_ancestors=True)
            outs.append({k: v.cpu().numpy() for k, v in out.items()})
        # the per-chunk kernel in two pieces through the carry
        first, carry = bfa.bootstrap_particle_filter(pp, ys[..., :4, :], N, bfa.PRNGKey(5), None, 0.6, resampler=resampler,
                                                     output="both", return_carry=True)
        second = bfa.bootstrap_particle_filter(pp, ys[..., 4:, :], N, None, None, 0.6, resampler=resampler, output="both",
                                               carry=carry)
    finally:
        lib.bf_set_option(b"bpf_hbm_mode", 0)
    a, b_ = outs
    res = a["resampled"].reshape(-1)
    assert 0 < res.sum() < res.size, res           # both branches exercised
    for k in ("weights", "particles", "ancestors", "ess", "logz", "resampled"):
        assert np.array_equal(a[k], b_[k]), k
    assert cm.rel_err(b_["mean"], a["mean"]) < 1e-5
    t_axis = a["weights"].ndim - 1
    joined = np.concatenate([first["weights"].cpu().numpy(), second["weights"].cpu().numpy()], axis=t_axis)
    assert np.array_equal(joined, b_["weights"])
    joined = np.concatenate([first["particles"].cpu().numpy(), second["particles"].cpu().numpy()], axis=t_axis if B == 1 else t_axis)
    assert np.array_equal(joined, b_["particles"])


@pytest.mark.parametrize("N,n", [(64, 3), (300, 2), (5000, 3)])
def test_stochastic_volatility_log_density(N, n):
    """The reference's adaptive experiment (docs/experiments/adaptive_experiment.py:47-57, :150-164): linear dynamics,
    emission u beta exp(x / sigma) r + (1 - u)(0.1 x + r) switched on half-way by the input, and the state-dependent
    log-density lmsvlp = MVN(glmsv(x, r0, u), M R M^T).  In-register kernel (N = 64, 300) and workgroup-per-chunk
    kernels (N = 5000): ancestors bit-exact, particles / weights to rounding."""
    bfa, nl = _bfa()
    T = 12
    Phi = (0.8 * np.eye(n)).astype(F32)
    Q = (2.0 * np.eye(n)).astype(F32)
    R = (1e-1 * np.eye(n) + 0.02).astype(F32)              # a full R: exercises the forward substitution
    r0 = np.zeros(n, F32)
    u = np.array([0] * (T // 2) + [1] * (T - T // 2), F32)
    hn = om.StochVol(n)
    po = go.ParamsBPF(np.zeros(n, F32), np.eye(n, dtype=F32), om.Linear(Phi), np.zeros(n, F32), Q, hn, r0, R,
                      go.StochVolEmissionLogProb(hn, R))
    h = nl.stoch_vol(n)
    pp = bfa.ParamsBPF(np.zeros(n, F32), np.eye(n, dtype=F32), nl.linear_dynamics(Phi), np.zeros(n, F32), Q, h, r0, R,
                       nl.stoch_vol_log_prob(h, R))
    ys = go.sample_ssm(go.ParamsNLSSM(*po[:8]), otf.PRNGKey(21), T, u.reshape(T, 1))[1]
    key = otf.PRNGKey(4)
    if N <= 300:
        ref, dbg = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=u.reshape(T, 1), debug=True, arith="canonical")
        out = bfa.bootstrap_particle_filter(pp, ys, N, key, u, output="both", return_ancestors=True)
        assert np.array_equal(out["resampled"].cpu().numpy() > 0.5, dbg["resampled"])
        assert dbg["resampled"].any()
        assert np.array_equal(out["ancestors"].cpu().numpy().T, dbg["ancestors"])
        assert _bits_equal(out["particles"].cpu().numpy(), ref["particles"])
        assert _bits_equal(out["weights"].cpu().numpy(), ref["weights"])
        assert _bits_equal(out["ess"].cpu().numpy(), dbg["ess"])
        ref0 = go.bootstrap_particle_filter(po, ys, N, key=key, inputs=u.reshape(T, 1))        # libm arithmetic: to rounding
        assert np.max(np.abs(out["weights"].cpu().numpy()[:, :3] - ref0["weights"][:, :3])) < 2e-6
    else:
        # the particles-in-HBM kernels against the in-register kernel (same trees; bpf_hbm_mode 1 keeps N = 5000 in one
        # workgroup's registers only when it fits -- n = 3 does, 16 particles per thread)
        from bayesianfiltering_amd import _lib
        lib = _lib.load()
        outs = []
        try:
            for mode in (1, 2):
                _lib.check(lib.bf_set_option(b"bpf_hbm_mode", mode))
                o = bfa.bootstrap_particle_filter(pp, ys, N, key, u, output="both", return_ancestors=True)
                outs.append({k: v.cpu().numpy() for k, v in o.items()})
        finally:
            lib.bf_set_option(b"bpf_hbm_mode", 0)
        a, b_ = outs
        assert (a["ancestors"] == b_["ancestors"]).mean() > 0.999
        assert cm.rel_err(b_["ess"], a["ess"]) < 1e-4
        assert bool(np.isfinite(b_["mean"]).all())


def test_constant_covariance_log_prob_rejects_stochastic_volatility():
    bfa, nl = _bfa()
    with pytest.raises(ValueError):
        nl.gaussian_log_prob(nl.stoch_vol(3), np.eye(3, dtype=F32))
    with pytest.raises(ValueError):
        nl.stoch_vol_log_prob(nl.linear_emission(np.eye(3, dtype=F32)), np.eye(3, dtype=F32))


@pytest.mark.parametrize("n,N", [(8, 100), (8, 1000), (16, 4096), (16, 700)])
def test_compile_time_model_instance_equals_the_run_time_one(n, N):
    """The Lorenz-96 / selection-emission / diagonal-covariance structure of BASELINE configs[3] runs on an instance whose
    model structure is a compile-time property (ssm_device.hpp: SpecFixed; bf_set_option "bpf_spec" = 1, the default) --
    noise added as each Threefry block's normals arrive, no per-particle switch.  Same operations in the same order: every
    weight, particle, ancestor and summary equal to the run-time instance's bits (non-zero noise bias, non-uniform diagonal
    covariances, noise-bias evaluation point of the log-density included), one-shot and through the carry."""
    import torch
    bfa, nl = _bfa()
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    rng = np.random.default_rng(n + N)
    m, T, B = n // 2, 9, 3
    Q = np.diag(rng.uniform(0.02, 0.2, size=n)).astype(F32)
    R = np.diag(rng.uniform(0.3, 0.8, size=m)).astype(F32)
    q0 = (0.05 * rng.normal(size=n)).astype(F32)
    r0 = (0.05 * rng.normal(size=m)).astype(F32)
    g = nl.pick_even(n)
    pp = bfa.ParamsBPF(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), q0, Q, g, r0, R, nl.gaussian_log_prob(g, R, r0))
    ys = cm.device_observations(bfa.ParamsNLSSM(*pp[:8]), (n, n, m, m), B, T, seed=n).cpu().numpy()
    key = np.array([3, 7], np.uint32)
    res = {}
    for spec in (1, 0):
        _lib.check(lib.bf_set_option(b"bpf_spec", spec))
        try:
            res[spec] = bfa.bootstrap_particle_filter(pp, ys, N, key, output="both", return_ancestors=True, return_carry=True)
            h1, c1 = bfa.bootstrap_particle_filter(pp, ys[:, :4], N, key, output="both", return_carry=True)
            h2, c2 = bfa.bootstrap_particle_filter(pp, ys[:, 4:], N, None, output="both", carry=c1, return_carry=True)
            for k in ("weights", "particles", "mean", "ess", "logz", "resampled"):
                tdim = 2 if k in ("weights", "particles") else 1
                assert torch.equal(torch.cat([h1[k], h2[k]], dim=tdim), res[spec][0][k]), (spec, k)     # chunked == one-shot
        finally:
            _lib.check(lib.bf_set_option(b"bpf_spec", 1))
    assert float(res[1][0]["resampled"].mean()) > 0
    for k in res[1][0]:
        assert torch.equal(res[1][0][k], res[0][0][k]), k
    assert torch.equal(res[1][1].particles, res[0][1].particles) and torch.equal(res[1][1].weights, res[0][1].weights)
    assert torch.equal(res[1][1].key, res[0][1].key)


@pytest.mark.parametrize("n,N", [(8, 256), (16, 4096), (3, 1000)])
def test_hardware_arithmetic_option_agrees_to_rounding(n, N):
    """bf_set_option "bpf_arith" = 1 (per call: options={"bpf_arith": 1}): the same kernel with v_log_f32 / v_exp_f32 where the
    default build has the engine's defined fp32 arithmetic -- for callers who want the speed and do not need resampling indices
    reproducible bit for bit.  The two builds draw the same Threefry bits and differ by ~1 ulp per transcendental: the weighted
    means agree to rounding step after step until a uniform draw lands within that rounding of a CDF step (from there the runs
    are two equally valid samples; with 4 096 draws per step that happens within the first few steps), the resampling decisions agree while they do, and the libm-arithmetic oracle -- the third
    rounding of the same recursion -- sits at the same distance from both."""
    import torch
    bfa, nl = _bfa()
    T, B = 12, 2
    if n == 3:
        m = 1
        pp = bfa.ParamsBPF(np.array([0.0, 1.0, 1.05], F32), np.eye(3, dtype=F32), nl.lorenz63(10.0, 28.0, 2.667, 0.01), np.zeros(3, F32),
                           0.1 * np.eye(3, dtype=F32), nl.quadratic(3, 0.05), np.zeros(1, F32), np.eye(1, dtype=F32),
                           nl.gaussian_log_prob(nl.quadratic(3, 0.05), np.eye(1, dtype=F32)))
        po = go.ParamsBPF(*pp[:2], om.Lorenz63(), pp[3], pp[4], om.Quadratic(3, 0.05), pp[6], pp[7],
                          go.GaussianEmissionLogProb(om.Quadratic(3, 0.05), np.eye(1, dtype=F32)))
    else:
        m = n // 2
        g = nl.pick_even(n)
        R = 0.5 * np.eye(m, dtype=F32)
        pp = bfa.ParamsBPF(8 * np.ones(n, F32), np.eye(n, dtype=F32), nl.lorenz96(n), np.zeros(n, F32), 1e-2 * np.eye(n, dtype=F32), g,
                           np.zeros(m, F32), R, nl.gaussian_log_prob(g, R))
        po = go.ParamsBPF(*pp[:2], om.Lorenz96(n), pp[3], pp[4], om.PickEven(n), pp[6], pp[7], go.GaussianEmissionLogProb(om.PickEven(n), R))
    ys = cm.device_observations(bfa.ParamsNLSSM(*pp[:8]), (n, n, m, m), B, T, seed=7 + n).cpu().numpy()
    key = np.array([1, 9], np.uint32)
    # without resampling nothing can branch: every step's weighted mean, ESS and evidence agree to rounding
    a0 = bfa.bootstrap_particle_filter(pp, ys, N, key, None, 0.0, output="summary")
    h0 = bfa.bootstrap_particle_filter(pp, ys, N, key, None, 0.0, output="summary", options={"bpf_arith": 1})
    assert not torch.equal(a0["mean"], h0["mean"])                                         # another arithmetic ...
    for k in ("mean", "ess", "logz"):
        fin = torch.isfinite(a0[k]) & torch.isfinite(h0[k])
        assert fin.float().mean() > 0.9
        assert cm.rel_err(h0[k][fin].cpu().numpy(), a0[k][fin].cpu().numpy()) < 2e-5, k     # ... the same filter
    # with resampling: the same decisions and means to rounding until the first ancestor differs
    a = bfa.bootstrap_particle_filter(pp, ys, N, key, output="summary")
    h = bfa.bootstrap_particle_filter(pp, ys, N, key, output="summary", options={"bpf_arith": 1})
    again = bfa.bootstrap_particle_filter(pp, ys, N, key, output="summary")
    assert torch.equal(torch.nan_to_num(a["mean"]), torch.nan_to_num(again["mean"]))     # the override lasted one call
    am, hm = a["mean"].cpu().numpy(), h["mean"].cpu().numpy()
    for b in range(B):
        d = np.abs(am[b] - hm[b]).max(axis=1) / np.abs(am[b]).max()
        t_ok = int(np.argmax(d > 1e-4)) if (d > 1e-4).any() else T
        assert t_ok >= (1 if N > 1024 else 4), (b, d)      # (4 096 draws per step: a draw within an ulp of a CDF step comes early)
        assert np.array_equal(a["resampled"][b, :t_ok].cpu().numpy(), h["resampled"][b, :t_ok].cpu().numpy())
        assert np.isfinite(hm[b]).all()
        if N <= 1024:
            ref = go.bootstrap_particle_filter(po, ys[b], N, key=key)
            rm = np.einsum("itd,it->td", ref["particles"], ref["weights"])
            dl = np.abs(rm - hm[b]).max(axis=1) / np.abs(rm).max()
            assert (dl[:4] < 1e-4).all(), (b, dl)
    assert float(h["resampled"].mean()) > 0
