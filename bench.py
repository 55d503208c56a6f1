#!/usr/bin/env python3
"""Benchmark of the filtering hot path: filter timesteps/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                     # headline: BASELINE configs[1]
    python bench.py --config {gsf32,bpf4096,kalman64} --gpus N ...    # configs[2..4]

One "step" = one pass of the hot path (the whole T-step scan) over one batch of synthetic trajectories already
resident in HBM.  Observations are drawn FROM THE MODEL ITSELF on the device (bf_sample_ssm_f32 =
NonlinearSSM.sample, gaussfiltax/models.py:240-289), so the filters track and stay finite; the JSON line reports the
fraction of trajectories whose final posterior is finite.

N > 1 (launched by torch.distributed.run, one rank per GPU; the process group is initialised before anything else
touches the GPU): trajectories are independent, the batch axis shards with no traffic during the scan, and the
per-trajectory posterior summaries are all-gathered over RCCL inside the timed region.
  * kalman4 (headline):           every rank filters its own 65 536 trajectories            -> "scaling": "weak"
  * gsf32 / bpf4096 / kalman64:   BASELINE's total batch (16 384 / 8 192 / 32 768) is split
                                  over the ranks with distributed.shard_bounds               -> "scaling": "strong"
Rank 0 prints ONE JSON line carrying `roofline` (dominant kernel: algorithmic bytes / flops / issue slots per launch over
its HIP-event time) and, for the headline at N = 1, `cpu_baseline` (the oracle's plain-C port on the host cores, a
bounded sample of the same workload -- a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL across processes)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
FP32_PEAK_TFS = 157.3      # fp32 vector = fp32 MFMA peak
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2   # wave64 VALU instructions/ns the chip can issue: 256 CUs x 4 SIMDs x 2.4 GHz, 2 cycles each
F32 = np.float32


def cv_model():
    """constant-velocity LGSSM n=4, m=2 (BOT_Experiment_script.py:31-32,40)."""
    dt = 0.5
    return dict(
        A=np.array([[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1]], F32),
        G=np.array([[0.5, 0], [1, 0], [0, 0.5], [0, 1]], F32),
        H=np.array([[1, 0, 0, 0], [0, 0, 1, 0]], F32), D=np.eye(2, dtype=F32),
        Q=1e-2 * np.eye(2, dtype=F32), R=1e-1 * np.eye(2, dtype=F32),
        m0=np.zeros(4, F32), P0=np.eye(4, dtype=F32), q0=np.zeros(2, F32), r0=np.zeros(2, F32))


def random_stable_lgssm(n, m, seed):
    """dense random-stable LGSSM (A = 0.95 x orthogonal, H ~ N(0, 1/n), Q = 1e-2 I, R = 1e-1 I)."""
    rng = np.random.default_rng(seed)
    Aq, _ = np.linalg.qr(rng.normal(size=(n, n)))
    return dict(A=(0.95 * Aq).astype(F32), G=np.eye(n, dtype=F32), H=(rng.normal(size=(m, n)) / np.sqrt(n)).astype(F32),
                D=np.eye(m, dtype=F32), Q=(1e-2 * np.eye(n)).astype(F32), R=(1e-1 * np.eye(m)).astype(F32),
                m0=np.zeros(n, F32), P0=np.eye(n, dtype=F32), q0=np.zeros(n, F32), r0=np.zeros(m, F32))


def simulate_on_device(params, dims, B, T, seed, first=0):
    """Synthetic observations (B, T, m): independent trajectories of the model itself from the engine's device data
    generator with keys split(PRNGKey(seed), first + B)[first:].  Setup, untimed."""
    import ctypes as C
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    lib = _lib.load()
    key = bfa.PRNGKey(seed)
    keys = np.empty((first + B, 2), dtype=np.uint32)
    _lib.check(lib.bf_random_split(key.ctypes.data_as(C.POINTER(C.c_uint32)), first + B, keys.ctypes.data_as(C.POINTER(C.c_uint32))))
    _, y = bfa.NonlinearSSM(*dims).sample(params, keys[first:], T)
    return y


def cpu_baseline(a, T, target_s=12.0):
    """Time the oracle's C port on a bounded sample of the same workload (rank 0, N=1 only)."""
    from oracle import c_oracle
    from tests import common as cm
    cores = c_oracle.max_threads()
    big = a["A"].shape[0] > 16
    probe_B = (2 if big else 64) * cores
    ys = cm.simulate_batch(a, probe_B, 100 if big else 1000, seed=5)
    init = np.tile(a["m0"], (probe_B, 1))
    fields = ("weights", "means", "covariances", "predicted_means", "predicted_covariances")
    t0 = time.perf_counter()
    c_oracle.kalman_filter(a, ys, init, fields=fields)
    rate = probe_B * ys.shape[1] / (time.perf_counter() - t0)
    Bs = int(max(cores, min(65536, rate * target_s / T)))
    ys = cm.simulate_batch(a, Bs, T, seed=6)
    init = np.tile(a["m0"], (Bs, 1))
    t0 = time.perf_counter()
    c_oracle.kalman_filter(a, ys, init, fields=fields)
    dt = time.perf_counter() - t0
    n, m = a["A"].shape[0], a["H"].shape[0]
    return {"value": Bs * T / dt, "unit": "timesteps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/c/kf_oracle.c (OpenMP over batch), B={Bs} of the same n={n} m={m} T={T} workload, "
                      f"all five streams, {dt:.1f} s"}


def cpu_baseline_gsf(y_host, init_host, K, mode, T, target_s=12.0):
    """configs[2] on the host cores: the C port of the EKF bank + weight update (oracle_gsf_lorenz96_f32), all FULL5-equivalent
    per-component streams it has (weights, means, covariances), on the first trajectories of the same observations."""
    from oracle import c_oracle
    cores = c_oracle.max_threads()
    n, m = 8, 4
    th = np.array([1.0, 1.0, 8.0, 0.01, 1.0 if mode == "matrix_power" else 0.0], F32)
    H = np.zeros((m, n), F32)
    H[np.arange(m), 2 * np.arange(m)] = 1
    args = (th, H, 1e-2 * np.eye(n, dtype=F32), 1e-1 * np.eye(m, dtype=F32), np.zeros(n, F32), np.zeros(m, F32))
    t0 = time.perf_counter()
    c_oracle.gsf_lorenz96(*args, y_host[:cores, :500], init_host[:cores], np.eye(n, dtype=F32))
    rate = cores * 500 / (time.perf_counter() - t0)
    Bs = int(max(cores, min(len(y_host), rate * target_s / T) // cores * cores))
    t0 = time.perf_counter()
    c_oracle.gsf_lorenz96(*args, y_host[:Bs], init_host[:Bs], np.eye(n, dtype=F32))
    dt = time.perf_counter() - t0
    return {"value": Bs * T / dt, "unit": "timesteps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/c/kf_oracle.c oracle_gsf_lorenz96_f32 (OpenMP over batch), the first B={Bs} trajectories of the same "
                      f"K={K} n={n} m={m} T={T} workload, weights + means + covariances written, {dt:.1f} s"}


def cpu_baseline_bpf(y_host, N, T, target_s=12.0):
    """configs[3] on the host cores: the C port of the bootstrap particle filter (oracle_bpf_lorenz96_f32: Threefry draws,
    libm exp, sequential sums), one trajectory per core, on a prefix of the same observations."""
    from oracle import c_oracle
    cores = c_oracle.max_threads()
    n, m = 16, 8
    th = np.array([1.0, 1.0, 8.0, 0.01, 1.0], F32)
    args = (th, np.zeros(n, F32), 1e-2 * np.ones(n, F32), 0.5 * np.ones(m, F32), 8 * np.ones(n, F32), np.ones(n, F32))
    key = np.array([0, 1], np.uint32)
    Bs = min(cores, len(y_host))
    t0 = time.perf_counter()
    c_oracle.bpf_lorenz96(*args, y_host[:Bs, :8], N, key)
    rate = Bs * 8 / (time.perf_counter() - t0)
    Ts = int(max(8, min(T, rate * target_s / Bs)))
    t0 = time.perf_counter()
    c_oracle.bpf_lorenz96(*args, y_host[:Bs, :Ts], N, key)
    dt = time.perf_counter() - t0
    return {"value": Bs * Ts / dt, "unit": "timesteps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/c/kf_oracle.c oracle_bpf_lorenz96_f32 (OpenMP over batch), the first {Ts} steps of the first B={Bs} "
                      f"trajectories of the same N={N} n={n} m={m} workload, weighted means written, {dt:.1f} s"}


def profiled(name):
    """A figure measured by rocprofv3 PMC passes in ANOTHER run and committed under profiles/ (never read as live):
    returns (value, source path) or (None, None)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        return None, None


# ------------------------------------------------------------------------------------------------ workloads
def make_kalman4(args, rank, world, device):
    """BASELINE configs[1]: n = 4, m = 2, T = 10 000, B = 65 536 per GPU, FULL5 in the reference layout."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_emit_mode", args.emit_mode))
    a = cv_model()
    nl = bfa.nonlinearities
    params = bfa.ParamsNLSSM(a["m0"], a["P0"], nl.linear_dynamics(a["A"], a["G"]), a["q0"], a["Q"],
                             nl.linear_emission(a["H"], a["D"]), a["r0"], a["R"])
    B = args.batch or 65536
    T, n, m = args.T or 10000, 4, 2
    y = simulate_on_device(params, (4, 2, 2, 2), B, T, seed=1000 + rank)
    init = torch.zeros((B, n), device=device)
    st = {"post": None}

    def kernels():
        st["post"] = bfa.kalman_filter(params, y, initial_means=init, layout=args.layout, out=st["post"])

    def summary():
        p = st["post"]
        return torch.cat([p.means[:, 0, -1], p.covariances[:, 0, -1].reshape(B, n * n)], dim=1)

    def finite():
        p = st["post"]
        return float((torch.isfinite(p.means[:, 0, -1]).all(dim=1) & torch.isfinite(p.covariances[:, 0, -1]).all(dim=(1, 2))).float().mean())

    bps = int(lib.bf_bytes_per_step(n, m, 1, None))      # 4m + 4(1 + 2n + 2n^2) = 172
    return dict(kernels=kernels, summary=summary, finite=finite, units=B * T, total_units=world * B * T, scaling="weak",
                gather_rows=world * B, a=a, T=T,
                roofline=lambda ms: {"bound": "hbm", "achieved": bps * B * T / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "kernel": "kf_scan_group_kernel<n=4,m=2,NL=2,EMIT_STAGED>", "bytes_per_step": bps},
                workload=f"batched Kalman filter state_dim=4 obs_dim=2 T={T} batch={B} per GPU, K=1, all five posterior streams "
                         f"(FULL5), layout={args.layout}, observations drawn from the model",
                extra={"batch_per_gpu": B, "T": T, "state_dim": n, "obs_dim": m})


def make_gsf32(args, rank, world, device):
    """BASELINE configs[2]: Lorenz-96 n = 8, m = 4, K = 32, T = 5 000, B = 16 384 (split over the ranks)."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import distributed as bdist
    nl = bfa.nonlinearities
    Bt, T, K, n, m, Tc = args.batch or 16384, args.T or 5000, 32, 8, 4, args.chunk
    lo, hi = bdist.shard_bounds(Bt, rank, world)
    B = hi - lo
    # --l96-mode matrix_power: the intended Lorenz-96 ((B x)_i = x_{i+1} - x_{i-2}); as_written: gaussfiltax/nonlinearities.py:48
    # literally (element-wise jnp.power => B == 0, a linear contraction).  On the chaotic model the reference's linear-domain
    # weights (inference.py:347-350) extinguish in finite time -- finite_frac reports how many trajectories still carry
    # finite weights at T; the arithmetic per step is the same either way
    p = bfa.ParamsNLSSM(8 * np.ones(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8, mode=args.l96_mode), np.zeros(8, F32),
                        1e-2 * np.eye(8, dtype=F32), nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
    y = simulate_on_device(p, (8, 8, 4, 4), B, T, seed=2000, first=lo)
    g = torch.Generator(device=device).manual_seed(20 + rank)
    init = 8.0 + torch.randn((B, K, n), device=device, generator=g)       # initial component means ~ N(m0, P0) (inference.py:367)
    st = {"post": None, "carry": None}
    collapsed = args.mode == "collapsed"

    def kernels():
        if collapsed:      # COLLAPSED mode: per step only the moment-matched Gaussian leaves the chip
            _, st["carry"], _ = bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=(), return_carry=True, return_collapsed=True)
            return
        carry = None
        for t0 in range(0, T, Tc):
            yc = y[:, t0:t0 + Tc]
            reuse = st["post"] if (st["post"] is not None and st["post"].weights.shape[2] == yc.shape[1]) else None
            st["post"], carry = bfa.gaussian_sum_filter(p, yc, K, 1, initial_means=init, carry=carry, out=reuse, return_carry=True)
        st["carry"] = carry

    def summary():
        c = st["carry"]
        return torch.cat([c.weights, c.means.reshape(B, K * n)], dim=1)

    def finite():
        c = st["carry"]
        return float((torch.isfinite(c.weights).all(dim=1) & torch.isfinite(c.means).all(dim=(1, 2))).float().mean())

    if collapsed:
        bps, fl = 4 * m + 4 * (n + n * n + K), 1.4e5
        roof = lambda ms: {"bound": "mfma", "achieved": fl * B * T / (ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFS, "unit": "TFLOP/s",
                           "kernel": "gsf_scan_kernel<8,4,NL=2,EMIT_NONE,EXT>", "flop_per_step": fl, "bytes_per_step": bps,
                           "note": "fp32 vector ALU work (no matrix products): 'mfma' stands for the fp32 vector peak"}
        work = f"Gaussian-sum filter 32 components, Lorenz-96[{args.l96_mode}] state_dim=8 obs_dim=4, T={T} batch={Bt}, COLLAPSED output (in-scan moment matching)"
    else:
        bps = 4 * m + 4 * K * (1 + 2 * n + 2 * n * n)
        roof = lambda ms: {"bound": "hbm", "achieved": bps * B * T / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "kernel": "gsf_scan_kernel<8,4,NL=2,EMIT_STAGED,L96_PICK>", "bytes_per_step": bps}
        work = f"Gaussian-sum filter 32 components, Lorenz-96[{args.l96_mode}] state_dim=8 obs_dim=4, T={T} batch={Bt}, FULL5 in T-chunks of {Tc}"
    return dict(kernels=kernels, summary=summary, finite=finite, units=B * T, total_units=Bt * T, scaling="strong", gather_rows=Bt,
                roofline=roof, workload=work + ", observations drawn from the model", extra={"batch_total": Bt, "batch_this_rank": B, "T": T},
                cpu=lambda: cpu_baseline_gsf(y[:4096].cpu().numpy(), init[:4096].cpu().numpy(), K, args.l96_mode, T))


def make_kalman64(args, rank, world, device):
    """BASELINE configs[4]: n = 64, m = 32, T = 2 000, B = 32 768 split over the ranks (matrix-core path, bf16 three-term products)."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import distributed as bdist
    nl = bfa.nonlinearities
    Bt, T, n, m, Tc = args.batch or 32768, args.T or 2000, 64, 32, 100
    lo, hi = bdist.shard_bounds(Bt, rank, world)
    B = hi - lo
    a = random_stable_lgssm(64, 32, seed=64)
    p = bfa.ParamsNLSSM(a["m0"], a["P0"], nl.linear_dynamics(a["A"], a["G"]), a["q0"], a["Q"], nl.linear_emission(a["H"], a["D"]),
                        a["r0"], a["R"])
    y = simulate_on_device(p, (64, 64, 32, 32), B, T, seed=5000, first=lo)
    init = torch.zeros((B, n), device=device)
    st = {"post": None, "carry": None}

    def kernels():
        carry = None
        for t0 in range(0, T, Tc):
            st["post"], carry = bfa.kalman_filter(p, y[:, t0:t0 + Tc], initial_means=init, carry=carry, out=st["post"], return_carry=True)
        st["carry"] = carry

    def summary():
        c = st["carry"]
        return torch.cat([c.means.reshape(B, n), c.covariances.reshape(B, n * n)], dim=1)

    def finite():
        c = st["carry"]
        return float((torch.isfinite(c.means.reshape(B, n)).all(dim=1) & torch.isfinite(c.covariances.reshape(B, n * n)).all(dim=1)).float().mean())

    # algorithmic flop per step of the reference's formulation (2 n^3 x 2 predict, H P, S, the gain solve,
    # K S K^T ~ 2.0e6).  The kernel runs the five products as three-term bf16 splits (6 x v_mfma_f32_32x32x16_bf16 per
    # K = 16 chunk, fp32 accumulation, fp32-level rounding) plus a ~2 900-instruction fp32 vector Cholesky; "peak" stays the
    # fp32 matrix / vector peak the same algebra would be priced against in fp32
    bps, fl = 4 * m + 4 * (1 + 2 * n + 2 * n * n), 2.0e6
    return dict(kernels=kernels, summary=summary, finite=finite, units=B * T, total_units=Bt * T, scaling="strong", gather_rows=Bt, a=a, T=T,
                roofline=lambda ms: {"bound": "mfma", "achieved": fl * B * T / (ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFS, "unit": "TFLOP/s",
                                     "kernel": "kf_scan_mfma5_kernel<64,32>", "flop_per_step": fl, "bytes_per_step": bps,
                                     "hbm_GBs": bps * B * T / (ms * 1e-3) / 1e9,
                                     "note": "fp32 MFMA and fp32 vector instructions share ONE datapath per SIMD on gfx950 "
                                             "(profiles/r02_f32_pipe_probe.txt), so the products run on the bf16 matrix pipe as exact "
                                             "three-term splits of the fp32 operands: 336 bf16 MFMAs (5.5 us of SIMD-time) + ~8 us of fp32 "
                                             "vector work per step over 4 SIMDs; parity with the fp32 oracle 4e-6 over 2 000 steps"},
                workload=f"Kalman filter state_dim=64 obs_dim=32 T={T} batch={Bt}, FULL5 in T-chunks of {Tc}, MFMA path (bf16 three-term splits, fp32 rounding), "
                         "observations drawn from the model",
                extra={"batch_total": Bt, "batch_this_rank": B, "T": T})


def make_bpf4096(args, rank, world, device):
    """BASELINE configs[3]: bootstrap PF, N = 4 096 particles, Lorenz-96 n = 16, m = 8, T = 2 000, B = 8 192 split over the ranks."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import distributed as bdist
    nl = bfa.nonlinearities
    Bt, T, N, n, m = args.batch or 8192, args.T or 2000, 4096, 16, 8
    lo, hi = bdist.shard_bounds(Bt, rank, world)
    B = hi - lo
    g = nl.pick_even(16)
    R = 0.5 * np.eye(8, dtype=F32)
    p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32),
                      1e-2 * np.eye(16, dtype=F32), g, np.zeros(8, F32), R, nl.gaussian_log_prob(g, R))
    # (the explicit-Euler Lorenz-96 of the reference driven by noise leaves the finite numbers on ~10 % of 2 000-step
    # trajectories -- in the DATA, before any filter: finite_frac reports it)
    y = simulate_on_device(bfa.ParamsNLSSM(*p[:8]), (16, 16, 8, 8), B, T, seed=4000, first=lo)
    st = {"out": None}

    def kernels():
        st["out"] = bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), output="summary")

    def summary():
        return st["out"]["mean"].reshape(B, T * n)           # the (B, T, n) point estimates of SURVEY.md 2 (1.05 GB in total)

    def finite():
        return float(torch.isfinite(st["out"]["mean"]).all(dim=(1, 2)).float().mean())

    prof, src = profiled("r03_pmc_bpf4096.json")
    ipp = (prof or {}).get("valu_wave_inst_per_particle_step_x64")   # SQ_INSTS_VALU per launch / particle-steps per launch x 64 lanes
    bps = 4 * m + 4 * (n + 3)

    def roof(ms):
        r = {"bound": "valu", "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
             "kernel": "bpf_scan_kernel<16,16,8,PPT=4,NW=16>", "bytes_per_step": bps, "particle_steps_per_s": N * B * T / (ms * 1e-3),
             "note": "integer / fp32 VALU work (Threefry, canonical erf_inv / exp), no matrix products and ~100 B of HBM per step: the "
                     "bound is VALU issue, 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (the guide's figure).  Measured on "
                     "this part (profiles/r03_valu_issue_probe.txt, four waves per SIMD): 2.5 cycles for v_add_u32 / v_xor_b32, 2.7-2.9 for "
                     "v_fma_f32 / v_fmaak_f32, 4.3 for v_alignbit_b32 / v_add3_u32 (a Threefry round = 9.3 cycles, not 6): the kernel's "
                     "instruction mix averages ~2.9 cycles per instruction -- frac_issue_cost_weighted prices the same count at that"}
        if ipp:
            r["achieved"] = ipp / 64.0 * N * B * T / (ms * 1e-3) / 1e9
            r["valu_inst_per_particle_step"] = ipp
            r["valu_inst_source"] = src
            r["frac_issue_cost_weighted"] = r["achieved"] / (256 * 4 * 2.4 / 2.9)
        else:
            r["achieved"] = None
        return r

    return dict(kernels=kernels, summary=summary, finite=finite, units=B * T, total_units=Bt * T, scaling="strong", gather_rows=Bt,
                roofline=roof,
                workload=f"bootstrap particle filter {N} particles, Lorenz-96 state_dim=16 obs_dim=8, T={T} batch={Bt}, SUMMARY output, "
                         "observations drawn from the model",
                extra={"batch_total": Bt, "batch_this_rank": B, "T": T, "particles": N},
                cpu=lambda: cpu_baseline_bpf(y[:256].cpu().numpy(), N, T))


MAKERS = {"kalman4": make_kalman4, "gsf32": make_gsf32, "bpf4096": make_bpf4096, "kalman64": make_kalman64}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="override the config's batch (per GPU for kalman4, total otherwise)")
    ap.add_argument("--T", type=int, default=0)
    ap.add_argument("--layout", default="reference", choices=["reference", "batch_inner"])
    ap.add_argument("--emit-mode", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=500,
                    help="gsf32 only: steps per launch (the posterior history of one chunk must fit HBM: 304 MB per step)")
    ap.add_argument("--mode", default="full5", choices=["full5", "collapsed"],
                    help="gsf32 only: full5 = the five posterior streams (T-chunked), collapsed = in-scan moment matching")
    ap.add_argument("--l96-mode", default="matrix_power", choices=["matrix_power", "as_written"], help="gsf32 only (see make_gsf32)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 code path -- init_process_group('nccl'), the RCCL all-gather of the summaries inside the timed "
                         "region, barriers, max-over-ranks -- even with ONE rank (also the default under torchrun --nproc-per-node 1)")
    ap.add_argument("--config", default="kalman4", choices=sorted(MAKERS),
                    help="kalman4 = BASELINE configs[1] (headline, default); gsf32 / bpf4096 / kalman64 = configs[2..4]")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="bf_set_option(NAME, VALUE) before the run (tuning experiments; repeatable), e.g. --option bpf_arith=1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and args.gpus > 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")

    # the collective path: always for N > 1; for ONE rank when launched by torchrun (RANK / WORLD_SIZE in the environment) or
    # with --force-collective, so that the exact N > 1 code executes on a single MI355X as well
    group = world > 1 or args.force_collective or ("RANK" in os.environ and "WORLD_SIZE" in os.environ)
    import torch
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if group:                          # the process group first: nothing else has touched the GPU yet
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from bayesianfiltering_amd import distributed as bdist
    if args.option:
        from bayesianfiltering_amd import _lib
        for kv in args.option:
            name, _, value = kv.partition("=")
            _lib.check(_lib.require_gpu().bf_set_option(name.encode(), int(value)))
    w = MAKERS[args.config](args, rank, world, device)

    gathered = {"t": None}

    def step():
        # one pass of the hot path over this rank's trajectories + the path's one exchange step:
        # RCCL all-gather of the per-trajectory posterior summaries
        w["kernels"]()
        if group:
            gathered["t"] = bdist.all_gather_summaries(w["summary"](), w["gather_rows"], force=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if group:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()               # the engine launches on torch's current stream: these events bracket its kernels
        w["kernels"]()
        ev[i][1].record()
        if group:
            gathered["t"] = bdist.all_gather_summaries(w["summary"](), w["gather_rows"], force=True)
    torch.cuda.synchronize()
    if group:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if group:
        assert tuple(gathered["t"].shape)[0] == w["gather_rows"], (tuple(gathered["t"].shape), w["gather_rows"])
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    fin = torch.tensor([w["finite"]() * w["units"]], device=device, dtype=torch.float64)
    if group:
        dist.all_reduce(fin)
    finite_frac = float(fin.item()) / w["total_units"]

    if rank == 0:
        roof = w["roofline"](kernel_ms)
        if roof.get("achieved") is not None:
            roof["frac"] = roof["achieved"] / roof["peak"]
        roof["kernel_ms"] = kernel_ms
        roof["traffic"] = None         # HBM bytes from PMC counters: not collected inside this run (see profiles/ for the rocprofv3 passes)
        if args.config == "kalman4" and w["extra"]["batch_per_gpu"] == 65536 and w["T"] == 10000:
            prof, src = profiled("r03_bench_%s_summary.json" % args.layout)
            if prof is None:
                prof, src = profiled("r02_bench_%s_summary.json" % args.layout)
            if prof and "pmc" in prof:
                try:
                    roof["traffic_profiled"] = {"bytes": (prof["pmc"]["WRITE_SIZE"]["mean_per_dispatch"] + prof["pmc"]["FETCH_SIZE"]["mean_per_dispatch"]) * 1024.0,
                                                "source": src}
                except KeyError:
                    pass
        if args.config == "gsf32" and args.mode == "full5" and w["extra"]["batch_total"] == 16384 and args.chunk == 500 and world == 1:
            prof, src = profiled("r03_pmc_gsf32_traffic.json")     # HBM bytes per 500-step chunk from separate rocprofv3 --pmc passes
            if prof and "pmc" in prof:
                try:
                    roof["traffic_profiled"] = {"bytes_per_chunk_launch": (prof["pmc"]["WRITE_SIZE"]["mean_per_dispatch"] +
                                                                           prof["pmc"]["FETCH_SIZE"]["mean_per_dispatch"]) * 1024.0,
                                                "algorithmic_bytes_per_chunk_launch": prof.get("algorithmic_bytes_per_dispatch"), "source": src}
                except KeyError:
                    pass
        line = {
            "metric": "filter timesteps/sec (batch x T)", "value": w["total_units"] * args.steps / elapsed, "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": w["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["workload"], **w["extra"],
                       "parallelism": f"batch-sharded x{world}" + (" + RCCL all-gather of summaries" if group else "")},
            "finite_frac": finite_frac,
            "roofline": roof,
            **({"options": args.option} if args.option else {}),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = w["cpu"]() if "cpu" in w else cpu_baseline(w["a"], w["T"])
        print(json.dumps(line), flush=True)
    if group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
