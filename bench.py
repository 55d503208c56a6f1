#!/usr/bin/env python3
"""Headline benchmark: batched Kalman filter timesteps/s (BASELINE.json metric, configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (bf_kalman_filter_f32: the whole T-step scan) over one
batch of B synthetic trajectories already resident in HBM.  N > 1 (launched by
torch.distributed.run, one rank per GPU): every rank filters its own B trajectories (weak
scaling: trajectories are independent, the batch axis shards with no traffic during the scan)
and the per-trajectory posterior summaries (final mean, covariance, total log-likelihood) are
all-gathered over RCCL inside the timed region.  Rank 0 prints ONE JSON line.

The line carries `roofline` (dominant kernel: algorithmic HBM bytes / HIP-event time vs the
8 TB/s HBM3E peak) and `cpu_baseline` (the oracle's plain-C port on the host cores, a bounded
sample of the same workload -- a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def cv_model():
    """constant-velocity LGSSM n=4, m=2 (BOT_Experiment_script.py:31-32,40)."""
    dt = 0.5
    f32 = np.float32
    return dict(
        A=np.array([[1, dt, 0, 0], [0, 1, 0, 0], [0, 0, 1, dt], [0, 0, 0, 1]], f32),
        G=np.array([[0.5, 0], [1, 0], [0, 0.5], [0, 1]], f32),
        H=np.array([[1, 0, 0, 0], [0, 0, 1, 0]], f32), D=np.eye(2, dtype=f32),
        Q=1e-2 * np.eye(2, dtype=f32), R=1e-1 * np.eye(2, dtype=f32),
        m0=np.zeros(4, f32), P0=np.eye(4, dtype=f32), q0=np.zeros(2, f32), r0=np.zeros(2, f32))


def simulate_on_device(a, B, T, seed, device):
    """Synthetic observations (B, T, m) generated on the GPU with torch (setup, untimed)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    A, G, H = (torch.as_tensor(a[k], device=device) for k in ("A", "G", "H"))
    sq, sr = float(np.sqrt(a["Q"][0, 0])), float(np.sqrt(a["R"][0, 0]))
    x = torch.randn((B, 4), generator=g, device=device)
    y = torch.empty((B, T, 2), device=device)
    for t in range(T):
        x = x @ A.T + (sq * torch.randn((B, 2), generator=g, device=device)) @ G.T
        y[:, t] = x @ H.T + sr * torch.randn((B, 2), generator=g, device=device)
    return y


def cpu_baseline(a, T, target_s=12.0):
    """Time the oracle's C port on a bounded sample of the same workload (rank 0, N=1 only)."""
    from oracle import c_oracle
    from tests import common as cm
    cores = c_oracle.max_threads()
    probe_B = 64 * cores
    ys = cm.simulate_batch(a, probe_B, 1000, seed=5)
    init = np.tile(a["m0"], (probe_B, 1))
    fields = ("weights", "means", "covariances", "predicted_means", "predicted_covariances")
    t0 = time.perf_counter()
    c_oracle.kalman_filter(a, ys, init, fields=fields)
    rate = probe_B * 1000 / (time.perf_counter() - t0)
    Bs = int(max(cores, min(8192, rate * target_s / T)))
    ys = cm.simulate_batch(a, Bs, T, seed=6)
    init = np.tile(a["m0"], (Bs, 1))
    t0 = time.perf_counter()
    c_oracle.kalman_filter(a, ys, init, fields=fields)
    dt = time.perf_counter() - t0
    return {"value": Bs * T / dt, "unit": "timesteps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/c/kf_oracle.c (OpenMP over batch), B={Bs} of the same n=4 m=2 T={T} workload, "
                      f"all five streams, {dt:.1f} s"}


def pmc_traffic(B, T, layout):
    """HBM bytes per launch of the scan kernel from the committed rocprofv3 PMC passes
    (profiles/*_summary.json: WRITE_SIZE and FETCH_SIZE collected in separate runs of the same
    launch, in KiB).  WRITE_SIZE is exact for 16-byte-per-lane stores; FETCH_SIZE is taken as
    reported (the reads here are 4-byte LDS-DMA loads, for which the gfx950 half-count of wide
    loads is not calibrated -- reads are 5 % of the traffic).  None if no matching profile."""
    path = os.path.join(ROOT, "profiles", "r01_bench_%s_summary.json" % layout)
    if not (os.path.exists(path) and B == 65536 and T == 10000):
        return None
    try:
        pmc = json.load(open(path)).get("pmc", {})
        return (pmc["WRITE_SIZE"]["mean_per_dispatch"] + pmc["FETCH_SIZE"]["mean_per_dispatch"]) * 1024.0
    except (KeyError, ValueError):
        return None


def other_configs(args):
    """BASELINE configs[2..4] at their per-GPU shapes.  Same timing contract as the headline (W
    warm-up steps, K timed steps bracketed by synchronize); one step = the whole T-step scan,
    processed in T-chunks through the scan carry where the full posterior history exceeds HBM
    (output buffers are reused between chunks, the bytes are still written)."""
    import torch
    import bayesianfiltering_amd as bfa
    F32 = np.float32
    nl = bfa.nonlinearities
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if args.config == "gsf32":          # configs[2]: Lorenz-96 n=8, m=4, K=32, T=5000, B=16384
        B, T, K, n, m, Tc = 16384, 5000, 32, 8, 4, 100
        p = bfa.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32),
                            1e-2 * np.eye(8, dtype=F32), nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
        y = 8.0 + torch.randn((B, T, m), device=dev)
        init = 8.0 + torch.randn((B, K, n), device=dev)
        state = {"post": None}

        def step():
            carry = None
            for t0 in range(0, T, Tc):
                state["post"], carry = bfa.gaussian_sum_filter(p, y[:, t0:t0 + Tc], K, 1, initial_means=init, carry=carry,
                                                               out=state["post"], return_carry=True)
        bytes_per_step, flop_per_step = 4 * m + 4 * K * (1 + 2 * n + 2 * n * n), 1.4e5
        work = f"Gaussian-sum filter 32 components, Lorenz-96 state_dim=8 obs_dim=4, T={T} batch={B}, FULL5 in T-chunks of {Tc}"
        kernel = "gsf_scan_kernel<8,4,NL=4>"
    elif args.config == "kalman64":     # configs[4] per GPU: n=64, m=32, T=2000, B=32768/8
        from tests import common as cm
        B, T, n, m, Tc = 4096, 2000, 64, 32, 100
        a = cm.random_stable_lgssm(64, 32, seed=64)
        a["Q"] = (1e-2 * np.eye(64)).astype(F32); a["R"] = (1e-1 * np.eye(32)).astype(F32)
        p = cm.product_params(a)
        y = torch.randn((B, T, m), device=dev)
        init = torch.zeros((B, n), device=dev)
        state = {"post": None}

        def step():
            carry = None
            for t0 in range(0, T, Tc):
                state["post"], carry = bfa.kalman_filter(p, y[:, t0:t0 + Tc], initial_means=init, carry=carry,
                                                         out=state["post"], return_carry=True)
        bytes_per_step, flop_per_step = 4 * m + 4 * (1 + 2 * n + 2 * n * n), 2.0e6
        work = f"Kalman filter state_dim=64 obs_dim=32 T={T} batch={B} (one GPU's share of 32768), FULL5 in T-chunks of {Tc}, fp32 MFMA path"
        kernel = "kf_scan_mfma_kernel<64,32>"
    else:                               # configs[3] per GPU: bootstrap PF N=4096, n=16, T=2000, B=8192/8
        B, T, N, n, m = 1024, 2000, 4096, 16, 8
        g = nl.pick_even(16); R = 0.5 * np.eye(8, dtype=F32)
        p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32),
                          1e-1 * np.eye(16, dtype=F32), g, np.zeros(8, F32), R, nl.gaussian_log_prob(g, R))
        y = 8.0 + torch.randn((B, T, m), device=dev)

        def step():
            bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), output="summary")
        bytes_per_step, flop_per_step = 4 * m + 4 * (n + 3), 2.0e3 * N
        work = f"bootstrap particle filter {N} particles, Lorenz-96 state_dim=16 obs_dim=8, T={T} batch={B} (one GPU's share of 8192), SUMMARY output"
        kernel = "bpf_scan_kernel<16,16,8,PPT=4,NW=16>"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rate = B * T * args.steps / el
    gbs, tfs = bytes_per_step * rate / 1e9, flop_per_step * rate / 1e12
    bound = "hbm" if gbs / HBM_PEAK_GBS > tfs / 157.3 else "mfma"
    roof = ({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
            if bound == "hbm" else
            {"bound": "mfma", "achieved": tfs, "peak": 157.3, "unit": "TFLOP/s", "frac": tfs / 157.3})
    roof.update({"traffic": None, "kernel": kernel, "bytes_per_step": bytes_per_step, "flop_per_step": flop_per_step,
                 "hbm_GBs": gbs, "fp32_TFLOPs": tfs})
    print(json.dumps({"metric": "filter timesteps/sec (batch x T)", "value": rate, "unit": "timesteps/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": work}, "roofline": roof}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--T", type=int, default=10000)
    ap.add_argument("--layout", default="reference", choices=["reference", "batch_inner"])
    ap.add_argument("--emit-mode", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="kalman4", choices=["kalman4", "gsf32", "bpf4096", "kalman64"],
                    help="kalman4 = BASELINE configs[1] (headline, default); gsf32 / bpf4096 / kalman64 = configs[2..4] "
                         "at their per-GPU shapes (extra lines, not the headline)")
    args = ap.parse_args()
    if args.config != "kalman4":
        return other_configs(args)

    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    lib = _lib.require_gpu()
    _lib.check(lib.bf_set_option(b"kf_emit_mode", args.emit_mode))

    a = cv_model()
    nl = bfa.nonlinearities
    params = bfa.ParamsNLSSM(a["m0"], a["P0"], nl.linear_dynamics(a["A"], a["G"]), a["q0"], a["Q"],
                             nl.linear_emission(a["H"], a["D"]), a["r0"], a["R"])
    B, T, n, m = args.batch, args.T, 4, 2
    y = simulate_on_device(a, B, T, seed=1000 + rank, device=device)
    init = torch.zeros((B, n), device=device)

    post = None
    summary_local = torch.zeros((B, n + n * n), device=device)
    summary_all = None

    from bayesianfiltering_amd import distributed as bdist

    def step():
        # one pass of the hot path over this rank's batch + the path's one exchange step:
        # RCCL all-gather of the per-trajectory posterior summaries (final mean, covariance)
        nonlocal post, summary_all
        post = bfa.kalman_filter(params, y, initial_means=init, layout=args.layout, out=post)
        if world > 1:
            summary_local[:, :n] = post.means[:, 0, -1]
            summary_local[:, n:n + n * n] = post.covariances[:, 0, -1].reshape(B, n * n)
            summary_all = bdist.all_gather_summaries(summary_local, world * B)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        post = bfa.kalman_filter(params, y, initial_means=init, layout=args.layout, out=post)
        ev[i][1].record()
        if world > 1:
            summary_local[:, :n] = post.means[:, 0, -1]
            summary_local[:, n:n + n * n] = post.covariances[:, 0, -1].reshape(B, n * n)
            summary_all = bdist.all_gather_summaries(summary_local, world * B)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    if rank == 0:
        bytes_per_step = int(lib.bf_bytes_per_step(n, m, 1, None))      # 4m + 4(1 + 2n + 2n^2) = 172
        achieved = bytes_per_step * B * T / (kernel_ms * 1e-3) / 1e9
        traffic = pmc_traffic(B, T, args.layout)
        value = world * B * T * args.steps / elapsed
        line = {
            "metric": "filter timesteps/sec (batch x T)", "value": value, "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batched Kalman filter state_dim=4 obs_dim=2 T={T} batch={B} per GPU, K=1, "
                                   f"all five posterior streams (FULL5), layout={args.layout}",
                       "batch_per_gpu": B, "T": T, "state_dim": n, "obs_dim": m,
                       "parallelism": f"batch-sharded x{world}" + (" + RCCL all-gather of summaries" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "kf_scan_group_kernel<n=4,m=2>", "kernel_ms": kernel_ms,
                         "bytes_per_step": bytes_per_step},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
