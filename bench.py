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
    args = ap.parse_args()

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
