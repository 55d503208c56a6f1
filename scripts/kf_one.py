"""Run ONE configuration of the batched Kalman filter a few times (for rocprofv3 / PMC runs).
env: PB batch, PT steps, PK kernel (0 cols, 1 lane-per-chain), PL layout, PM emit mode, PF fields preset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
import bench

B = int(os.environ.get("PB", 65536)); T = int(os.environ.get("PT", 2000))
K = int(os.environ.get("PK", 0)); L = os.environ.get("PL", "reference"); M = int(os.environ.get("PM", -1))
F = {"none": (), "full5": bfa.FULL5, "covs": ("covariances", "predicted_covariances")}[os.environ.get("PF", "full5")]
reps = int(os.environ.get("PR", 3))
a = bench.cv_model(); nl = bfa.nonlinearities
params = bfa.ParamsNLSSM(a["m0"], a["P0"], nl.linear_dynamics(a["A"], a["G"]), a["q0"], a["Q"],
                         nl.linear_emission(a["H"], a["D"]), a["r0"], a["R"])
dev = torch.device("cuda")
y = torch.randn((B, T, 2), device=dev); init = torch.zeros((B, 4), device=dev)
lib = _lib.require_gpu()
lib.bf_set_option(b"kf_emit_mode", M); lib.bf_set_option(b"kf_lanes", K)
post = bfa.kalman_filter(params, y, initial_means=init, layout=L, fields=F, return_carry=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    bfa.kalman_filter(params, y, initial_means=init, layout=L, fields=F, out=post[0], return_carry=True)
e.record(); torch.cuda.synchronize()
print(f"kernel={K} layout={L} mode={M} fields={os.environ.get('PF','full5')} B={B} T={T}: {s.elapsed_time(e)/reps:.3f} ms")
