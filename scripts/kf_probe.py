"""Ad-hoc probe: kernel time of the batched Kalman filter for subsets of the output streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
import bench

B = int(os.environ.get("PB", 65536)); T = int(os.environ.get("PT", 2000))
a = bench.cv_model(); nl = bfa.nonlinearities
params = bfa.ParamsNLSSM(a["m0"], a["P0"], nl.linear_dynamics(a["A"], a["G"]), a["q0"], a["Q"],
                         nl.linear_emission(a["H"], a["D"]), a["r0"], a["R"])
dev = torch.device("cuda")
y = torch.randn((B, T, 2), device=dev)
init = torch.zeros((B, 4), device=dev)
lib = _lib.require_gpu()

def run(fields, layout, mode, lanes=0, reps=3):
    lib.bf_set_option(b"kf_emit_mode", mode)
    lib.bf_set_option(b"kf_lanes", lanes)
    post = bfa.kalman_filter(params, y, initial_means=init, layout=layout, fields=fields, return_carry=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        bfa.kalman_filter(params, y, initial_means=init, layout=layout, fields=fields, out=post[0], return_carry=True)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    ms = best
    nb = {"weights": 4, "means": 16, "covariances": 64, "predicted_means": 16, "predicted_covariances": 64}
    by = 8 + sum(nb[f] for f in fields)
    print(f"lanes={lanes} {layout:12s} mode={mode:2d} fields={','.join(f[:6] for f in fields) or '-':40s} {ms:8.3f} ms  {B*T/ms/1e6:8.2f} Gstep/s  {by*B*T/ms/1e6:8.1f} GB/s", flush=True)

F5 = bfa.FULL5
for rnd in range(2):
    for lanes in (1, 2, 4):
        run((), "reference", 2, lanes)
        run(("covariances", "predicted_covariances"), "reference", 2, lanes)
        run(F5, "reference", 2, lanes)
        run(F5, "batch_inner", -1, lanes)
