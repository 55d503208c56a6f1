"""Ad-hoc probe: Gaussian-sum filter kernel time at cfg3 shape (Lorenz-96 n=8, m=4, K=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
F32 = np.float32
B = int(os.environ.get("PB", 16384)); T = int(os.environ.get("PT", 100)); K = int(os.environ.get("PK", 32))
nl = bfa.nonlinearities
p = bfa.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32), 1e-2 * np.eye(8, dtype=F32),
                    nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
dev = torch.device("cuda")
y = 8.0 + torch.randn((B, T, 4), device=dev)
init = 8.0 + torch.randn((B, K, 8), device=dev)
lib = _lib.require_gpu()
for lanes in [int(v) for v in os.environ.get("PLANES", "2,4,8").split(",")]:
  lib.bf_set_option(b"kf_lanes", lanes)
  print("lanes", lanes)
  for fields, name in ((bfa.FULL5, "FULL5"), ((), "none")):
    for mode in (-1,):
        lib.bf_set_option(b"kf_emit_mode", mode)
        post = bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=fields, return_carry=True)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            bfa.gaussian_sum_filter(p, y, K, 1, initial_means=init, fields=fields, out=post[0], return_carry=True)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        per = {"FULL5": 18576, "FILTERED": 9360, "none": 16}[name]
        print(f"{name:9s} mode={mode:2d} B={B} T={T} K={K}: {best:8.3f} ms  {B*T/best/1e3:8.2f} Mstep/s  {per*B*T/best/1e6:8.1f} GB/s  "
              f"{1.4e5*B*T/best/1e9:6.1f} TFLOP/s", flush=True)
