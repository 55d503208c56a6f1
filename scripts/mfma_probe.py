"""Ad-hoc probe: MFMA Kalman kernel (n=64, m=32) at cfg5's per-GPU shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from tests import common as cm
F32 = np.float32
B = int(os.environ.get("PB", 4096)); T = int(os.environ.get("PT", 50))
a = cm.random_stable_lgssm(64, 32, seed=64)
a["Q"] = (1e-2 * np.eye(64)).astype(F32); a["R"] = (1e-1 * np.eye(32)).astype(F32)
p = cm.product_params(a)
y = torch.randn((B, T, 32), device="cuda")
init = torch.zeros((B, 64), device="cuda")
for fields, name, per in ((bfa.FULL5, "FULL5", 33412), (bfa.FILTERED, "FILTERED", 16772), ((), "none", 128)):
    post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        bfa.kalman_filter(p, y, initial_means=init, fields=fields, out=post[0], return_carry=True)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    print(f"{name:9s} B={B} T={T}: {best:8.2f} ms  {B*T/best/1e3:8.2f} Mstep/s  {per*B*T/best/1e6:8.1f} GB/s  {2.0e6*B*T/best/1e9:6.1f} TFLOP/s", flush=True)
