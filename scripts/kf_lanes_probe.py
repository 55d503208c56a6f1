"""Ad-hoc: headline Kalman scan time per lanes-per-trajectory setting and layout (B=65536, T=2000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
from tests import common as cm
lib = _lib.require_gpu()
a = cm.cv_model_arrays(); p = cm.product_params(a)
B, T = int(os.environ.get("PB", 65536)), int(os.environ.get("PT", 2000))
y = torch.randn((B, T, 2), device="cuda"); init = torch.zeros((B, 4), device="cuda")
for layout in ("reference", "batch_inner"):
    for lanes in (0, 1, 2, 4):
        lib.bf_set_option(b"kf_lanes", lanes)
        post = bfa.kalman_filter(p, y, initial_means=init, layout=layout)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); bfa.kalman_filter(p, y, initial_means=init, layout=layout, out=post); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        print(f"{layout:12s} lanes={lanes}: {best:7.3f} ms  {172*B*T/best/1e6:7.1f} GB/s", flush=True)
        del post
lib.bf_set_option(b"kf_lanes", 0)
