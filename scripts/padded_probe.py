"""Ad-hoc: Kalman scan throughput by state dimension, default routing (matrix-core kernel with the model zero-padded into
the (64, 32) tiles from n = 24 up) against the run-time-dimension kernel (force_generic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
from tests import common as cm
lib = _lib.require_gpu()
for n, m, B, T in ((12, 4, 16384, 100), (16, 8, 16384, 100), (20, 8, 16384, 100), (24, 12, 16384, 100), (32, 16, 16384, 100), (32, 32, 16384, 100), (48, 24, 8192, 50), (64, 16, 8192, 50), (64, 32, 8192, 50)):
    a = cm.random_stable_lgssm(n, m, seed=n)
    p = cm.product_params(a)
    y = torch.randn((B, T, m), device="cuda")
    init = torch.zeros((B, n), device="cuda")
    for force in (0, 1):
        lib.bf_set_option(b"force_generic", force)
        for fields, name in (((), "none"), (bfa.FULL5, "FULL5")):
            post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            bfa.kalman_filter(p, y, initial_means=init, fields=fields, out=post[0], return_carry=True)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e)
            print(f"{'generic' if force else 'default'} n={n:3d} m={m:3d} B={B} T={T} {name:5s}: {ms:8.2f} ms  {B*T/ms/1e3:9.3f} Mstep/s", flush=True)
lib.bf_set_option(b"force_generic", 0)
