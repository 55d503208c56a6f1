"""Ad-hoc: throughput of the run-time-dimension kernel (generic_scan.hip) on dense random-stable Kalman models."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
from tests import common as cm
lib = _lib.require_gpu()
for n, m, B, T, force in ((8, 4, 16384, 200, True), (12, 4, 16384, 200, False), (16, 8, 16384, 200, False), (24, 12, 8192, 100, False), (32, 16, 8192, 100, False),
                          (48, 24, 4096, 50, False), (64, 32, 4096, 50, True)):
    a = cm.random_stable_lgssm(n, m, seed=n)
    p = cm.product_params(a)
    y = torch.randn((B, T, m), device="cuda")
    init = torch.zeros((B, n), device="cuda")
    lib.bf_set_option(b"force_generic", 1 if force else 0)
    for fields, name in (((), "none"), (bfa.FULL5, "FULL5")):
        post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        bfa.kalman_filter(p, y, initial_means=init, fields=fields, out=post[0], return_carry=True)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        flop = 2 * (2 * n ** 3 + 2 * m * n * n + 2 * m * m * n) + 2 * m ** 3 / 3
        print(f"generic n={n:3d} m={m:3d} B={B} T={T} {name:5s}: {ms:8.2f} ms  {B*T/ms/1e3:9.3f} Mstep/s  {flop*B*T/ms/1e9:6.2f} TFLOP/s", flush=True)
lib.bf_set_option(b"force_generic", 0)
