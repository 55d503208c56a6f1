import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
nl = bfa.nonlinearities
F32 = np.float32
lib = _lib.require_gpu()
pp = bfa.ParamsNLSSM(np.zeros(8, F32), np.eye(8, dtype=F32), nl.lorenz96(8), np.zeros(8, F32),
                     1e-2 * np.eye(8, dtype=F32), nl.pick_even(8), np.zeros(4, F32), 1e-1 * np.eye(4, dtype=F32))
rng = np.random.default_rng(0)
for K in (8, 16, 32):
    for B in (3, 4, 8):
        T = 4
        ys = (8 + rng.normal(size=(B, T, 4))).astype(F32)
        init = (8 + rng.normal(size=(B, K, 8))).astype(F32)
        res = {}
        for st in (1, 0):
            lib.bf_set_option(b"gsf_structured", st)
            res[st] = bfa.gaussian_sum_filter(pp, ys, K, 1, initial_means=init)
        lib.bf_set_option(b"gsf_structured", 1)
        dw = (res[1].weights - res[0].weights).abs().max().item()
        dm = (res[1].means - res[0].means).abs().max().item()
        print(K, B, "dw", dw, "dm", dm, "sumw", res[1].weights.sum(1)[:, -1].cpu().numpy()[:3], res[0].weights.sum(1)[:, -1].cpu().numpy()[:3])
