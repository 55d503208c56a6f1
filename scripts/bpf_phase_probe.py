"""Ad-hoc: per-phase wall-clock ticks (10 ns) of every wave of workgroup 0 from the instrumented particle-filter build
(bayesianfiltering_amd/csrc/alt/build_timers.sh; run with BAYESFILT_HIP_LIB=.../alt/libbayesfilt_bpftimers.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
import bench
F32 = np.float32
B = int(os.environ.get("PB", 1024)); T = int(os.environ.get("PT", 100)); N = 4096
nl = bfa.nonlinearities
g = nl.pick_even(16); R = 0.5 * np.eye(8, dtype=F32)
p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32), 1e-2 * np.eye(16, dtype=F32),
                  g, np.zeros(8, F32), R, nl.gaussian_log_prob(g, R))
y = bench.simulate_on_device(bfa.ParamsNLSSM(*p[:8]), (16, 16, 8, 8), B, T, seed=4000)
lib = _lib.require_gpu()
names = ["propagate", "max-red", "exp+sum-red", "norm+ess-red", "cdf+search", "gather", "emit+mean", "loop-top"]
for spec in [int(v) for v in os.environ.get("PSPECS", "1,0").split(",")]:
    for ess in (0.5, 0.0):
        _lib.check(lib.bf_set_option(b"bpf_spec", spec))
        for rep in range(2):
            out, carry = bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), None, ess, output="summary", return_carry=True)
        torch.cuda.synchronize()
        tim = carry.weights[0, :128].cpu().numpy().reshape(16, 8) / T * 10.0 / 1000.0     # us per step
        print(f"== spec={spec} ess={ess}: us per step, per wave (rows: wave 0..15)")
        print("      " + " ".join(f"{n:>12s}" for n in names) + "        total")
        for w in range(16):
            print(f"w{w:02d}   " + " ".join(f"{v:12.2f}" for v in tim[w]) + f"   {tim[w].sum():10.2f}")
_lib.check(lib.bf_set_option(b"bpf_spec", 1))
