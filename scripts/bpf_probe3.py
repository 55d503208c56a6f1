"""Ad-hoc probe (round 3): the cfg4 particle-filter launch (Lorenz-96 n = 16, m = 8, N = 4096, bench.py's model and data) under
the kernel's options.  PB / PT / PESS / PREP from the environment; prints one line per (option set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
import bench
F32 = np.float32
B = int(os.environ.get("PB", 1024)); T = int(os.environ.get("PT", 100)); N = int(os.environ.get("PN", 4096))
ess = float(os.environ.get("PESS", 0.5))
nl = bfa.nonlinearities
g = nl.pick_even(16); R = 0.5 * np.eye(8, dtype=F32)
p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32), 1e-2 * np.eye(16, dtype=F32),
                  g, np.zeros(8, F32), R, nl.gaussian_log_prob(g, R))
y = bench.simulate_on_device(bfa.ParamsNLSSM(*p[:8]), (16, 16, 8, 8), B, T, seed=4000)
lib = _lib.require_gpu()
sets = [s for s in os.environ.get("PSETS", "spec=1;spec=0").split(";") if s]
ref = None
for s in sets:
    opts = dict(kv.split("=") for kv in s.split(",") if kv)
    for k, v in opts.items():
        _lib.check(lib.bf_set_option(("bpf_" + k).encode(), int(v)))
    best = 1e30
    for rep in range(int(os.environ.get("PREP", 3))):
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        out = bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), None, ess, output="summary")
        e0.record(); torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(e0))
    same = "" if ref is None else f" same-as-first={bool(torch.equal(torch.nan_to_num(out['mean']), torch.nan_to_num(ref)))}"
    if ref is None:
        ref = out["mean"].clone()
    print(f"{s:24s} B={B} T={T} N={N} ess={ess}: {best:8.2f} ms  {B*T/best/1e3:7.3f} Mstep/s  resampled {out['resampled'].mean().item():.2f}{same}", flush=True)
    for k in opts:
        _lib.check(lib.bf_set_option(("bpf_" + k).encode(), 1 if k == "spec" else 0))
