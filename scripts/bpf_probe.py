"""Ad-hoc probe: bootstrap particle filter kernel time at cfg4 shape (Lorenz-96 n=16, m=8, N=4096)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
F32 = np.float32
B = int(os.environ.get("PB", 1024)); T = int(os.environ.get("PT", 50)); N = int(os.environ.get("PN", 4096))
nl = bfa.nonlinearities
g = nl.pick_even(16); R = 0.5 * np.eye(8, dtype=F32)
p = bfa.ParamsBPF(8 * np.ones(16, F32), np.eye(16, dtype=F32), nl.lorenz96(16), np.zeros(16, F32), 1e-1 * np.eye(16, dtype=F32),
                  g, np.zeros(8, F32), R, nl.gaussian_log_prob(g, R))
y = 8.0 + torch.randn((B, T, 8), device="cuda")
from bayesianfiltering_amd import _lib
for rep in range(int(os.environ.get("PREP", 4))):
    _lib.require_gpu().bf_set_option(b"bpf_variant", rep % 2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = bfa.bootstrap_particle_filter(p, y, N, np.array([0, 1], np.uint32), None, float(os.environ.get("PESS", 0.5)), resampler=os.environ.get("PRES", "multinomial"), output="summary")
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    print(f"variant={rep % 2} B={B} T={T} N={N}: {ms:8.2f} ms  {B*T/ms/1e3:8.3f} Mstep/s  {B*T*N/ms/1e6:8.2f} G particle-steps/s  resampled {out['resampled'].mean().item():.2f}", flush=True)
