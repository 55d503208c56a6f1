"""A/B of the two n = 64, m = 32 Kalman kernels on one box: variant 1 (round 1: explicit inverse through LDS) against
variant 2 (gain-free update, factorization in registers); agreement between them and with the oracle's C port."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from bayesianfiltering_amd import _lib
from tests import common as cm
from oracle import c_oracle
F32 = np.float32
B = int(os.environ.get("PB", 4096)); T = int(os.environ.get("PT", 100))
a = cm.random_stable_lgssm(64, 32, seed=64)
a["Q"] = (1e-2 * np.eye(64)).astype(F32); a["R"] = (1e-1 * np.eye(32)).astype(F32); a["m0"] = np.zeros(64, F32)
p = cm.product_params(a)
lib = _lib.require_gpu()
y = cm.device_observations(p, (64, 64, 32, 32), B, T, seed=5)
init = torch.zeros((B, 64), device="cuda")
res = {}
for v in (1, 2, 3, 1, 2, 3):
    _lib.check(lib.bf_set_option(b"kf_mfma_variant", v))
    for fields, name in ((bfa.FULL5, "FULL5"), ((), "none")):
        post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True, return_loglik=True)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            bfa.kalman_filter(p, y, initial_means=init, fields=fields, out=post[0], return_carry=True, return_loglik=True)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        print(f"variant {v} {name:6s} B={B} T={T}: {best:8.2f} ms  {B*T/best/1e3:8.2f} Mstep/s  {2.0e6*B*T/best/1e9:6.1f} TFLOP/s", flush=True)
        if name == "FULL5":
            res[v] = post
ref = c_oracle.kalman_filter(a, y[:2].cpu().numpy(), np.zeros((2, 64), F32))
for v in (1, 2, 3):
    post, ll, carry = res[v]
    print("variant", v, "vs oracle:", {k: float(cm.rel_err(getattr(post, k)[:2].cpu().numpy(), ref[k])) for k in bfa.FULL5},
          "loglik", float(cm.rel_err(ll[:2].cpu().numpy(), ref["loglik"])))
print("1 vs 3:", {k: float(cm.rel_err(getattr(res[1][0], k).cpu().numpy(), getattr(res[3][0], k).cpu().numpy())) for k in bfa.FULL5})
