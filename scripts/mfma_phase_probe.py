"""Ad-hoc: prints per-phase cycle counters of an instrumented MFMA kernel build (not for the committed kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesianfiltering_amd as bfa
from tests import common as cm
F32 = np.float32
B = int(os.environ.get("PB", 512)); T = 50
a = cm.random_stable_lgssm(64, 32, seed=64)
a["Q"] = (1e-2 * np.eye(64)).astype(F32); a["R"] = (1e-1 * np.eye(32)).astype(F32)
p = cm.product_params(a)
y = torch.randn((B, T, 32), device="cuda")
init = torch.zeros((B, 64), device="cuda")
for fields in ((), bfa.FULL5):
    post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True)
    torch.cuda.synchronize()
    carry = post[-1]
    P = carry.covariances.reshape(B, -1)[0].cpu().numpy()
    names = ["A", "B", "C", "E", "F", "G", "H", "I", "J"]
    for w in range(4):
        print("role", w, " ".join(f"{names[i]}={P[w*16+i]/T:7.0f}" for i in range(9)), " total/step", P[w*16:w*16+9].sum() / T)
