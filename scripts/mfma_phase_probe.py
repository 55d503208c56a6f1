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
from bayesianfiltering_amd import _lib
_lib.check(_lib.require_gpu().bf_set_option(b"kf_mfma_variant", int(os.environ.get("PV", "3"))))
y = torch.randn((B, T, 32), device="cuda")
init = torch.zeros((B, 64), device="cuda")
for fields in ((), bfa.FULL5):
    post = bfa.kalman_filter(p, y, initial_means=init, fields=fields, return_carry=True)
    torch.cuda.synchronize()
    carry = post[-1]
    P = carry.covariances.reshape(B, -1)[0].cpu().numpy()
    # variant 1: nine phases; variants 2 / 3: (compute, barrier wait) of the five phases A, B+C, H, I, J -- ticks of 10 ns
    names = ["A", "B", "C", "E", "F", "G", "H", "I", "J", "-"] if os.environ.get("PV", "3") == "1" else ["A", "wA", "BC", "wBC", "H", "wH", "I", "wI", "J", "wJ", "S", "-"]
    for w in range(4):
        print("role", w, " ".join(f"{names[i]}={P[w*16+i]/T:7.0f}" for i in range(len(names))), " total/step", P[w*16:w*16+len(names)].sum() / T)
