"""Dev helper: the particle kernel as bf_set_option "bpf_arith" = 1 builds it (registry model, BF_BPF_HW_ARITH), compiled here
(no GPU needed); prints the VGPR / spill summary of the code object."""
import os, re, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
from jit_try import compile_src
root = os.path.join(here, "..", "bayesianfiltering_amd", "csrc")
os.chdir(root)
def text(name):
    t = open(name).read().replace("#pragma once", "").replace("#include <hip/hip_runtime.h>", "")
    return re.sub(r'^#include "[^"]+".*$', "", t, flags=re.M)
um = open("user_model.hip").read()
consts = dict(re.findall(r'const char\* const (\w+) = R"BFSRC\((.*?)\)BFSRC";', um, flags=re.S))
hw = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
fixed = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
n, dq, m, ppt, nw = 16, 16, 8, 4, 16
s = "#define BF_JIT 1\n#include <cstdint>\n#include <type_traits>\n" + ("#define BF_BPF_HW_ARITH 1\n" if hw else "")
s += "#define BF_N %d\n#define BF_DQ %d\n#define BF_M %d\n#define BF_DR %d\n" % (n, dq, m, m)
s += """namespace bf { struct CView { const float* p; long long sB, sT, sE; };
struct SView { float* p; long long sB, sK, sT, sE; };
struct OutViews { SView w, m, P, pm, pP, ll; SView cm, cP; };
struct CarryView { const float* w_in; const float* m_in; const float* P_in; float* w_out; float* m_out; float* P_out; }; }
"""
s += text("kf_math.hpp") + text("bf_canon_math.hpp") + consts["kSamplingUserMath"] + "}  // namespace bfu\n"
for h in ("scan_common.hpp", "bf_rng.hpp", "models.hpp", "ssm_device.hpp", "bpf_scan.hpp"):
    s += text(h)
spec = "bf::SpecFixed<bf::DYN_LORENZ96, bf::EMI_LINEAR, true, true, true, true>" if fixed else "bf::SpecRuntime"
s += """extern "C" __global__ void __launch_bounds__(%d) bf_user_bpf(const bf::BpfModel<BF_N, BF_DQ, BF_M>* __restrict__ mdlp, const bf::BpfArgs<BF_N, BF_DQ, BF_M> args_by_value) {
  (void)args_by_value;
  bf::bpf_scan_body<BF_N, BF_DQ, BF_M, %d, %d, %s>(mdlp);
}
""" % (64 * nw, ppt, nw, spec)
import ctypes as C
rc, log = compile_src(s)
print("rc", rc)
errs = [l for l in log.splitlines() if "error" in l]
print("\n".join(errs[:25]) if errs else log[:800])
