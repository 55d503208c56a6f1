"""Dev helper: assemble the source user_model.hip: build_bpf_source() hands to hiprtc for the augmented filter (kind = ukf | ekf,
waves per trajectory) around the BOT functions of tests/test_user_model_gpu.py, and compile it here (no GPU needed)."""
import os, re, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
from jit_try import compile_src
root = os.path.join(here, "..", "bayesianfiltering_amd", "csrc")
os.chdir(root)
sys.path.insert(0, root)
import importlib.util
def text(name):
    t = open(name).read().replace("#pragma once", "").replace("#include <hip/hip_runtime.h>", "")
    return re.sub(r'^#include "[^"]+".*$', "", t, flags=re.M)
um = open("user_model.hip").read()
consts = dict(re.findall(r'const char\* const (\w+) = R"BFSRC\((.*?)\)BFSRC";', um, flags=re.S))
kind, nw = (sys.argv[1] if len(sys.argv) > 1 else "ekf"), int(sys.argv[2]) if len(sys.argv) > 2 else 1
t = open(os.path.join(here, "..", "tests", "test_user_model_gpu.py")).read()
dyn = re.search(r'BOT_DYN_SRC = """(.*?)"""', t, flags=re.S).group(1)
emi = re.search(r'BOT_EMI_SRC = """(.*?)"""', t, flags=re.S).group(1)
s = "#define BF_JIT 1\n#include <cstdint>\n#include <type_traits>\n#define BF_USER_DYN 1\n#define BF_USER_EMI 1\n#define BF_N 4\n#define BF_DQ 2\n#define BF_M 2\n#define BF_DR 2\n"
s += """namespace bf { struct CView { const float* p; long long sB, sT, sE; };
struct SView { float* p; long long sB, sK, sT, sE; };
struct OutViews { SView w, m, P, pm, pP, ll; SView cm, cP; };
struct CarryView { const float* w_in; const float* m_in; const float* P_in; float* w_out; float* m_out; float* P_out; }; }
"""
s += ("#define BF_USER_EKF_NODES 1\n" if kind in ("ekf", "gsf") else "") + text("kf_math.hpp") + text("bf_canon_math.hpp") + consts["kSamplingUserMath"]
if kind in ("ekf", "gsf"):
    s += consts["kDualCore"] + consts["kDualMath"]
s += dyn + emi + "}  // namespace bfu\n"
for h in ("scan_common.hpp", "bf_rng.hpp", "models.hpp", "ssm_device.hpp", "bpf_scan.hpp"):
    s += text(h)
s += "namespace bf { struct UView { const float* p; long long sB, sT; }; }\n" + text("ugsf_scan.hpp") + text("agsf_geom.hpp") + text("agsf_scan.hpp")
nodes = "bf::UkfNodes<BF_N, BF_DQ, BF_M, BF_DR, bf::SpecUser<true, true, false>>" if kind == "ukf" else "bf::UserEkfNodes<BF_N, BF_DQ, BF_M, BF_DR, bf::SpecUser<true, true, false>>"
if kind == "gsf":
    s += """extern "C" __global__ void __launch_bounds__(256) bf_user_gsf_regs(const bf::UkfModel<BF_N, BF_DQ, BF_M, BF_DR>* __restrict__ mdlp, bf::CView y,
  const float* __restrict__ uptr, long long u_sB, long long u_sT, bf::CarryView carry, bf::OutViews out, long long B, long long T, int K, int KP,
  const float* __restrict__ tvq, const float* __restrict__ tvr) {
  bf::ugsf_scan_body<BF_N, BF_DQ, BF_M, BF_DR, bf::SpecUser<true, true, false>, %s>(mdlp, y, uptr, u_sB, u_sT, carry, out, B, T, K, KP, tvq, tvr);
}
""" % nodes
    rc, log = compile_src(s)
    print("rc", rc)
    errs = [l for l in log.splitlines() if "error" in l]
    print("\n".join(errs[:25]) if errs else log[:1500])
    sys.exit(0)
s += """extern "C" __global__ void __launch_bounds__(%d) bf_user_agsf(const bf::UkfModel<BF_N, BF_DQ, BF_M, BF_DR>* __restrict__ mdlp, bf::CView y, bf::UView uin,
  bf::CarryView carry, bf::AgsfOut out, long long B, long long T, int N0, int N1, int N2, int MP, float a0, float a1, uint32_t key0, uint32_t key1, int variant,
  int carry_records, const float* __restrict__ tvq, const float* __restrict__ tvr) {
  bf::agsf_scan_body<BF_N, BF_M, %s, %d>(mdlp, y, uin, carry, out, B, T, N0, N1, N2, MP, a0, a1, key0, key1, variant, carry_records, tvq, tvr);
}
""" % (256 if nw == 1 else 64 * nw, nodes, nw)
open("/tmp/jit_agsf.hip", "w").write(s)
rc, log = compile_src(s)
print("rc", rc)
errs = [l for l in log.splitlines() if "error" in l]
print("\n".join(errs[:25]) if errs else log[:1500])
