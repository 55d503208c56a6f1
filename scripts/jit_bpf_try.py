import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jit_try import compile_src
root = "/root/repo/bayesianfiltering_amd/csrc/"
def hdr(name):
    t = open(root + name).read().replace("#pragma once", "")
    t = t.replace("#include <hip/hip_runtime.h>", "")
    return re.sub(r'^#include "[^"]+"\s*$', "", t, flags=re.M)
PRELUDE = """#define BF_JIT 1
#include <cstdint>
#include <type_traits>
namespace bf {
struct CView { const float* p; long long sB, sT, sE; };
}
"""
BFU = """
namespace bfu {
#pragma clang fp contract(off)
__device__ inline float sin(float x) { return bf::canon_sin(x); }
__device__ inline float cos(float x) { float s, c; bf::canon_sincos(x, &s, &c); return c; }
__device__ inline float exp(float x) { return bf::canon_exp(x); }
__device__ inline float log(float x) { return bf::canon_log(x); }
__device__ inline float sqrt(float x) { return __builtin_sqrtf(x); }
__device__ inline float atan2(float y, float x) { return bf::canon_atan2(y, x); }
__device__ inline float abs(float x) { return __builtin_fabsf(x); }
"""
user = """
template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* th, T* out) {
  out[0] = th[3] * th[0] * (x[1] - x[0]) + x[0] + q[0];
  out[1] = th[3] * (x[0] * th[1] - x[1] - x[0] * x[2]) + x[1] + q[1];
  out[2] = th[3] * (x[0] * x[1] - th[2] * x[2]) + x[2] + q[2];
}
"""
src = PRELUDE + "#define BF_USER_DYN 1\n" + hdr("kf_math.hpp") + hdr("bf_canon_math.hpp") + BFU + user + "}\n" + hdr("scan_common.hpp") + hdr("bf_rng.hpp") + hdr("models.hpp") + hdr("ssm_device.hpp") + hdr("bpf_scan.hpp") + """
extern "C" __global__ void __launch_bounds__(1024) bf_user_bpf(const bf::BpfModel<3, 3, 1>* __restrict__ mdlp, const bf::BpfArgs<3, 3, 1> a) {
  bf::bpf_scan_body<3, 3, 1, 1, 16, bf::SpecUser<true, false, false>>(mdlp);
}
"""
open("/tmp/bpf/jit_bpf.hip", "w").write(src)
rc, log = compile_src(src)
print("rc", rc)
errs = [l for l in log.splitlines() if "error" in l]
print("\n".join(errs[:25]) if errs else log[:1500])
