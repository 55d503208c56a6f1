// user_model: dynamics / emission functions given as SOURCE TEXT at run time.
//
// The reference takes arbitrary Python callables f(x, q, u), h(x, r, u) (gaussfiltax/models.py:46-49) and differentiates
// them with jacfwd (gaussfiltax/inference.py:328-329).  A Python callable cannot run inside a HIP kernel; what can cross
// the C-ABI is the function's source.  bf_user_model_create compiles, with hiprtc, the run-time-dimension scan kernel
// (generic_device.hpp, whose text is embedded in this library) together with the caller's
//     template <class T> __device__ void dynamics(const T* x, const T* q, T u, const float* theta, T* out);
//     template <class T> __device__ void emission(const T* x, const T* r, T u, const float* theta, T* out);
// and the Jacobians come from forward-mode dual numbers (T = bfu::Dual), i.e. exactly what jacfwd computes: one lane per
// seed direction.  Code objects are cached by source hash, in memory and on disk.  hiprtc is loaded lazily (dlopen), from
// next to the HIP runtime the process already uses, so the library itself carries no link-time dependency on it.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "bf_common.hpp"
#include "bpf_big.hpp"    // (brings bpf_scan.hpp) BpfArgs / BpfCarry / BpfOut / BigScratch, the run-time-dimension model fill
#include "ugsf_scan.hpp"  // UkfModelView, fill_ukf_model_view
#include "agsf_geom.hpp"  // AgsfOut, agsf_lds_bytes

#ifndef BF_ARCH_NAME
#define BF_ARCH_NAME "gfx950"
#endif

struct bf_user_model {
  hipModule_t mod = nullptr;              // the Gaussian-sum scan with dual-number Jacobians (built when f or h is given)
  hipFunction_t k64 = nullptr, k256 = nullptr;
  int n = 0, dq = 0, m = 0, dr = 0;
  int device = -1;
  bool has_dyn = false, has_emi = false, has_lp = false;
  bool hw_arith = false;                  // internal handle of bf_set_option "bpf_arith" = 1: registry functions, hardware transcendentals
  std::string dyn_src, emi_src, lp_src;  // kept: the particle-filter kernels are built on first use, per particle capacity
  std::map<int, hipFunction_t> bpf;       // key = PPT * 100 + NW
  hipFunction_t ugsf = nullptr;           // the unscented Gaussian-sum scan, built on first use
  hipFunction_t gsf_regs = nullptr;       // the Gaussian-sum scan with the state in registers (n <= 8), built on first use
  hipFunction_t sample = nullptr;         // NonlinearSSM.sample, built on first use
  hipFunction_t bpf_big = nullptr;        // the particle filter with the particles in HBM, built on first use
  std::map<int, hipFunction_t> agsf;      // the augmented Gaussian-sum scan; key = kind * 100 + waves per trajectory
  std::vector<hipModule_t> extra_mods;
};

namespace bf {

extern const char* const kGenericDeviceSource;  // generic_device.hpp, embedded at build time (jit_sources.hip)
extern const char* const kSamplingSourceA;      // kf_math.hpp + bf_canon_math.hpp
extern const char* const kSamplingSourceB;
extern const char* const kAgsfSource;           // agsf_scan.hpp
extern const char* const kSampleSource;         // sample_ssm.hpp
extern const char* const kBpfBigSource;         // bpf_big.hpp
extern const char* const kUgsfSource;           // ugsf_scan.hpp      // scan_common / bf_rng / models / ssm_device / bpf_scan

namespace {

// ---- hiprtc, resolved at first use
typedef struct _hiprtcProgram* hiprtcProgram;
struct Rtc {
  void* h = nullptr;
  int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
  int (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
  int (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetCode)(hiprtcProgram, char*) = nullptr;
  int (*DestroyProgram)(hiprtcProgram*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*Version)(int*, int*) = nullptr;
};
Rtc g_rtc;
std::mutex g_mu;
std::map<std::string, bf_user_model*> g_models;  // by source hash: a model compiled once is shared (never freed while cached)

bool load_rtc(std::string& why) {
  if (g_rtc.h) return true;
  std::vector<std::string> cand;
  const char* forced = std::getenv("BAYESFILT_HIPRTC_LIB");   // when set: this library and no other
  if (forced && *forced) {
    cand.push_back(forced);
  } else {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipModuleLoadData), &info) && info.dli_fname) {  // next to the runtime in use
      std::string p(info.dli_fname);
      const size_t slash = p.rfind('/');
      if (slash != std::string::npos) cand.push_back(p.substr(0, slash + 1) + "libhiprtc.so");
    }
    cand.push_back("libhiprtc.so");
    cand.push_back("libhiprtc.so.7");
    cand.push_back("/opt/rocm/lib/libhiprtc.so");
  }
  for (const std::string& c : cand) {
    void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      const char* de = dlerror();  // ONE call: dlerror() clears the message it returns
      why += c + ": " + (de ? de : "?") + "; ";
      continue;
    }
#define BF_RTC_SYM(F_) *reinterpret_cast<void**>(&g_rtc.F_) = dlsym(h, "hiprtc" #F_)
    BF_RTC_SYM(CreateProgram); BF_RTC_SYM(CompileProgram); BF_RTC_SYM(GetProgramLogSize); BF_RTC_SYM(GetProgramLog);
    BF_RTC_SYM(GetCodeSize); BF_RTC_SYM(GetCode); BF_RTC_SYM(DestroyProgram); BF_RTC_SYM(GetErrorString); BF_RTC_SYM(Version);
#undef BF_RTC_SYM
    if (g_rtc.CreateProgram && g_rtc.CompileProgram && g_rtc.GetCodeSize && g_rtc.GetCode && g_rtc.DestroyProgram) {
      g_rtc.h = h;
      return true;
    }
    why += c + ": hiprtc entry points missing; ";
    dlclose(h);
  }
  return false;
}

uint64_t fnv1a(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h;
}

// forward-mode dual numbers + the elementary functions a model may call, for float and Dual alike (namespace bfu: the
// caller's source is compiled inside it, so unqualified sin / exp / sqrt ... resolve for both instantiations)
const char* const kDualCore = R"BFSRC(
struct Dual {
  float v, d;
  __device__ Dual() : v(0.f), d(0.f) {}
  __device__ Dual(float a) : v(a), d(0.f) {}
  __device__ Dual(float a, float b) : v(a), d(b) {}
};
#pragma clang fp contract(off)
__device__ inline Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ inline Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ inline Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ inline Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
__device__ inline Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ inline Dual operator+(Dual a) { return a; }
__device__ inline Dual operator+(Dual a, float b) { return Dual(a.v + b, a.d); }
__device__ inline Dual operator+(float a, Dual b) { return Dual(a + b.v, b.d); }
__device__ inline Dual operator-(Dual a, float b) { return Dual(a.v - b, a.d); }
__device__ inline Dual operator-(float a, Dual b) { return Dual(a - b.v, -b.d); }
__device__ inline Dual operator*(Dual a, float b) { return Dual(a.v * b, a.d * b); }
__device__ inline Dual operator*(float a, Dual b) { return Dual(a * b.v, a * b.d); }
__device__ inline Dual operator/(Dual a, float b) { return Dual(a.v / b, a.d / b); }
__device__ inline Dual operator/(float a, Dual b) { const float q = a / b.v; return Dual(q, -q * b.d / b.v); }
__device__ inline Dual& operator+=(Dual& a, Dual b) { a = a + b; return a; }
__device__ inline Dual& operator-=(Dual& a, Dual b) { a = a - b; return a; }
__device__ inline Dual& operator*=(Dual& a, Dual b) { a = a * b; return a; }
__device__ inline Dual& operator/=(Dual& a, Dual b) { a = a / b; return a; }
__device__ inline bool operator<(Dual a, Dual b) { return a.v < b.v; }
__device__ inline bool operator>(Dual a, Dual b) { return a.v > b.v; }
__device__ inline bool operator<=(Dual a, Dual b) { return a.v <= b.v; }
__device__ inline bool operator>=(Dual a, Dual b) { return a.v >= b.v; }
__device__ inline bool operator==(Dual a, Dual b) { return a.v == b.v; }
__device__ inline bool operator!=(Dual a, Dual b) { return a.v != b.v; }
)BFSRC";
const char* const kLibmMath = R"BFSRC(
__device__ inline float sin(float x) { return ::sinf(x); }
__device__ inline float cos(float x) { return ::cosf(x); }
__device__ inline float tan(float x) { return ::tanf(x); }
__device__ inline float exp(float x) { return ::expf(x); }
__device__ inline float log(float x) { return ::logf(x); }
__device__ inline float sqrt(float x) { return ::sqrtf(x); }
__device__ inline float tanh(float x) { return ::tanhf(x); }
__device__ inline float atan(float x) { return ::atanf(x); }
__device__ inline float atan2(float y, float x) { return ::atan2f(y, x); }
__device__ inline float pow(float x, float p) { return ::powf(x, p); }
__device__ inline float abs(float x) { return ::fabsf(x); }
__device__ inline void sincos(float x, float* s, float* c) { *s = ::sinf(x); *c = ::cosf(x); }
__device__ inline float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
)BFSRC";
// (on top of whichever float functions precede it: libm's for the extended-Kalman scan, the canonical ones for the sampling kernels)
const char* const kDualMath = R"BFSRC(
__device__ inline Dual sin(Dual x) { return Dual(sin(x.v), cos(x.v) * x.d); }
__device__ inline Dual cos(Dual x) { return Dual(cos(x.v), -sin(x.v) * x.d); }
__device__ inline Dual tan(Dual x) { const float t = tan(x.v); return Dual(t, (1.f + t * t) * x.d); }
__device__ inline Dual exp(Dual x) { const float e = exp(x.v); return Dual(e, e * x.d); }
__device__ inline Dual log(Dual x) { return Dual(log(x.v), x.d / x.v); }
__device__ inline Dual sqrt(Dual x) { const float s = sqrt(x.v); return Dual(s, x.d / (2.f * s)); }
__device__ inline Dual tanh(Dual x) { const float t = tanh(x.v); return Dual(t, (1.f - t * t) * x.d); }
__device__ inline Dual atan(Dual x) { return Dual(atan(x.v), x.d / (1.f + x.v * x.v)); }
__device__ inline Dual atan2(Dual y, Dual x) { const float r2 = x.v * x.v + y.v * y.v; return Dual(atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / r2); }
__device__ inline Dual pow(Dual x, float p) { const float w = pow(x.v, p - 1.f); return Dual(w * x.v, p * w * x.d); }
__device__ inline Dual abs(Dual x) { return x.v < 0.f ? -x : x; }
__device__ inline void sincos(Dual x, Dual* s, Dual* c) { float sv, cv; sincos(x.v, &sv, &cv); *s = Dual(sv, cv * x.d); *c = Dual(cv, -sv * x.d); }
__device__ inline Dual fma(Dual a, Dual b, Dual c) { return a * b + c; }
__device__ inline Dual fma(float a, Dual b, Dual c) { return a * b + c; }
__device__ inline Dual fma(Dual a, float b, Dual c) { return a * b + c; }
)BFSRC";

std::string build_source(const char* dyn_src, const char* emi_src, int n, int dq, int m, int dr) {
  std::string s;
  s += "#define BF_JIT 1\n";
  if (dyn_src) s += "#define BF_USER_DYN 1\n";
  if (emi_src) s += "#define BF_USER_EMI 1\n";
  s += "#define BF_N " + std::to_string(n) + "\n#define BF_DQ " + std::to_string(dq) + "\n#define BF_M " + std::to_string(m) +
       "\n#define BF_DR " + std::to_string(dr) + "\n";
  s += "namespace bfu {\n";
  s += kDualCore;
  s += kLibmMath;
  s += kDualMath;
  s += "\n// ---- the caller's functions\n";
  if (dyn_src) s += std::string(dyn_src) + "\n";
  if (emi_src) s += std::string(emi_src) + "\n";
  s += "}  // namespace bfu\n";
  s += kGenericDeviceSource;
  s += R"BFSRC(
extern "C" __global__ void __launch_bounds__(64) bf_user_scan_64(bf::GenModel p, bf::CView y, bf::UViewG u, bf::CarryView carry,
    bf::OutViews out, float* gm, float* gP, long long B, long long T, int K, int KP) {
  bf::gsf_generic_body<64>(p, y, u, carry, out, gm, gP, B, T, K, KP);
}
extern "C" __global__ void __launch_bounds__(256) bf_user_scan_256(bf::GenModel p, bf::CView y, bf::UViewG u, bf::CarryView carry,
    bf::OutViews out, float* gm, float* gP, long long B, long long T, int K, int KP) {
  bf::gsf_generic_body<256>(p, y, u, carry, out, gm, gP, B, T, K, KP);
}
)BFSRC";
  return s;
}

std::string cache_dir() {  // $BAYESFILT_CACHE_DIR, else .jit_cache next to this library
  const char* e = std::getenv("BAYESFILT_CACHE_DIR");
  std::string d;
  if (e && *e) {
    d = e;
  } else {
    Dl_info info;
    d = ".";
    if (dladdr(reinterpret_cast<void*>(&bf::set_error), &info) && info.dli_fname) {
      const std::string p(info.dli_fname);
      const size_t slash = p.rfind('/');
      if (slash != std::string::npos) d = p.substr(0, slash);
    }
    d += "/.jit_cache";
  }
  mkdir(d.c_str(), 0755);
  return d;
}

}  // namespace

int check_user_model(const bf_user_model* um, const bf_model* p) {
  // the JIT kernel indexes LDS and registers with the COMPILE-TIME dimensions of bf_user_model_create while the launch
  // carves LDS from the run-time bf_model: they must be the same model
  if (um->n != p->n || um->dq != p->dq || um->m != p->m || um->dr != p->dr)
    return set_error(BF_EINVAL, "bf_model.user was compiled for (n, dq, m, dr) = (%d, %d, %d, %d) but the model says (%d, %d, %d, %d)",
                     um->n, um->dq, um->m, um->dr, p->n, p->dq, p->m, p->dr);
  if (p->dyn_id == BF_FN_USER && !um->has_dyn)
    return set_error(BF_EINVAL, "dyn_id = BF_FN_USER but bf_model.user was created without dynamics source");
  if (p->emi_id == BF_FN_USER && !um->has_emi)
    return set_error(BF_EINVAL, "emi_id = BF_FN_USER but bf_model.user was created without emission source");
  if (p->dyn_id != BF_FN_USER && um->has_dyn)
    return set_error(BF_EINVAL, "bf_model.user holds dynamics source: dyn_id must be BF_FN_USER");
  if (p->emi_id != BF_FN_USER && um->has_emi)
    return set_error(BF_EINVAL, "bf_model.user holds emission source: emi_id must be BF_FN_USER");
  return BF_OK;
}

int launch_user_kernel(const bf_user_model* um, int nt, unsigned grid, size_t lds_bytes, hipStream_t stream, void** args) {
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device)
    return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d (create one handle per device)", um->device, dev);
  hipFunction_t f = nt == 64 ? um->k64 : um->k256;
  // (a module function needs no opt-in for more than 64 KiB of dynamic LDS on gfx950: the launch itself checks the 160 KiB limit)
  BF_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, (unsigned)nt, 1, 1, (unsigned)lds_bytes, stream, args, nullptr));
  return BF_OK;
}

int launch_bpf_user_impl(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess,
                         int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream);

namespace {

bool read_file(const std::string& path, std::vector<char>& code) {
  code.clear();
  if (FILE* f = std::fopen(path.c_str(), "rb")) {
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz > 0) {
      code.resize((size_t)sz);
      if (std::fread(code.data(), 1, (size_t)sz, f) != (size_t)sz) code.clear();
    }
    std::fclose(f);
  }
  return !code.empty();
}

// every rank of a torchrun job misses at the same moment: each writes its OWN temporary (pid + counter) and renames it over
// the final name -- rename is atomic, a reader sees either nothing or a whole file
void write_file_atomically(const std::string& path, const std::vector<char>& code) {
  static std::atomic<unsigned> counter{0};
  const std::string tmp = path + "." + std::to_string((long long)getpid()) + "." + std::to_string(counter.fetch_add(1)) + ".tmp";
  if (FILE* f = std::fopen(tmp.c_str(), "wb")) {  // best effort
    const bool ok = std::fwrite(code.data(), 1, code.size(), f) == code.size();
    const bool closed = std::fclose(f) == 0;
    if (ok && closed && std::rename(tmp.c_str(), path.c_str()) == 0) return;
    std::remove(tmp.c_str());
  }
}

int compile_with_hiprtc(const std::string& src, std::vector<char>& code, bool contract_off = true) {
  std::string why;
  if (!load_rtc(why)) return set_error(BF_EUNSUPPORTED, "hiprtc is not available: %.400s", why.c_str());
  hiprtcProgram prog = nullptr;
  int rc = g_rtc.CreateProgram(&prog, src.c_str(), "bf_user_model.hip", 0, nullptr, nullptr);
  if (rc != 0) return set_error(BF_EHIP, "hiprtcCreateProgram failed (%d)", rc);
  // (contract_off = false: the translation unit keeps hipcc's default contraction -- what the ahead-of-time build of the same
  // kernel was compiled with -- and the source itself switches contraction off around the caller's functions)
  const char* opts[] = {"--offload-arch=" BF_ARCH_NAME, "-O3", "-std=c++17", "-ffp-contract=off"};
  rc = g_rtc.CompileProgram(prog, contract_off ? 4 : 3, opts);
  if (rc != 0) {
    size_t ls = 0;
    std::string log;
    if (g_rtc.GetProgramLogSize && g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) {
      log.resize(ls);
      g_rtc.GetProgramLog(prog, &log[0]);
    }
    g_rtc.DestroyProgram(&prog);
    // the first error lines are what the author of the source needs
    const size_t pos = log.find("error");
    return set_error(BF_EINVAL, "the model source does not compile: %.440s", (pos == std::string::npos ? log : log.substr(pos)).c_str());
  }
  size_t cs = 0;
  rc = g_rtc.GetCodeSize(prog, &cs);
  if (rc == 0 && cs > 0) {
    code.resize(cs);
    rc = g_rtc.GetCode(prog, code.data());
  }
  g_rtc.DestroyProgram(&prog);
  if (rc != 0 || code.empty()) return set_error(BF_EHIP, "hiprtc returned no code object (%d)", rc);
  return BF_OK;
}

hipError_t load_module(bf_user_model* um, const std::vector<char>& code) {
  hipError_t e = hipModuleLoadData(&um->mod, code.data());
  if (e == hipSuccess) e = hipModuleGetFunction(&um->k64, um->mod, "bf_user_scan_64");
  if (e == hipSuccess) e = hipModuleGetFunction(&um->k256, um->mod, "bf_user_scan_256");
  if (e != hipSuccess && um->mod) {
    (void)hipModuleUnload(um->mod);
    um->mod = nullptr;
  }
  return e;
}

// source -> code object through the disk cache (a cached file that does not load is deleted and rebuilt) -> one kernel
int build_function(const std::string& src, const char* kernel_name, hipModule_t* mod, hipFunction_t* fn, bool contract_off = true) {
  int rtver = 0;
  (void)hipRuntimeGetVersion(&rtver);
  char key[32];
  std::snprintf(key, sizeof(key), "%016llx", (unsigned long long)fnv1a(src + "|" BF_ARCH_NAME "|" + std::to_string(rtver)));
  const std::string path = cache_dir() + "/user_" + key + "_" BF_ARCH_NAME ".co";
  std::vector<char> code;
  hipError_t e = hipErrorUnknown;
  *mod = nullptr;
  auto load = [&]() {
    hipError_t le = hipModuleLoadData(mod, code.data());
    if (le == hipSuccess) le = hipModuleGetFunction(fn, *mod, kernel_name);
    if (le != hipSuccess && *mod) {
      (void)hipModuleUnload(*mod);
      *mod = nullptr;
    }
    return le;
  };
  if (read_file(path, code)) {
    e = load();
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return set_error(BF_ENOGPU, "loading the compiled model failed: %s", hipGetErrorString(e));
      std::remove(path.c_str());
      code.clear();
    }
  }
  if (code.empty()) {
    const int rc = compile_with_hiprtc(src, code, contract_off);
    if (rc != BF_OK) return rc;
    write_file_atomically(path, code);
    e = load();
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_error(e == hipErrorNoDevice ? BF_ENOGPU : BF_EHIP, "loading the compiled model failed: %s", hipGetErrorString(e));
  }
  return BF_OK;
}

// The particle-filter kernel (bpf_scan.hpp) with the caller's functions compiled in: state in registers at the compile-time
// dimensions of the handle, one entry per particle capacity.  The caller's functions see the CANONICAL arithmetic of the
// weight path (bf_canon_math.hpp: sin / cos / atan2 / exp / log as defined there, IEEE sqrt, no contraction), so a function
// written like its registry twin gives the registry twin's bits.
const char* const kSamplingUserMath = R"BFSRC(
#pragma clang fp contract(off)   // no contraction in the caller's functions, as in the registry's
namespace bfu {
__device__ inline float sin(float x) { return bf::canon_sin(x); }
__device__ inline float cos(float x) { float s, c; bf::canon_sincos(x, &s, &c); return c; }
__device__ inline void sincos(float x, float* s, float* c) { bf::canon_sincos(x, s, c); }
__device__ inline float exp(float x) { return bf::canon_exp(x); }
__device__ inline float log(float x) { return bf::canon_log(x); }
__device__ inline float sqrt(float x) { return __builtin_sqrtf(x); }
__device__ inline float atan2(float y, float x) { return bf::canon_atan2(y, x); }
__device__ inline float atan(float x) { return bf::canon_atan(x); }
__device__ inline float abs(float x) { return __builtin_fabsf(x); }
__device__ inline float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline float tan(float x) { return ::tanf(x); }       // (no canonical definition: none of the sampling paths' twins needs one)
__device__ inline float tanh(float x) { return ::tanhf(x); }
__device__ inline float pow(float x, float p) { return ::powf(x, p); }
)BFSRC";

enum { JIT_BPF = 0, JIT_UGSF = 1, JIT_AGSF_UKF = 2, JIT_AGSF_EKF = 3, JIT_GSF_REGS = 4, JIT_SAMPLE = 5, JIT_BPF_BIG = 6 };

std::string build_bpf_source(const bf_user_model* um, int ppt, int nw, int kind = JIT_BPF, const char* spec_override = nullptr) {
  std::string s = "#define BF_JIT 1\n#include <cstdint>\n#include <type_traits>\n";
  if (um->hw_arith) s += "#define BF_BPF_HW_ARITH 1\n";
  if (um->has_dyn) s += "#define BF_USER_DYN 1\n";
  if (um->has_emi) s += "#define BF_USER_EMI 1\n";
  if (um->has_lp) s += "#define BF_USER_LP 1\n";
  s += "#define BF_N " + std::to_string(um->n) + "\n#define BF_DQ " + std::to_string(um->dq) + "\n#define BF_M " + std::to_string(um->m) +
       "\n#define BF_DR " + std::to_string(um->dr) + "\n";
  s += "namespace bf { struct CView { const float* p; long long sB, sT, sE; };\n"
       "struct SView { float* p; long long sB, sK, sT, sE; };\n"
       "struct OutViews { SView w, m, P, pm, pP, ll; SView cm, cP; };\n"
       "struct CarryView { const float* w_in; const float* m_in; const float* P_in; float* w_out; float* m_out; float* P_out; }; }\n";
  if (kind == JIT_AGSF_EKF || kind == JIT_GSF_REGS) s += "#define BF_USER_EKF_NODES 1\n";
  s += kSamplingSourceA;
  s += kSamplingUserMath;
  if (kind == JIT_AGSF_EKF || kind == JIT_GSF_REGS) {  // the Jacobians of the extended-Kalman nodes: dual numbers over the same float functions
    s += kDualCore;
    s += kDualMath;
  }
  s += "\n// ---- the caller's functions\n";
  if (um->has_dyn) s += um->dyn_src + "\n";
  if (um->has_emi) s += um->emi_src + "\n";
  if (um->has_lp) s += um->lp_src + "\n";
  s += "}  // namespace bfu\n";
  s += kSamplingSourceB;
  const std::string spec = spec_override ? std::string(spec_override)
                                         : std::string("bf::SpecUser<") + (um->has_dyn ? "true" : "false") + ", " + (um->has_emi ? "true" : "false") + ", " +
                                               (um->has_lp ? "true" : "false") + ">";
  if (kind == JIT_AGSF_UKF || kind == JIT_AGSF_EKF) {
    s += "namespace bf { struct UView { const float* p; long long sB, sT; }; }\n";
    s += kUgsfSource;
    s += kAgsfSource;
    const std::string nodes = kind == JIT_AGSF_UKF ? "bf::UkfNodes<BF_N, BF_DQ, BF_M, BF_DR, " + spec + ">" : "bf::UserEkfNodes<BF_N, BF_DQ, BF_M, BF_DR, " + spec + ">";
    s += "extern \"C\" __global__ void __launch_bounds__(" + std::to_string(nw == 1 ? 256 : 64 * nw) + ") bf_user_agsf(const bf::UkfModel<BF_N, BF_DQ, BF_M, BF_DR>* "
         "__restrict__ mdlp, bf::CView y, bf::UView uin, bf::CarryView carry, bf::AgsfOut out, long long B, long long T, int N0, int N1, int N2, int MP, "
         "float a0, float a1, uint32_t key0, uint32_t key1, int variant, int carry_records, const float* __restrict__ tvq, const float* __restrict__ tvr) {\n"
         "  bf::agsf_scan_body<BF_N, BF_M, " + nodes + ", " + std::to_string(nw) + ">(mdlp, y, uin, carry, out, B, T, N0, N1, N2, MP, a0, a1, key0, key1, "
         "variant, carry_records, tvq, tvr);\n}\n";
    return s;
  }
  if (kind == JIT_BPF_BIG) {   // the particle filter with the particles in HBM (bpf_big.hpp): up to 2^20 particles per trajectory
    s += kBpfBigSource;
    s += "extern \"C\" __global__ void __launch_bounds__(1024) bf_user_bpf_big(const bf::BpfModel<BF_N, BF_DQ, BF_M>* __restrict__ mdlp, bf::CView y, "
         "const float* __restrict__ uptr, long long u_sB, long long u_sT, bf::BpfCarry carry, bf::BpfOut out, bf::BigScratch sc, long long B, long long T, "
         "int NP, float ess_threshold, int resampler, uint32_t key0, uint32_t key1) {\n  bf::bpf_big_body<BF_N, BF_DQ, BF_M, " + spec +
         ">(mdlp, y, uptr, u_sB, u_sT, carry, out, sc, B, T, NP, ess_threshold, resampler, key0, key1);\n}\n";
    return s;
  }
  if (kind == JIT_SAMPLE) {   // NonlinearSSM.sample with the caller's functions (sample_ssm.hpp), a lane per trajectory
    s += kSampleSource;
    s += "extern \"C\" __global__ void __launch_bounds__(64) bf_user_sample(const bf::BpfModel<BF_N, BF_DQ, BF_M>* __restrict__ mdlp, "
         "const bf::EmissionNoise<BF_M>* __restrict__ enp, const uint32_t* __restrict__ keys, const float* __restrict__ uptr, long long u_sB, long long u_sT, "
         "float* __restrict__ states, float* __restrict__ emis, long long B, long long T) {\n  bf::sample_ssm_body<BF_N, BF_DQ, BF_M, " + spec +
         ">(mdlp, enp, keys, uptr, u_sB, u_sT, states, emis, B, T);\n}\n";
    return s;
  }
  if (kind == JIT_GSF_REGS) {   // the Gaussian-sum scan with extended-Kalman operations, one lane per (trajectory, component)
    s += kUgsfSource;
    s += "extern \"C\" __global__ void __launch_bounds__(256) bf_user_gsf_regs(const bf::UkfModel<BF_N, BF_DQ, BF_M, BF_DR>* __restrict__ mdlp, bf::CView y, "
         "const float* __restrict__ uptr, long long u_sB, long long u_sT, bf::CarryView carry, bf::OutViews out, long long B, long long T, int K, int KP, "
         "const float* __restrict__ tvq, const float* __restrict__ tvr) {\n  bf::ugsf_scan_body<BF_N, BF_DQ, BF_M, BF_DR, " + spec +
         ", bf::UserEkfNodes<BF_N, BF_DQ, BF_M, BF_DR, " + spec + ">>(mdlp, y, uptr, u_sB, u_sT, carry, out, B, T, K, KP, tvq, tvr);\n}\n";
    return s;
  }
  if (kind == JIT_UGSF) {
    s += kUgsfSource;
    s += "extern \"C\" __global__ void __launch_bounds__(256) bf_user_ugsf(const bf::UkfModel<BF_N, BF_DQ, BF_M, BF_DR>* __restrict__ mdlp, bf::CView y, "
         "const float* __restrict__ uptr, long long u_sB, long long u_sT, bf::CarryView carry, bf::OutViews out, long long B, long long T, int K, int KP, "
         "const float* __restrict__ tvsq, const float* __restrict__ tvsr) {\n  bf::ugsf_scan_body<BF_N, BF_DQ, BF_M, BF_DR, " + spec +
         ">(mdlp, y, uptr, u_sB, u_sT, carry, out, B, T, K, KP, tvsq, tvsr);\n}\n";
    return s;
  }
  s += "extern \"C\" __global__ void __launch_bounds__(" + std::to_string(64 * nw) + ") bf_user_bpf(const bf::BpfModel<BF_N, BF_DQ, BF_M>* __restrict__ mdlp, "
       "const bf::BpfArgs<BF_N, BF_DQ, BF_M> args_by_value) {\n  (void)args_by_value;\n  bf::bpf_scan_body<BF_N, BF_DQ, BF_M, " +
       std::to_string(ppt) + ", " + std::to_string(nw) + ", " + spec + ">(mdlp);\n}\n";
  return s;
}

}  // namespace


static int launch_bpf_jit(bf_user_model* um, const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP,
                          float ess, int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream);

int launch_bpf_user_impl(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess,
                         int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream) {
  return launch_bpf_jit(const_cast<bf_user_model*>(bp->ssm.user), bp, y, u, B, T, NP, ess, resampler, key, carry, o, stream);
}

// An internal handle without sources, per (dimensions, device, arithmetic): the sampling kernels compiled at run time for a
// REGISTRY model -- bf_set_option "bpf_arith" = 1 (v_log_f32 / v_exp_f32 in place of the defined arithmetic: BF_BPF_HW_ARITH in
// bf_canon_math.hpp / bf_rng.hpp), and every (n, dq, m, dr) the compiled instance tables of the particle / unscented / augmented
// kernels do not hold (bpf_scan.hip, ugsf_scan.hip, agsf_ukf.hip, agsf_scan.hip put it into a copy of the model and take the
// from-source path: "not compiled in" becomes "compiled now").
const bf_user_model* registry_jit_handle(const bf_model* p, bool hw_arith) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const std::string mem_key = std::string(hw_arith ? "hw_arith:" : "registry:") + std::to_string(p->n) + "," + std::to_string(p->dq) + "," +
                              std::to_string(p->m) + "," + std::to_string(p->dr) + "@" + std::to_string(dev);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_models.find(mem_key);
  if (it != g_models.end()) return it->second;
  bf_user_model* um = new bf_user_model;
  um->n = p->n; um->dq = p->dq; um->m = p->m; um->dr = p->dr; um->device = dev; um->hw_arith = hw_arith;
  g_models[mem_key] = um;
  return um;
}

int launch_bpf_hw_arith_impl(const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP, float ess,
                             int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream) {
  const bf_user_model* um = registry_jit_handle(&bp->ssm, true);
  if (!um) return set_error(BF_ENOGPU, "no current device");
  return launch_bpf_jit(const_cast<bf_user_model*>(um), bp, y, u, B, T, NP, ess, resampler, key, carry, o, stream);
}

static int launch_bpf_jit(bf_user_model* um, const bf_bpf_model* bp, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int NP,
                          float ess, int resampler, const uint32_t key[2], const bf_bpf_carry* carry, const bf_bpf_out* o, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
  if (um->n != p->n || um->dq != p->dq || um->m != p->m || um->dr != p->dr)
    return set_error(BF_EINVAL, "bf_model.user was compiled for (n, dq, m, dr) = (%d, %d, %d, %d) but the model says (%d, %d, %d, %d)",
                     um->n, um->dq, um->m, um->dr, p->n, p->dq, p->m, p->dr);
  if ((p->dyn_id == BF_FN_USER) != um->has_dyn) return set_error(BF_EINVAL, "dyn_id = BF_FN_USER exactly when bf_model.user holds dynamics source");
  if ((p->emi_id == BF_FN_USER) != um->has_emi) return set_error(BF_EINVAL, "emi_id = BF_FN_USER exactly when bf_model.user holds emission source");
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device) return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d", um->device, dev);
  const int N = p->n, M = p->m;
  // smallest particle capacity that holds NP (the geometries of bpf_scan.hpp: particles in registers)
  int ppt, nw;
  if (NP <= 64) { ppt = 1; nw = 1; }
  else if (NP <= 128) { ppt = 1; nw = 2; }
  else if (NP <= 256) { ppt = 1; nw = 4; }
  else if (NP <= 512) { ppt = 1; nw = 8; }
  else if (NP <= 1024) { ppt = 1; nw = 16; }
  else if (NP <= 4096 && N <= 16) { ppt = 4; nw = 16; }
  else { ppt = 0; nw = 16; }   // beyond the register capacities: the particles live in HBM (bpf_big.hpp), up to 2^20 per trajectory
  // the model, word for word the BpfModel<N, DQ, M> of the kernel
  std::vector<uint32_t> words(bpf_model_words(N, p->dq, M), 0u);
  const int flags = (um->has_dyn ? 1 : 0) | (um->has_emi ? 2 : 0) | (um->has_lp ? 4 : 0);
  const BpfModelView view = bpf_model_view_flat(words.data(), N, p->dq, M);
  int rc = fill_bpf_model_view(bp, view, flags, bp->lp_theta, bp->n_lp_theta);
  if (rc != BF_OK) return rc;
  // registry models (the hardware-arithmetic build): the model structure as a compile-time spec where bpf_scan.hpp has one
  const char* spec = nullptr;
  int spec_id = 0;
  if (um->hw_arith) {
    const bool l96_pick = N == p->dq && N >= 8 && 2 * M <= N + 1 && *view.dyn_id == DYN_LORENZ96 && *view.emi_id == EMI_LINEAR && *view.g_identity &&
                          *view.lq_diag && *view.lr_diag && *view.h_pick;
    spec = l96_pick ? "bf::SpecFixed<bf::DYN_LORENZ96, bf::EMI_LINEAR, true, true, true, true>" : "bf::SpecRuntime";
    spec_id = l96_pick ? 2 : 1;
  }
  const void* dv = nullptr;
  if (ppt == 0) {
    if (NP > 1024 * 1024) return set_error(BF_EUNSUPPORTED, "bootstrap particle filter: %d particles exceed the capacity of %d per trajectory", NP, 1024 * 1024);
    hipFunction_t big = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_mu);
      if (!um->bpf_big) {
        hipModule_t mod = nullptr;
        rc = build_function(build_bpf_source(um, 0, 0, JIT_BPF_BIG, spec), "bf_user_bpf_big", &mod, &um->bpf_big);
        if (rc != BF_OK) return rc;
        um->extra_mods.push_back(mod);
      }
      big = um->bpf_big;
    }
    rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
    if (rc != BF_OK) return rc;
    const size_t per = (size_t)B * NP;
    float* buf = nullptr;
    BF_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&buf), sizeof(float) * per * (2 * (size_t)N + 3) + sizeof(int) * per, stream));
    BigScratch sc;
    sc.xa = buf;
    sc.xb = sc.xa + per * N;
    sc.w = sc.xb + per * N;
    sc.ll = sc.w + per;
    sc.cdf = sc.ll + per;
    sc.anc = reinterpret_cast<int*>(sc.cdf + per);
    CView yv{y->ptr, y->sB, y->sT, y->sE};
    const float* uptr = (u && u->ptr) ? u->ptr : nullptr;
    long long u_sB = u ? u->sB : 0, u_sT = u ? u->sT : 0;
    BpfCarry cr{carry ? carry->x_in : nullptr, carry ? carry->w_in : nullptr, carry ? carry->key_in : nullptr,
                carry ? carry->x_out : nullptr, carry ? carry->w_out : nullptr, carry ? carry->key_out : nullptr};
    BpfOut ov{o->weights, o->w_sB, o->w_sN, o->w_sT, o->particles, o->x_sB, o->x_sN, o->x_sT, o->ancestors, o->mean, o->ess, o->logz, o->resampled};
    uint32_t k0 = key[0], k1 = key[1];
    void* args[] = {&dv, &yv, &uptr, &u_sB, &u_sT, &cr, &ov, &sc, &B, &T, &NP, &ess, &resampler, &k0, &k1};
    const hipError_t le = hipModuleLaunchKernel(big, (unsigned)B, 1, 1, 1024, 1, 1, 0, stream, args, nullptr);
    const hipError_t fe = hipFreeAsync(buf, stream);
    BF_HIP_CHECK(le);
    BF_HIP_CHECK(fe);
    return BF_OK;
  }
  hipFunction_t fn = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    const int fkey = spec_id * 10000 + ppt * 100 + nw;
    auto it = um->bpf.find(fkey);
    if (it != um->bpf.end()) {
      fn = it->second;
    } else {
      hipModule_t mod = nullptr;
      rc = build_function(build_bpf_source(um, ppt, nw, JIT_BPF, spec), "bf_user_bpf", &mod, &fn);
      if (rc != BF_OK) return rc;
      um->extra_mods.push_back(mod);
      um->bpf[fkey] = fn;
    }
  }
  rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  BpfArgs<1, 1, 1> a;   // (the argument struct does not depend on the dimensions)
  std::memset(&a, 0, sizeof(a));
  a.y = CView{y->ptr, y->sB, y->sT, y->sE};
  a.uptr = (u && u->ptr) ? u->ptr : nullptr;
  a.u_sB = u ? u->sB : 0;
  a.u_sT = u ? u->sT : 0;
  a.carry = BpfCarry{carry ? carry->x_in : nullptr, carry ? carry->w_in : nullptr, carry ? carry->key_in : nullptr,
                     carry ? carry->x_out : nullptr, carry ? carry->w_out : nullptr, carry ? carry->key_out : nullptr};
  a.out = BpfOut{o->weights, o->w_sB, o->w_sN, o->w_sT, o->particles, o->x_sB, o->x_sN, o->x_sT, o->ancestors, o->mean, o->ess, o->logz, o->resampled};
  a.B = B; a.T = T; a.NP = NP; a.ess_threshold = ess; a.resampler = resampler; a.key0 = key[0]; a.key1 = key[1];
  const int cap = 64 * nw * ppt, dch = (ppt >= 16) ? 1 : ((N >= 8) ? 8 : N);
  const size_t lds_bytes = sizeof(float) * (size_t)(((cdf_words(cap) + 3) & ~3) + 64 + ((nw * N + 3) & ~3) + cap * dch);
  if (lds_bytes > 160 * 1024) return set_error(BF_EUNSUPPORTED, "particle tile exceeds the 160 KiB LDS");
  struct { const void* mdl; BpfArgs<1, 1, 1> a; } packed{dv, a};   // the kernarg segment: the model pointer, then the struct
  size_t psz = sizeof(packed);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psz, HIP_LAUNCH_PARAM_END};
  BF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)B, 1, 1, (unsigned)(64 * nw), 1, 1, (unsigned)lds_bytes, stream, nullptr, config));
  return BF_OK;
}


// The unscented Gaussian-sum scan (ugsf_scan.hpp: a lane per (trajectory, component), sigma points through f and h) with the
// caller's functions: state in registers, so the state dimension is bounded like the compiled instances' (n <= 8)
int launch_ugsf_user_impl(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                          int K, const bf_carry* carry, const bf_out_desc* out, hipStream_t stream) {
  bf_user_model* um = const_cast<bf_user_model*>(p->user);
  int rc = check_user_model(um, p);
  if (rc != BF_OK) return rc;
  if (um->has_lp) return set_error(BF_EINVAL, "a log-density from source belongs to the particle filter, not to the unscented filter");
  if (p->n > 8 || p->dq > 8 || p->m > 8 || p->dr > 8)
    return set_error(BF_EUNSUPPORTED, "unscented filter with functions from source: dimensions up to 8 (the sigma points live in registers)");
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device) return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d", um->device, dev);
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  int KP = 1;
  while (KP < K) KP <<= 1;
  if (KP > 256) return set_error(BF_EUNSUPPORTED, "unscented Gaussian-sum filter: %d components exceed one workgroup (256 lanes)", K);
  if (out->coll_mean.ptr || out->coll_cov.ptr) return set_error(BF_EUNSUPPORTED, "collapsed streams are produced by bf_gsf_ekf_f32 only");
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!um->ugsf) {
      hipModule_t mod = nullptr;
      rc = build_function(build_bpf_source(um, 0, 0, JIT_UGSF), "bf_user_ugsf", &mod, &um->ugsf);
      if (rc != BF_OK) return rc;
      um->extra_mods.push_back(mod);
    }
  }
  std::vector<uint32_t> words(ukf_model_words(p->n, p->dq, p->m, p->dr), 0u);
  std::vector<float> tvsq, tvsr;
  rc = fill_ukf_model_view(p, up, ukf_model_view_flat(words.data(), p->n, p->dq, p->m, p->dr), (um->has_dyn ? 1 : 0) | (um->has_emi ? 2 : 0), &tvsq, &tvsr);
  if (rc != BF_OK) return rc;
  const void* dv = nullptr;
  rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float *d_tvsq = nullptr, *d_tvsr = nullptr;
  if ((rc = upload_table(tvsq, stream, &d_tvsq)) != BF_OK || (rc = upload_table(tvsr, stream, &d_tvsr)) != BF_OK) return rc;
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  const float* uptr = (u && u->ptr) ? u->ptr : nullptr;
  long long u_sB = u ? u->sB : 0, u_sT = u ? u->sT : 0;
  void* args[] = {&dv, &yv, &uptr, &u_sB, &u_sT, &cv, &ov, &B, &T, &K, &KP, &d_tvsq, &d_tvsr};
  const int tpb = 256 / KP;
  BF_HIP_CHECK(hipModuleLaunchKernel(um->ugsf, (unsigned)((B + tpb - 1) / tpb), 1, 1, 256, 1, 1, 0, stream, args, nullptr));
  return BF_OK;
}

// NonlinearSSM.sample (gaussfiltax/models.py:240-289) with the caller's f(x, q, u) / h(x, r, u): sample_ssm.hpp compiled around them
// (a lane per trajectory, Threefry draws in JAX's layout -- the keys a reference run would consume)
int launch_sample_user_impl(const bf_bpf_model* bp, const uint32_t* d_keys, const bf_cstream* u, long long B, long long T, float* d_states,
                            float* d_emis, hipStream_t stream) {
  const bf_model* p = &bp->ssm;
  bf_user_model* um = const_cast<bf_user_model*>(p->user);
  int rc = check_user_model(um, p);
  if (rc != BF_OK) return rc;
  if (p->dr != p->m) return set_error(BF_EUNSUPPORTED, "sample_ssm: emission noise dimension must equal the emission dimension");
  if (p->n > 32 || p->dq > 32 || p->m > 32) return set_error(BF_EUNSUPPORTED, "sample_ssm with functions from source: dimensions up to 32 (a trajectory's state lives in registers)");
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device) return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d", um->device, dev);
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!um->sample) {
      hipModule_t mod = nullptr;
      rc = build_function(build_bpf_source(um, 0, 0, JIT_SAMPLE), "bf_user_sample", &mod, &um->sample);
      if (rc != BF_OK) return rc;
      um->extra_mods.push_back(mod);
    }
  }
  const int N = p->n, DQ = p->dq, M = p->m;
  // [BpfModel<N, DQ, M> words][EmissionNoise<M>: 4 ints, Dm (M x M), LRn (M x M), r0 (M)] in one constant block
  const size_t mw = bpf_model_words(N, DQ, M), ew = 4 + 2 * (size_t)M * M + M;
  std::vector<uint32_t> words(mw + ew, 0u);
  bf_bpf_model tmp = *bp;
  tmp.lp_cov = p->R;      // chol(R) through the particle filter's model fill: the emission-noise covariance stands in for the log-density's
  tmp.r_eval = nullptr;
  const BpfModelView view = bpf_model_view_flat(words.data(), N, DQ, M);
  rc = fill_bpf_model_view(&tmp, view, (um->has_dyn ? 1 : 0) | (um->has_emi ? 2 : 0), nullptr, 0);
  if (rc != BF_OK) return rc;
  int* ei = reinterpret_cast<int*>(words.data() + mw);
  float* ef = reinterpret_cast<float*>(words.data() + mw + 4);
  ei[0] = 1;                                            // d_identity
  ei[1] = (!um->has_emi && p->emi_id == EMI_STOCH_VOL) ? 1 : 0;
  if (!um->has_emi && p->emi_id == EMI_LINEAR) {
    ei[0] = 0;
    for (int i = 0; i < M * M; ++i) ef[i] = p->emi_theta[M * N + i];
  }
  for (int i = 0; i < M * M; ++i) ef[M * M + i] = view.LR[i];
  for (int i = 0; i < M; ++i) ef[2 * M * M + i] = p->r0 ? p->r0[i] : 0.f;
  if (ei[1]) return set_error(BF_EUNSUPPORTED, "sample_ssm: the stochastic-volatility emission beside a dynamics function from source is not built; give h as source too");
  const void* dv = nullptr;
  rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const void* d_mdl = dv;
  const void* d_en = static_cast<const uint32_t*>(dv) + mw;
  const float* uptr = (u && u->ptr) ? u->ptr : nullptr;
  long long u_sB = u ? u->sB : 0, u_sT = u ? u->sT : 0;
  void* args[] = {&d_mdl, &d_en, &d_keys, &uptr, &u_sB, &u_sT, &d_states, &d_emis, &B, &T};
  BF_HIP_CHECK(hipModuleLaunchKernel(um->sample, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, stream, args, nullptr));
  return BF_OK;
}

// bf_gsf_ekf_f32 with functions from source and small dimensions: the Gaussian-sum scan of inference.py:333-371 with one lane per
// (trajectory, component), mean and covariance in registers, the Jacobians by dual numbers (ugsf_scan.hpp: UserEkfNodes) -- two
// orders of magnitude faster than the run-time-dimension kernel the same handle also carries (state in LDS, any n), which remains
// the path for n > 8, for a nonlinear registry function beside one from source, legacy flags, collapsed streams and K > 256.
bool gsf_user_regs_eligible(const bf_model* p, int K, const bf_out_desc* out) {
  const bf_user_model* um = p->user;
  if (!um || um->has_lp || p->flags != 0) return false;
  if (p->n > 8 || p->dq > 8 || p->m > 8 || p->dr > 8 || K > 256) return false;
  if (out->coll_mean.ptr || out->coll_cov.ptr) return false;
  if (!(um->has_dyn || p->dyn_id == DYN_LINEAR) || !(um->has_emi || p->emi_id == EMI_LINEAR)) return false;
  return true;
}

int launch_gsf_user_regs_impl(const bf_model* p, const bf_cstream* y, const bf_cstream* u, long long B, long long T, int K, const bf_carry* carry,
                              const bf_out_desc* out, hipStream_t stream) {
  bf_user_model* um = const_cast<bf_user_model*>(p->user);
  int rc = check_user_model(um, p);
  if (rc != BF_OK) return rc;
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device) return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d", um->device, dev);
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  int KP = 1;
  while (KP < K) KP <<= 1;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!um->gsf_regs) {
      hipModule_t mod = nullptr;
      rc = build_function(build_bpf_source(um, 0, 0, JIT_GSF_REGS), "bf_user_gsf_regs", &mod, &um->gsf_regs);
      if (rc != BF_OK) return rc;
      um->extra_mods.push_back(mod);
    }
  }
  std::vector<uint32_t> words(ukf_model_words(p->n, p->dq, p->m, p->dr), 0u);
  std::vector<float> tvq, tvr;
  const bf_ukf_params unit{1.f, 0.f, 0.f};  // (the extended-Kalman operations ignore the unscented constants)
  rc = fill_ukf_model_view(p, &unit, ukf_model_view_flat(words.data(), p->n, p->dq, p->m, p->dr), (um->has_dyn ? 1 : 0) | (um->has_emi ? 2 : 0) | 4, &tvq,
                           &tvr);
  if (rc != BF_OK) return rc;
  const void* dv = nullptr;
  rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float *d_tvq = nullptr, *d_tvr = nullptr;
  if ((rc = upload_table(tvq, stream, &d_tvq)) != BF_OK || (rc = upload_table(tvr, stream, &d_tvr)) != BF_OK) return rc;
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  OutViews ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs),
              make_sview(out->pred_means), make_sview(out->pred_covs), make_sview(out->loglik)};
  const float* uptr = (u && u->ptr) ? u->ptr : nullptr;
  long long u_sB = u ? u->sB : 0, u_sT = u ? u->sT : 0;
  void* args[] = {&dv, &yv, &uptr, &u_sB, &u_sT, &cv, &ov, &B, &T, &K, &KP, &d_tvq, &d_tvr};
  const int tpb = 256 / KP;
  BF_HIP_CHECK(hipModuleLaunchKernel(um->gsf_regs, (unsigned)((B + tpb - 1) / tpb), 1, 1, 256, 1, 1, 0, stream, args, nullptr));
  return BF_OK;
}

// The augmented Gaussian-sum scan (agsf_scan.hpp: a lane per leaf of the [N0, N1, N2] tree) around the caller's functions.
// up != NULL: unscented nodes (speedy_unscented_agsf / unscented_agsf, inference.py:966-1156 / 813-965), either function may
// also come from the registry.  up == NULL: extended-Kalman nodes (inference.py:621-812 / 458-620 / 1157-1300) with the
// Jacobians by dual numbers -- both functions from source (a registry function has its analytic Jacobian in the compiled kernels).
int launch_agsf_user_impl(const bf_model* p, const bf_ukf_params* up, const bf_cstream* y, const bf_cstream* u, long long B, long long T,
                          const int32_t nc[3], const uint32_t key[2], const float opt[2], const bf_carry* carry, const bf_out_desc* out,
                          int* d_leaf_idx, int variant, hipStream_t stream) {
  bf_user_model* um = const_cast<bf_user_model*>(p->user);
  int rc = check_user_model(um, p);
  if (rc != BF_OK) return rc;
  if (um->has_lp) return set_error(BF_EINVAL, "a log-density from source belongs to the particle filter, not to the augmented filter");
  if (!up && !((um->has_dyn || p->dyn_id == DYN_LINEAR) && (um->has_emi || p->emi_id == EMI_LINEAR)))
    return set_error(BF_EUNSUPPORTED, "augmented filter with extended-Kalman nodes: give BOTH functions as source (beside a function from source "
                                      "only the registry's linear one can stand: its Jacobian needs no differentiation)");
  if (p->n > 8 || p->dq > 8 || p->m > 8 || p->dr > 8)
    return set_error(BF_EUNSUPPORTED, "augmented filter with functions from source: dimensions up to 8 (a leaf lives in registers)");
  if (p->flags != 0) return set_error(BF_EUNSUPPORTED, "legacy-class flags do not apply to the augmented filter");
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != um->device) return set_error(BF_EINVAL, "bf_model.user was loaded on device %d, the current device is %d", um->device, dev);
  if ((p->Q_steps > 1 && p->Q_steps != T) || (p->R_steps > 1 && p->R_steps != T))
    return set_error(BF_EINVAL, "time-varying covariances need one matrix per step (Q_steps / R_steps = T = %lld)", T);
  const long long Mleaf = (long long)nc[0] * nc[1] * nc[2];
  if (Mleaf > 1024) return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: %lld leaves per trajectory exceed one workgroup (1024)", Mleaf);
  if (out->pred_means.ptr || out->pred_covs.ptr || out->coll_mean.ptr || out->coll_cov.ptr || out->loglik.ptr)
    return set_error(BF_EINVAL, "the augmented filter emits weights, means and covariances only (inference.py:771-775)");
  int MP = 1;
  while (MP < Mleaf) MP <<= 1;
  int nw = 1;
  if (MP > 64) {
    if (p->n > 4) return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: more than 64 leaves per trajectory need state_dim <= 4");
    nw = MP <= 128 ? 2 : (MP <= 256 ? 4 : (MP <= 512 ? 8 : 16));
    MP = 64 * nw;
  }
  const size_t lds_bytes = agsf_lds_bytes(p->n, nw, nc[0]);
  if (lds_bytes > 160 * 1024)
    return set_error(BF_EUNSUPPORTED, "augmented Gaussian-sum filter: %d leaves and %d components of dimension %d exceed the 160 KiB LDS",
                     (int)Mleaf, nc[0], p->n);
  const int kind = up ? JIT_AGSF_UKF : JIT_AGSF_EKF;
  hipFunction_t fn = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = um->agsf.find(kind * 100 + nw);
    if (it == um->agsf.end()) {
      hipModule_t mod = nullptr;
      rc = build_function(build_bpf_source(um, 0, nw, kind), "bf_user_agsf", &mod, &fn);
      if (rc != BF_OK) return rc;
      um->extra_mods.push_back(mod);
      um->agsf[kind * 100 + nw] = fn;
    } else {
      fn = it->second;
    }
  }
  std::vector<uint32_t> words(ukf_model_words(p->n, p->dq, p->m, p->dr), 0u);
  std::vector<float> tvq, tvr;
  const bf_ukf_params unit{1.f, 0.f, 0.f};  // (the extended-Kalman nodes ignore the unscented constants)
  rc = fill_ukf_model_view(p, up ? up : &unit, ukf_model_view_flat(words.data(), p->n, p->dq, p->m, p->dr),
                           (um->has_dyn ? 1 : 0) | (um->has_emi ? 2 : 0) | (up ? 0 : 4), &tvq, &tvr);
  if (rc != BF_OK) return rc;
  const void* dv = nullptr;
  rc = device_constants(words.data(), sizeof(uint32_t) * words.size(), stream, &dv);
  if (rc != BF_OK) return rc;
  const float *d_tvq = nullptr, *d_tvr = nullptr;
  if ((rc = upload_table(tvq, stream, &d_tvq)) != BF_OK || (rc = upload_table(tvr, stream, &d_tvr)) != BF_OK) return rc;
  CView yv{y->ptr, y->sB, y->sT, y->sE};
  struct { const float* p; long long sB, sT; } uv{u && u->ptr ? u->ptr : nullptr, u ? u->sB : 0, u ? u->sT : 0};  // gsf_scan.hpp: UView
  CarryView cv{carry->w_in, carry->m_in, carry->P_in, carry->w_out, carry->m_out, carry->P_out};
  AgsfOut ov{make_sview(out->weights), make_sview(out->means), make_sview(out->covs), d_leaf_idx};
  const int nt = nw == 1 ? 256 : 64 * nw;
  int carry_records = nw == 1 ? 256 : ((nc[0] + 3) & ~3);
  int N0 = nc[0], N1 = nc[1], N2 = nc[2];
  float a0 = opt[0], a1 = opt[1];
  uint32_t k0 = key[0], k1 = key[1];
  const int tpb = nt / MP;
  void* args[] = {&dv, &yv, &uv, &cv, &ov, &B, &T, &N0, &N1, &N2, &MP, &a0, &a1, &k0, &k1, &variant, &carry_records, &d_tvq, &d_tvr};
  BF_HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)((B + tpb - 1) / tpb), 1, 1, (unsigned)nt, 1, 1, (unsigned)lds_bytes, stream, args, nullptr));
  return BF_OK;
}

}  // namespace bf

extern "C" {

int bf_user_model_create(const char* dynamics_src, const char* emission_src, int32_t n, int32_t dq, int32_t m, int32_t dr,
                         bf_user_model** model) {
  return bf_user_model_create_lp(dynamics_src, emission_src, nullptr, n, dq, m, dr, model);
}

int bf_user_model_create_lp(const char* dynamics_src, const char* emission_src, const char* log_prob_src, int32_t n, int32_t dq,
                            int32_t m, int32_t dr, bf_user_model** model) {
  using namespace bf;
  if (!model || (!dynamics_src && !emission_src && !log_prob_src)) return set_error(BF_EINVAL, "bf_user_model_create: no source given");
  if (n <= 0 || dq <= 0 || m <= 0 || dr <= 0 || n > 64 || dq > 64 || m > 64 || dr > 64)
    return set_error(BF_EINVAL, "bf_user_model_create: dimensions must be in 1..64");
  const bool gsf = dynamics_src || emission_src;   // the Gaussian-sum scan is built now; the particle-filter kernels on first use
  // (the log-density belongs to the particle kernel, built on first use; here it only makes the handle's key unique: every
  // line of it as a comment)
  std::string lp_comment;
  if (log_prob_src) {
    lp_comment = "// lp: ";
    for (const char* c = log_prob_src; *c; ++c) {
      lp_comment += *c;
      if (*c == '\n') lp_comment += "// ";
    }
    lp_comment += "\n";
  }
  const std::string src = build_source(dynamics_src, emission_src, n, dq, m, dr) + lp_comment;
  // the code object depends on the source, the target and the compiler: all three are in the key (the HIP runtime's version
  // stands for hiprtc's, which ships with it -- known without loading hiprtc on a cache hit)
  int rtver = 0;
  (void)hipRuntimeGetVersion(&rtver);
  char key[32];
  std::snprintf(key, sizeof(key), "%016llx", (unsigned long long)fnv1a(src + "|" BF_ARCH_NAME "|" + std::to_string(rtver)));
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    dev = -1;
    (void)hipGetLastError();
  }
  const std::string mem_key = std::string(key) + "@" + std::to_string(dev);   // a module is loaded on ONE device
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_models.find(mem_key);
  if (it != g_models.end()) {
    *model = it->second;
    return BF_OK;
  }
  bf_user_model* um = new bf_user_model;
  um->n = n; um->dq = dq; um->m = m; um->dr = dr; um->device = dev;
  um->has_dyn = dynamics_src != nullptr;
  um->has_emi = emission_src != nullptr;
  um->has_lp = log_prob_src != nullptr;
  if (dynamics_src) um->dyn_src = dynamics_src;
  if (emission_src) um->emi_src = emission_src;
  if (log_prob_src) um->lp_src = log_prob_src;
  if (!gsf) {   // a log-density alone: nothing to build before the first particle-filter call
    g_models[mem_key] = um;
    *model = um;
    return BF_OK;
  }
  // ---- code object: disk cache (a file that does not load -- truncated, stale, foreign -- is deleted and rebuilt), else hiprtc
  std::vector<char> code;
  const std::string path = cache_dir() + "/user_" + key + "_" BF_ARCH_NAME ".co";
  hipError_t e = hipErrorUnknown;
  if (read_file(path, code)) {
    e = load_module(um, code);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || dev < 0) {
        delete um;
        return set_error(BF_ENOGPU, "loading the compiled model failed: %s", hipGetErrorString(e));
      }
      std::remove(path.c_str());
      code.clear();
    }
  }
  if (code.empty()) {
    const int rc = compile_with_hiprtc(src, code);
    if (rc != BF_OK) {
      delete um;
      return rc;
    }
    write_file_atomically(path, code);
    e = load_module(um, code);
  }
  if (e != hipSuccess) {
    delete um;
    (void)hipGetLastError();
    return set_error(e == hipErrorNoDevice ? BF_ENOGPU : BF_EHIP, "loading the compiled model failed: %s", hipGetErrorString(e));
  }
  g_models[mem_key] = um;
  *model = um;
  return BF_OK;
}

void bf_user_model_destroy(bf_user_model* model) {
  (void)model;  // compiled models are shared through the cache and live as long as the process
}

}  // extern "C"
