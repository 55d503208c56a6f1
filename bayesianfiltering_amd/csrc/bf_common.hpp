// Internal helpers shared by the C-ABI translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <atomic>
#include "../../include/bayesfilt.h"

namespace bf {

// thread-local text for bf_last_error()
char* last_error_buf();
int set_error(int code, const char* fmt, ...);

#define BF_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return ::bf::set_error(BF_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                            \
  } while (0)

// A tuning option: a process-wide default (bf_set_option) that ONE call can override (bf_set_call_option arms an override on
// the calling thread; the next filter entry point on that thread reads it and disarms every override when it returns), so
// that one caller's choices never leak into another caller's launches.  Reads behave like the std::atomic<int> it replaces.
enum OptionId { OPT_KF_EMIT_MODE, OPT_KF_LANES, OPT_KF_MFMA_VARIANT, OPT_KF_SMALL_MODE, OPT_FORCE_GENERIC, OPT_GSF_STRUCTURED,
                OPT_BPF_VARIANT, OPT_BPF_HBM_MODE, OPT_BPF_SPEC, OPT_BPF_ARITH, OPT_COUNT };
struct CallOverrides {
  int value[OPT_COUNT];
  bool armed[OPT_COUNT];
};
CallOverrides& call_overrides();   // thread-local
struct Option {
  std::atomic<int> global;
  OptionId id;
  Option(int v, OptionId i) : global(v), id(i) {}
  int load() const {
    const CallOverrides& c = call_overrides();
    return c.armed[id] ? c.value[id] : global.load();
  }
  operator int() const { return load(); }
  Option& operator=(int v) {
    global = v;
    return *this;
  }
};
void begin_call_constants();    // const_cache.hip: constant blocks handed out from here on stay pinned ...
void release_call_constants();  // ... until the entry point returns (its kernels are enqueued by then)
struct CallOptionScope {   // at the top of every filter entry point: overrides live for exactly this call
  CallOptionScope() { begin_call_constants(); }
  ~CallOptionScope() {
    CallOverrides& c = call_overrides();
    for (int i = 0; i < OPT_COUNT; ++i) c.armed[i] = false;
    release_call_constants();
  }
};

// Device-resident copy of a host constant block (model struct, per-step covariance table), cached by content and
// uploaded stream-ordered through pinned staging: no host synchronisation, nothing to free (const_cache.hip).
int device_constants(const void* host, size_t bytes, hipStream_t stream, const void** d_out);

// Device-side view of one strided stream with the component axis folded in by the caller.
struct SView {
  float* p;
  long long sB, sK, sT, sE;
};
struct CView {
  const float* p;
  long long sB, sT, sE;
};

inline SView make_sview(const bf_stream& s) { return SView{s.ptr, s.sB, s.sK, s.sT, s.sE}; }

struct OutViews {
  SView w, m, P, pm, pP, ll;
  SView cm, cP;  // collapsed mean / covariance of the filtered mixture (Gaussian-sum kernel only)
};

struct CarryView {
  const float* w_in;
  const float* m_in;
  const float* P_in;
  float* w_out;
  float* m_out;
  float* P_out;
};

// Layout classification of an output descriptor (see bayesfilt.h, bf_stream).
enum Layout { LAYOUT_GENERIC = 0, LAYOUT_REFERENCE = 1, LAYOUT_BATCH_INNER = 2 };

}  // namespace bf
