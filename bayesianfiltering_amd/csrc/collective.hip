// bf_allgather_summaries: the path's one exchange step for callers that hold an RCCL communicator
// themselves (a C / C++ host program; the Python layer uses torch.distributed, whose `nccl` backend is the same RCCL).
// Trajectories are sharded contiguously over the ranks, nothing is exchanged during the scan; afterwards every rank
// contributes `bytes` of per-trajectory posterior summaries and receives all ranks' blocks, rank-major: ONE
// ncclAllGather on the caller's stream.  On the node's point-to-point xGMI mesh (7 links x ~153 GB/s per GPU) RCCL's
// direct all-gather pushes a shard over all links at once; the summaries of BASELINE configs[3] (131 MB per GPU) take
// ~1 ms.  RCCL is resolved at first use -- the copy the process already holds (the caller's communicator came from it),
// else dlopen next to the HIP runtime in use: the library has no link-time dependency on it and single-GPU users never
// load it.
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#include <string>
#include "bf_common.hpp"

namespace bf {
namespace {
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);
AllGatherFn g_allgather = nullptr;
ErrStrFn g_errstr = nullptr;
std::mutex g_mu;

bool load_rccl(std::string& why) {
  if (g_allgather) return true;
  // 1. the RCCL the process has ALREADY loaded -- the one the caller's ncclComm_t came from (torch's bundled copy, a
  //    system copy ...): an opaque communicator must never be handed to a second instance of the library
  g_allgather = reinterpret_cast<AllGatherFn>(dlsym(RTLD_DEFAULT, "ncclAllGather"));
  if (g_allgather) {
    g_errstr = reinterpret_cast<ErrStrFn>(dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return true;
  }
  // 2. otherwise: $BAYESFILT_RCCL_LIB alone when set, else next to the HIP runtime in use, then the usual names;
  //    each candidate first with RTLD_NOLOAD (already mapped without global symbols, e.g. by a Python extension)
  std::string cand[4];
  int nc = 0;
  const char* forced = std::getenv("BAYESFILT_RCCL_LIB");
  if (forced && *forced) {
    cand[nc++] = forced;
  } else {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
      std::string p(info.dli_fname);
      const size_t slash = p.rfind('/');
      if (slash != std::string::npos) cand[nc++] = p.substr(0, slash + 1) + "librccl.so";
    }
    cand[nc++] = "librccl.so";
    cand[nc++] = "librccl.so.1";
    cand[nc++] = "/opt/rocm/lib/librccl.so";
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < nc; ++i) {
      void* h = dlopen(cand[i].c_str(), RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (!h) {
        if (pass == 1) {
          const char* de = dlerror();  // ONE call: dlerror() clears the message it returns
          why += cand[i] + ": " + (de ? de : "?") + "; ";
        }
        continue;
      }
      g_allgather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
      g_errstr = reinterpret_cast<ErrStrFn>(dlsym(h, "ncclGetErrorString"));
      if (g_allgather) return true;
      why += cand[i] + ": ncclAllGather missing; ";
    }
  }
  return false;
}
}  // namespace
}  // namespace bf

extern "C" int bf_allgather_summaries(const void* d_send, void* d_recv, size_t bytes, void* nccl_comm, void* stream) {
  using namespace bf;
  if (!d_send || !d_recv || !nccl_comm) return set_error(BF_EINVAL, "bf_allgather_summaries: NULL argument");
  if (bytes == 0) return BF_OK;
  std::lock_guard<std::mutex> lock(g_mu);
  std::string why;
  if (!load_rccl(why)) return set_error(BF_EUNSUPPORTED, "RCCL is not available: %.400s", why.c_str());
  const int rc = g_allgather(d_send, d_recv, bytes, /* ncclChar */ 0, nccl_comm, static_cast<hipStream_t>(stream));
  if (rc != 0) return set_error(BF_EHIP, "ncclAllGather failed: %s", g_errstr ? g_errstr(rc) : "?");
  return BF_OK;
}
