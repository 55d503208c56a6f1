// Device-resident copies of the small constant blocks the launchers build on the host (model structs, per-step
// covariance tables), cached by content.
//
// Every entry point of include/bayesfilt.h is documented as asynchronous.  A pageable hipMemcpyAsync from a stack
// frame followed by hipStreamSynchronize (what the launchers did before) blocks the host once per call -- twenty times
// per step for a chunked scan -- and is not graph-capturable.  Here the block is copied once into pinned staging that
// lives as long as the cache entry, uploaded stream-ordered in front of the first kernel that needs it, and found
// again by content on later calls (no copy, no allocation: capturable).  Another stream that hits the entry waits on
// the upload's event.  Entries are evicted least-recently-used with hipFree, which waits for the device (a host
// synchronisation, on a miss beyond the cap only) -- but never an entry an entry point still in progress was handed: those
// are pinned from the lookup until the entry point returns, i.e. until the kernels reading them are enqueued, after which
// hipFree's device synchronisation covers them (another thread's miss, or the 2nd / 3rd lookup of the same call, could
// otherwise free a block before its reader was launched).  With everything pinned the cap is exceeded rather than enforced.
#include <cstring>
#include <mutex>
#include <vector>
#include "bf_common.hpp"

namespace bf {

namespace {
struct ConstEntry {
  uint64_t hash;
  size_t bytes;
  int device;
  void* dev;
  void* pinned;
  hipEvent_t ready;
  hipStream_t up_stream;
  uint64_t stamp;
  int pins;  // entry points in progress that were handed this block
};
std::mutex g_mu;
std::vector<ConstEntry> g_entries;
uint64_t g_clock = 0;
size_t g_total = 0;
constexpr size_t kMaxEntries = 128;
constexpr size_t kMaxBytes = 512u << 20;
struct CallPins {
  int depth = 0;
  std::vector<void*> dev;
};
CallPins& call_pins() {
  static thread_local CallPins p;
  return p;
}
void pin(ConstEntry& e) {
  CallPins& p = call_pins();
  if (p.depth > 0) {
    ++e.pins;
    p.dev.push_back(e.dev);
  }
}

uint64_t fnv1a(const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  uint64_t h = 1469598103934665603ull;
  // 8 bytes at a time (the blocks are float arrays): same avalanche as the byte loop for this purpose
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, b + i, 8);
    h = (h ^ w) * 1099511628211ull;
  }
  for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

void drop(ConstEntry& e) {
  (void)hipFree(e.dev);  // implicit device synchronisation: no kernel still reads it
  (void)hipHostFree(e.pinned);
  (void)hipEventDestroy(e.ready);
  g_total -= e.bytes;
}
}  // namespace

void begin_call_constants() { ++call_pins().depth; }
void release_call_constants() {
  CallPins& p = call_pins();
  if (--p.depth > 0 || p.dev.empty()) return;
  std::lock_guard<std::mutex> lock(g_mu);
  for (void* d : p.dev)
    for (ConstEntry& e : g_entries)
      if (e.dev == d && e.pins > 0) {
        --e.pins;
        break;
      }
  p.dev.clear();
}

int device_constants(const void* host, size_t bytes, hipStream_t stream, const void** d_out) {
  if (!host || bytes == 0 || !d_out) return set_error(BF_EINVAL, "device_constants: bad argument");
  int device = 0;
  BF_HIP_CHECK(hipGetDevice(&device));
  const uint64_t h = fnv1a(host, bytes);
  std::lock_guard<std::mutex> lock(g_mu);
  for (ConstEntry& e : g_entries) {
    if (e.hash == h && e.bytes == bytes && e.device == device && std::memcmp(e.pinned, host, bytes) == 0) {
      e.stamp = ++g_clock;
      if (e.up_stream != stream) BF_HIP_CHECK(hipStreamWaitEvent(stream, e.ready, 0));
      pin(e);
      *d_out = e.dev;
      return BF_OK;
    }
  }
  while (!g_entries.empty() && (g_entries.size() >= kMaxEntries || g_total + bytes > kMaxBytes)) {
    size_t lru = g_entries.size();
    for (size_t i = 0; i < g_entries.size(); ++i)
      if (g_entries[i].pins == 0 && (lru == g_entries.size() || g_entries[i].stamp < g_entries[lru].stamp)) lru = i;
    if (lru == g_entries.size()) break;  // everything is in use by a call in progress
    drop(g_entries[lru]);
    g_entries.erase(g_entries.begin() + (long)lru);
  }
  ConstEntry e{h, bytes, device, nullptr, nullptr, nullptr, stream, ++g_clock, 0};
  hipError_t rc = hipMalloc(&e.dev, bytes);
  if (rc == hipSuccess) rc = hipHostMalloc(&e.pinned, bytes, hipHostMallocDefault);
  if (rc == hipSuccess) rc = hipEventCreateWithFlags(&e.ready, hipEventDisableTiming);
  if (rc == hipSuccess) {
    std::memcpy(e.pinned, host, bytes);
    rc = hipMemcpyAsync(e.dev, e.pinned, bytes, hipMemcpyHostToDevice, stream);
  }
  if (rc == hipSuccess) rc = hipEventRecord(e.ready, stream);
  if (rc != hipSuccess) {
    if (e.dev) (void)hipFree(e.dev);
    if (e.pinned) (void)hipHostFree(e.pinned);
    if (e.ready) (void)hipEventDestroy(e.ready);
    return set_error(BF_EHIP, "constant upload failed: %s", hipGetErrorString(rc));
  }
  g_total += bytes;
  g_entries.push_back(e);
  pin(g_entries.back());
  *d_out = e.dev;
  return BF_OK;
}

}  // namespace bf
