// Caching device allocator.  A fit at BASELINE config 3 needs a 40 GB factor plus several
// multi-GB work buffers; hipMalloc / first-touch page mapping of that much HBM costs ~1 s per
// fit, comparable to all of the arithmetic.  Freed blocks are kept (per device, best fit within
// 12.5 % + 1 MiB) and handed back to later requests of the same size -- the second and later
// fits of a process never call hipMalloc for their large buffers.  MELLON_AMD_NO_CACHE=1 disables.
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "mln_internal.h"

namespace {
struct Block { void* p; int dev; };
std::mutex g_mu;
std::unordered_map<void*, std::pair<size_t, int>> g_live;
std::multimap<size_t, Block> g_free;
const bool g_enabled = (std::getenv("MELLON_AMD_NO_CACHE") == nullptr);
// MELLON_AMD_POISON=1 (debugging): every block handed out is filled with 0xFF bytes first (NaN as a double, -1 as an
// integer), so that a kernel reading memory nobody wrote shows up in the results instead of depending on what the
// block held before.
const bool g_poison = (std::getenv("MELLON_AMD_POISON") != nullptr && std::atoi(std::getenv("MELLON_AMD_POISON")) != 0);

void flush_locked() {
  for (auto& kv : g_free) (void)hipFree(kv.second.p);
  g_free.clear();
}
}  // namespace

hipError_t mln_dmalloc(void** out, size_t bytes) {
  if (bytes == 0) bytes = 8;
  bytes = (bytes + 255) & ~(size_t)255;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_enabled) {
    const size_t slack = bytes + bytes / 8 + (1u << 20);
    for (auto it = g_free.lower_bound(bytes); it != g_free.end() && it->first <= slack; ++it) {
      if (it->second.dev != dev) continue;
      *out = it->second.p;
      g_live[*out] = {it->first, dev};
      const size_t got = it->first;
      g_free.erase(it);
      if (g_poison) { (void)hipMemset(*out, 0xFF, got); (void)hipDeviceSynchronize(); }
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess) {  // out of memory: drop the cache and retry once
    (void)hipGetLastError();
    flush_locked();
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess) g_live[*out] = {bytes, dev};
  if (e == hipSuccess && g_poison) { (void)hipMemset(*out, 0xFF, bytes); (void)hipDeviceSynchronize(); }
  return e;
}

// Deferred frees of the calling thread (mln_dfree_defer): a chain that runs on a side stream NEXT TO a long kernel must not
// sit out that kernel in the device-wide synchronisation of every temporary it releases.
namespace {
thread_local std::vector<void*>* t_deferred = nullptr;
}
void mln_dfree_defer(std::vector<void*>* sink) { t_deferred = sink; }

namespace {
hipError_t release_block(void* p);
}

hipError_t mln_dfree(void* p) {
  if (!p) return hipSuccess;
  if (t_deferred) { t_deferred->push_back(p); return hipSuccess; }
  (void)hipDeviceSynchronize();  // same guarantee hipFree gives: nothing in flight touches the block
  return release_block(p);
}

// The caller has synchronised the ONE stream every user of the block was enqueued on: no device-wide wait, so a call that
// runs beside another context's long kernel (k-means landmarks beside the 1-NN search) does not sit that kernel out.
hipError_t mln_dfree_synced(void* p) {
  if (!p) return hipSuccess;
  if (t_deferred) { t_deferred->push_back(p); return hipSuccess; }
  return release_block(p);
}

namespace {
hipError_t release_block(void* p) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) return hipFree(p);
  const size_t bytes = it->second.first;
  const int dev = it->second.second;
  g_live.erase(it);
  if (!g_enabled) return hipFree(p);
  g_free.insert({bytes, Block{p, dev}});
  return hipSuccess;
}
}  // namespace

// ---- page-locked host blocks (a fit's mirrors of its m-vectors and of the solver's state) -----------------------------------
// hipHostMalloc pins pages: 0.1-1 ms normally, tens of milliseconds when the kernel has to compact memory first -- per fit,
// three times.  Freed blocks are kept and handed back to requests of the same size (a process fits the same m again and again).
namespace {
std::mutex g_hmu;
std::unordered_map<void*, std::pair<size_t, int>> g_hlive;      // block -> (bytes, device current when it was pinned)
std::multimap<std::pair<size_t, int>, void*> g_hfree;
}  // namespace

hipError_t mln_hmalloc(void** out, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_hmu);
    auto it = g_hfree.find({bytes, dev});
    if (g_enabled && it != g_hfree.end()) {
      *out = it->second;
      g_hlive[*out] = {bytes, dev};
      g_hfree.erase(it);
      return hipSuccess;
    }
  }
  const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_hmu); g_hlive[*out] = {bytes, dev}; }
  return e;
}

hipError_t mln_hfree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(g_hmu);
  auto it = g_hlive.find(p);
  if (it == g_hlive.end()) return hipHostFree(p);
  const std::pair<size_t, int> key = it->second;
  g_hlive.erase(it);
  if (!g_enabled || g_hfree.size() >= 64) return hipHostFree(p);     // (a bounded pool: 64 blocks of m-vector size)
  g_hfree.insert({key, p});
  return hipSuccess;
}

void mln_dcache_flush() {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    flush_locked();
  }
  // ... and the page-locked pool: "all cached blocks back to the driver" (mln_release_cached_memory) includes pinned host pages
  std::lock_guard<std::mutex> lk(g_hmu);
  for (auto& kv : g_hfree) (void)hipHostFree(kv.second);
  g_hfree.clear();
}
