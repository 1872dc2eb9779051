// hawk_api.hip — the C ABI of include/hawk.h over the kernels in hawk_kernels.hip.
// Host-side responsibilities only: HBM allocation, H2D/D2H staging, launch order on one HIP
// stream, HIP-event timing, error mapping.  No compute happens on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "hawk_host.h"

static thread_local char g_hip_err[256] = "";
char* hawk_hip_err_buf() { return g_hip_err; }

// ---------------------------------------------------------------------------- caching device allocator
namespace {
struct DevPool {
  std::multimap<size_t, void*> free_;         // cached blocks by size
  std::unordered_map<void*, size_t> live_;    // blocks handed out
  size_t cached = 0;
};
DevPool g_pool[HAWK_MAX_DEVICES];
std::mutex g_pool_mu;
size_t pool_cap() {  // bytes the cache may hold per device (HAWK_POOL_MAX_GB, default 96): beyond it frees go to hipFree
  static const size_t cap = [] { const char* e = getenv("HAWK_POOL_MAX_GB"); return (size_t)((e ? atof(e) : 96.0) * (double)(1ull << 30)); }();
  return cap;
}
size_t pool_round(size_t b) { return b < (1u << 20) ? (b + 255) / 256 * 256 : (b + (1u << 20) - 1) >> 20 << 20; }
DevPool* cur_pool() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= HAWK_MAX_DEVICES) return nullptr;
  return &g_pool[d];
}
void pool_trim_locked(DevPool* P) {
  for (auto& kv : P->free_) (void)hipFree(kv.second);
  P->free_.clear();
  P->cached = 0;
}
}  // namespace

int hawk_pool_alloc(void** p, size_t bytes) {
  *p = nullptr;
  std::lock_guard<std::mutex> g(g_pool_mu);
  DevPool* P = cur_pool();
  if (!P) { snprintf(g_hip_err, sizeof(g_hip_err), "hawk_pool_alloc: no current device"); return HAWK_E_HIP; }
  const size_t want = pool_round(std::max<size_t>(bytes, 1));
  auto it = P->free_.lower_bound(want);
  if (it != P->free_.end() && it->first <= want + want / 4 + (1u << 20)) {
    *p = it->second;
    P->live_[*p] = it->first;
    P->cached -= it->first;
    P->free_.erase(it);
    return HAWK_OK;
  }
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) {  // out of memory: give the cache back and try once more
    (void)hipGetLastError();
    pool_trim_locked(P);
    e = hipMalloc(p, want);
  }
  if (e != hipSuccess) {
    snprintf(g_hip_err, sizeof(g_hip_err), "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e));
    *p = nullptr;
    return HAWK_E_HIP;
  }
  P->live_[*p] = want;
  return HAWK_OK;
}

void hawk_pool_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(g_pool_mu);
  for (int d = 0; d < HAWK_MAX_DEVICES; ++d) {  // the block's own pool (the current device may differ at teardown)
    DevPool* P = &g_pool[d];
    auto it = P->live_.find(p);
    if (it == P->live_.end()) continue;
    const size_t sz = it->second;
    P->live_.erase(it);
    if (P->cached + sz > pool_cap()) { (void)hipFree(p); return; }
    P->free_.emplace(sz, p);
    P->cached += sz;
    return;
  }
  (void)hipFree(p);  // not ours
}

void hawk_pool_trim() {
  std::lock_guard<std::mutex> g(g_pool_mu);
  if (DevPool* P = cur_pool()) pool_trim_locked(P);
}

extern "C" {

const char* hawk_strerror(int s) {
  switch (s) {
    case HAWK_OK: return "ok";
    case HAWK_E_INVALID: return "invalid argument";
    case HAWK_E_HIP: return "HIP runtime error";
    case HAWK_E_CAPACITY: return "output capacity too small";
    case HAWK_E_IUPAC: return "non-IUPAC character in sequence";
    case HAWK_E_CFD: return "non-ACGT base under a CFD table lookup";
    case HAWK_E_NODEVICE: return "no GPU device visible";
    case HAWK_E_UNSUPPORTED: return "parameter outside supported range";
    case HAWK_E_COMM: return "RCCL failure";
    case HAWK_E_OVERLAP: return "a chromosome copy carries overlapping variants";
    case HAWK_E_CLAMP: return "variant beyond the original region length";
    default: return "unknown status";
  }
}
const char* hawk_last_hip_error(void) { return g_hip_err; }

int hawk_device_count(int* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
  if (n) *n = c;
  return HAWK_OK;
}

// One context - one stream - per device and process: the caching allocator's reuse of a freed block is safe because
// everything that touches the block is ordered on that one stream (hawk_host.h).  A second hawk_init for a device hands out
// the SAME context and counts the reference; the last hawk_destroy releases it.
namespace {
hawk_ctx* g_ctx_of[HAWK_MAX_DEVICES] = {};
int g_ctx_refs[HAWK_MAX_DEVICES] = {};
std::mutex g_ctx_mu;
}  // namespace

int hawk_init(int device, hawk_ctx** out) {
  if (!out) return HAWK_E_INVALID;
  int c = 0;
  hawk_device_count(&c);
  if (c <= 0) return HAWK_E_NODEVICE;
  if (device < 0 || device >= c || device >= HAWK_MAX_DEVICES) return HAWK_E_INVALID;
  std::lock_guard<std::mutex> g(g_ctx_mu);
  if (g_ctx_of[device]) { ++g_ctx_refs[device]; *out = g_ctx_of[device]; return HAWK_OK; }
  HIPCHK(hipSetDevice(device));
  hawk_ctx* ctx = new (std::nothrow) hawk_ctx();
  if (!ctx) return HAWK_E_INVALID;
  ctx->device = device;
  HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
  HIPCHK(hipHostMalloc(&ctx->pinned, 256, hipHostMallocDefault));
  g_ctx_of[device] = ctx; g_ctx_refs[device] = 1;
  *out = ctx;
  return HAWK_OK;
}

void hawk_destroy(hawk_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    const int d = ctx->device;
    if (d >= 0 && d < HAWK_MAX_DEVICES && g_ctx_of[d] == ctx) {
      if (--g_ctx_refs[d] > 0) return;
      g_ctx_of[d] = nullptr;
    }
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  delete ctx;
}

// page-locked host memory for result buffers: a device-to-host copy into it runs at link speed instead of through the
// driver's bounce buffers (the 131 MB of a C3 group export: 2.5 instead of 6.5 ms)
int hawk_host_alloc(hawk_ctx* ctx, uint64_t bytes, void** out) {
  if (!ctx || !out) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
  return HAWK_OK;
}
void hawk_host_free(void* p) { if (p) (void)hipHostFree(p); }

void* hawk_stream(hawk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int hawk_sync(hawk_ctx* ctx) {
  if (!ctx) return HAWK_E_INVALID;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_release_cached_memory(hawk_ctx* ctx) {
  if (!ctx) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hawk_pool_trim();
  return HAWK_OK;
}

}  // extern "C"
