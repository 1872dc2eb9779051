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

// ---------------------------------------------------------------------------- hapset
static int hapset_create_impl(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, bool zero_planes, hawk_hapset** out, bool alloc_planes = true) {
  if (!ctx || !n_hap || !hap_len || !out) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_hapset* hs = new (std::nothrow) hawk_hapset();
  if (!hs) return HAWK_E_INVALID;
  hs->ctx = ctx;
  hs->n_hap = n_hap;
  hs->hap_len.assign(hap_len, hap_len + n_hap);
  uint32_t maxw = 0;
  hs->total_len = 0;
  for (uint32_t h = 0; h < n_hap; ++h) {
    if (hap_len[h] >= (1u << 31) - 256) { delete hs; return HAWK_E_UNSUPPORTED; }
    maxw = std::max(maxw, (hap_len[h] + 31) / 32);
    hs->total_len += hap_len[h];
  }
  hs->S = (maxw + 2 + 3) / 4 * 4;
  const size_t words = (size_t)n_hap * hs->S;
  for (int p = 0; p < HAWK_PLANES; ++p) hs->plane[p] = nullptr;
  hs->d_hap_len = nullptr; hs->d_is_ref = nullptr; hs->d_scan_start = nullptr; hs->d_scan_stop = nullptr;
  hs->d_seg_off = nullptr; hs->d_seg_rel = nullptr; hs->d_seg_gen = nullptr; hs->d_tile_meta = nullptr;
  hs->bph = (hs->S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hs->ref_startp = 0;
  hs->ref_index = -1;
  hs->has_meta = false;
  int rc = HAWK_OK;
  for (int p = 0; p < HAWK_PLANES && !rc && alloc_planes; ++p) rc = hawk_pool_alloc((void**)&hs->plane[p], words * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_hap_len, (size_t)n_hap * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_is_ref, n_hap);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_scan_start, (size_t)n_hap * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_scan_stop, (size_t)n_hap * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_seg_off, (size_t)(n_hap + 1) * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_tile_meta, (size_t)n_hap * hs->bph * sizeof(TileMeta));
  if (rc) { hawk_hapset_destroy(hs); return rc; }
  if (zero_planes && alloc_planes)
    for (int p = 0; p < HAWK_PLANES; ++p) HIPCHK(hipMemsetAsync(hs->plane[p], 0, words * 4, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_hap_len, hap_len, (size_t)n_hap * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = hs;
  return HAWK_OK;
}

int hawk_hapset_create(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, hawk_hapset** out) {
  return hapset_create_impl(ctx, n_hap, hap_len, true, out);
}

int hawk_release_cached_memory(hawk_ctx* ctx) {
  if (!ctx) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hawk_pool_trim();
  return HAWK_OK;
}

void hawk_hapset_destroy(hawk_hapset* hs) {
  if (!hs) return;
  (void)hipSetDevice(hs->ctx->device);
  (void)hipStreamSynchronize(hs->ctx->stream);
  if (!hs->vplan) for (int p = 0; p < HAWK_PLANES; ++p) hawk_pool_free(hs->plane[p]);  // a view reads its plan's REF planes
  hawk_pool_free(hs->d_hap_len); hawk_pool_free(hs->d_is_ref); hawk_pool_free(hs->d_scan_start);
  hawk_pool_free(hs->d_scan_stop); hawk_pool_free(hs->d_seg_off);
  hawk_pool_free(hs->d_seg_rel); hawk_pool_free(hs->d_seg_gen); hawk_pool_free(hs->d_tile_meta);
  DevBuf* bufs[] = {&hs->keepF, &hs->keepR, &hs->counts, &hs->offsets, &hs->totals, &hs->misc, &hs->cfd, &hs->partial,
                    &hs->sites, &hs->hits, &hs->guides, &hs->lists, &hs->ckeys, &hs->cvals, &hs->cflags, &hs->cgidx,
                    &hs->ctemp, &hs->cgoff, &hs->cgc, &hs->ccnt, &hs->cfull, &hs->ctable, &hs->cocc, &hs->cdense, &hs->cgkey, &hs->cgslot, &hs->otoff, &hs->otcode, &hs->otid, &hs->othit, &hs->refbits,
                    &hs->big, &hs->refhp, &hs->vcnt0, &hs->cs_res, &hs->cs_tbase, &hs->cs_trows, &hs->cs_itb, &hs->cs_icnt, &hs->rowsA, &hs->cm_gid};
  for (auto& b : hs->cmini) b.release();
  for (auto* b : bufs) b->release();
  for (auto& b : hs->colsA) b.release();
  for (auto& b : hs->crep) b.release();
  delete hs;
}

int hawk_hapset_stride(const hawk_hapset* hs, uint32_t* stride_words) {
  if (!hs || !stride_words) return HAWK_E_INVALID;
  *stride_words = hs->S;
  return HAWK_OK;
}

int hawk_hapset_pack_ascii(hawk_hapset* hs, const char* seqs, const uint64_t* seq_off, uint64_t* bad_index) {
  if (!hs || !seqs || !seq_off) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  hs->refbits_valid = false;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  for (uint32_t h = 0; h < hs->n_hap; ++h)
    if (seq_off[h + 1] - seq_off[h] != hs->hap_len[h]) return HAWK_E_INVALID;
  uint64_t* d_off = nullptr;
  unsigned long long* d_bad = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_off, (hs->n_hap + 1) * 8);
  TEMPCHK(tmp, &d_bad, 8);
  HIPCHK(hipMemcpyAsync(d_off, seq_off, (hs->n_hap + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
  // stage the ASCII in batches of whole haplotypes (<= 256 MiB of HBM staging)
  const uint64_t kStage = 256ull << 20;
  uint64_t maxlen = 0;
  for (auto l : hs->hap_len) maxlen = std::max<uint64_t>(maxlen, l);
  const uint64_t stage_bytes = std::max(kStage, maxlen);
  uint8_t* d_stage = nullptr;
  TEMPCHK(tmp, &d_stage, std::min<uint64_t>(stage_bytes, std::max<uint64_t>(hs->total_len, 1)));
  uint32_t h0 = 0;
  while (h0 < hs->n_hap) {
    uint32_t h1 = h0;
    uint64_t bytes = 0;
    while (h1 < hs->n_hap && (h1 == h0 || bytes + hs->hap_len[h1] <= stage_bytes)) { bytes += hs->hap_len[h1]; ++h1; }
    if (bytes) HIPCHK(hipMemcpyAsync(d_stage, seqs + seq_off[h0], bytes, hipMemcpyHostToDevice, ctx->stream));
    hawk_launch_pack(ctx->stream, d_stage, d_off, h0, h1 - h0, seq_off[h0], hs->d_hap_len, hs->S, hs->plane, d_bad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next batch
    h0 = h1;
  }
  unsigned long long bad = ~0ull;
  HIPCHK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
  if (bad != ~0ull) {
    if (bad_index) *bad_index = bad;
    return HAWK_E_IUPAC;
  }
  return HAWK_OK;
}

// Host half of set_meta: validate what the kernels will trust and build the per-tile records.
static int meta_build(uint32_t n, const std::vector<uint32_t>& hap_len, uint32_t bph, const uint8_t* is_ref, const int32_t* scan_start,
                      const int32_t* scan_stop, const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen,
                      int32_t ref_index, std::vector<TileMeta>* t0, int64_t* min_gen, int64_t* max_gen) {
  if (!is_ref || !scan_start || !scan_stop || !seg_off || !seg_rel || !seg_gen) return HAWK_E_INVALID;
  if (ref_index >= (int32_t)n) return HAWK_E_INVALID;
  for (uint32_t h = 0; h < n; ++h) {
    if (seg_off[h + 1] <= seg_off[h] || seg_rel[seg_off[h]] != 0) return HAWK_E_INVALID;
    if (scan_start[h] < 0 || scan_stop[h] > (int32_t)hap_len[h]) return HAWK_E_INVALID;
    for (uint32_t k = seg_off[h] + 1; k < seg_off[h + 1]; ++k)
      if (seg_rel[k] <= seg_rel[k - 1]) return HAWK_E_INVALID;
  }
  if (ref_index >= 0 && seg_off[ref_index + 1] - seg_off[ref_index] != 1) return HAWK_E_INVALID;
  // first segment each tile needs: the last one starting at or before the tile's first base
  t0->resize((size_t)n * bph);
  for (uint32_t h = 0; h < n; ++h) {
    const uint32_t* sb = seg_rel + seg_off[h];
    const uint32_t* se = seg_rel + seg_off[h + 1];
    const uint32_t* it = sb;
    for (uint32_t blk = 0; blk < bph; ++blk) {
      const uint32_t q0 = blk * HAWK_BLOCK * 128u;
      while (it != se && *it <= q0) ++it;  // first seg_rel > q0 (tiles ascend: one walk per row)
      TileMeta& t = (*t0)[(size_t)h * bph + blk];
      t.h = h; t.blk = blk; t.hap_len = hap_len[h];
      t.scan_start = scan_start[h]; t.scan_stop = scan_stop[h]; t.is_ref = is_ref[h] ? 1u : 0u;
      t.seg0 = (uint32_t)((it - seg_rel) - 1); t.seg_end = seg_off[h + 1];
    }
  }
  *min_gen = INT64_MAX; *max_gen = INT64_MIN;
  for (uint32_t h = 0; h < n; ++h)
    for (uint32_t k = seg_off[h]; k < seg_off[h + 1]; ++k) {
      const uint32_t end = k + 1 < seg_off[h + 1] ? seg_rel[k + 1] : hap_len[h];
      *min_gen = std::min(*min_gen, seg_gen[k]);
      *max_gen = std::max(*max_gen, seg_gen[k] + (int64_t)(end - seg_rel[k]));
    }
  return HAWK_OK;
}

int hawk_hapset_set_meta(hawk_hapset* hs, const uint8_t* is_ref, const int32_t* scan_start, const int32_t* scan_stop,
                         const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, int32_t ref_index) {
  if (!hs || hs->vplan) return HAWK_E_INVALID;  // a view takes its metadata from the plan (hawk_xplan_set_meta)
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n = hs->n_hap;
  std::vector<TileMeta> t0;
  int64_t mn, mx;
  int rc = meta_build(n, hs->hap_len, hs->bph, is_ref, scan_start, scan_stop, seg_off, seg_rel, seg_gen, ref_index, &t0, &mn, &mx);
  if (rc) return rc;
  const uint32_t nseg = seg_off[n];
  hawk_pool_free(hs->d_seg_rel); hs->d_seg_rel = nullptr;
  hawk_pool_free(hs->d_seg_gen); hs->d_seg_gen = nullptr;
  POOLCHK(&hs->d_seg_rel, (size_t)nseg * 4);
  POOLCHK(&hs->d_seg_gen, (size_t)nseg * 8);
  HIPCHK(hipMemcpyAsync(hs->d_is_ref, is_ref, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_scan_start, scan_start, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_scan_stop, scan_stop, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_off, seg_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_rel, seg_rel, (size_t)nseg * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_gen, seg_gen, (size_t)nseg * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_tile_meta, t0.data(), t0.size() * sizeof(TileMeta), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hs->ref_startp = ref_index >= 0 ? seg_gen[seg_off[ref_index]] : 0;
  hs->min_gen = mn; hs->max_gen = mx;
  hs->scan_start.assign(scan_start, scan_start + n);
  hs->scan_stop.assign(scan_stop, scan_stop + n);
  hs->ref_index = ref_index;
  hs->n_ref_rows = 0;
  for (uint32_t h = 0; h < n; ++h) hs->n_ref_rows += is_ref[h] ? 1u : 0u;
  hs->has_meta = true;
  hs->refbits_valid = false;
  ++hs->cols_gen;  // tables written under the old metadata are stale
  return HAWK_OK;
}

int hawk_hapset_set_ref_partner_range(hawk_hapset* hs, int32_t start, int32_t stop) {
  if (!hs || !hs->has_meta || hs->ref_index < 0) return HAWK_E_INVALID;
  if (start < 0 || stop > (int32_t)hs->hap_len[hs->ref_index] || stop < start) return HAWK_E_INVALID;
  hs->has_partner = true; hs->partner_start = start; hs->partner_stop = stop;
  hs->refbits_valid = false;
  ++hs->cols_gen;
  return HAWK_OK;
}

static HapSetDev make_dev(const hawk_hapset* hs);
int hawk_hapset_rows_equal(hawk_hapset* hs, uint32_t n_pairs, const uint32_t* rows_a, const uint32_t* rows_b, uint8_t* equal) {
  if (!hs || hs->vplan || (n_pairs && (!rows_a || !rows_b || !equal))) return HAWK_E_INVALID;
  if (!n_pairs) return HAWK_OK;
  for (uint32_t i = 0; i < n_pairs; ++i)
    if (rows_a[i] >= hs->n_hap || rows_b[i] >= hs->n_hap) return HAWK_E_INVALID;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  uint32_t *d_a = nullptr, *d_b = nullptr;
  uint8_t* d_e = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_a, (size_t)n_pairs * 4); TEMPCHK(tmp, &d_b, (size_t)n_pairs * 4); TEMPCHK(tmp, &d_e, n_pairs);
  HIPCHK(hipMemcpyAsync(d_a, rows_a, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_b, rows_b, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
  hawk_launch_rows_equal(ctx->stream, make_dev(hs), n_pairs, d_a, d_b, d_e);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(equal, d_e, n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_hapset_download_plane(hawk_hapset* hs, int plane, uint32_t* out_words) {
  if (!hs || plane < 0 || plane >= HAWK_PLANES || !out_words) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  HIPCHK(hipSetDevice(hs->ctx->device));
  HIPCHK(hipStreamSynchronize(hs->ctx->stream));
  HIPCHK(hipMemcpy(out_words, hs->plane[plane], (size_t)hs->n_hap * hs->S * 4, hipMemcpyDeviceToHost));
  return HAWK_OK;
}

int hawk_hapset_upload_planes(hawk_hapset* hs, const uint32_t* planes) {
  if (!hs || !planes) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  hs->refbits_valid = false;
  HIPCHK(hipSetDevice(hs->ctx->device));
  const size_t words = (size_t)hs->n_hap * hs->S;
  for (int p = 0; p < HAWK_PLANES; ++p)
    HIPCHK(hipMemcpyAsync(hs->plane[p], planes + p * words, words * 4, hipMemcpyHostToDevice, hs->ctx->stream));
  HIPCHK(hipStreamSynchronize(hs->ctx->stream));
  return HAWK_OK;
}

static HapSetDev make_dev(const hawk_hapset* hs) {
  HapSetDev d;
  d.n_hap = hs->n_hap;
  d.S = hs->S;
  for (int p = 0; p < HAWK_PLANES; ++p) d.plane[p] = hs->plane[p];
  d.hap_len = hs->d_hap_len;
  d.is_ref = hs->d_is_ref;
  d.scan_start = hs->d_scan_start;
  d.scan_stop = hs->d_scan_stop;
  d.seg_off = hs->d_seg_off;
  d.seg_rel = hs->d_seg_rel;
  d.seg_gen = hs->d_seg_gen;
  d.ref_index = hs->ref_index;
  return d;
}

static int make_scan_params(const hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t guidelen,
                            uint32_t right, bool need_v, ScanParams* sp) {
  if (pamlen == 0 || pamlen > 16) return HAWK_E_UNSUPPORTED;
  if (guidelen + pamlen > HAWK_MAX_CORE) return HAWK_E_UNSUPPORTED;
  sp->pam_fwd = pam_fwd; sp->pam_rev = pam_rev;
  sp->pamlen = (int32_t)pamlen; sp->guidelen = (int32_t)guidelen; sp->right = right ? 1 : 0;
  sp->L = (int32_t)(guidelen + pamlen);
  sp->bph = hs->bph;
  uint32_t need = 0;
  for (uint32_t i = 0; i < pamlen; ++i) {
    const uint32_t a = (uint32_t)(pam_fwd >> (4 * i)) & 15u, b = (uint32_t)(pam_rev >> (4 * i)) & 15u;
    if (a == 0 || b == 0) return HAWK_E_INVALID;  // every PAM position is an IUPAC code (pam.py:55-58)
    if (a != 15u) need |= a;
    if (b != 15u) need |= b;
  }
  if (need_v) need |= 16u;
  sp->need = need;
  sp->poF = 0; sp->poR = 0;
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- K2 raw hits
int hawk_pam_scan(hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t* hits_fwd,
                  uint32_t* hits_rev, uint64_t cap_fwd, uint64_t cap_rev, uint64_t* off_fwd, uint64_t* off_rev) {
  if (!hs || !hs->has_meta || !off_fwd || !off_rev) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, pam_fwd, pam_rev, pamlen, 0, 0, false, &sp);
  if (rc) return rc;
  const HapSetDev d = make_dev(hs);
  const size_t words = (size_t)hs->n_hap * hs->S;
  const uint64_t ncnt = (uint64_t)hs->n_hap * 2 * sp.bph;  // [strand][haplotype][tile]
  if ((rc = hs->keepF.reserve(words * 4)) || (rc = hs->keepR.reserve(words * 4)) || (rc = hs->counts.reserve(ncnt * 4)) ||
      (rc = hs->offsets.reserve((ncnt + 1) * 8)) || (rc = hs->totals.reserve(sizeof(ScanTotals))) ||
      (rc = hs->partial.reserve((ncnt / 1024 + 2) * 8)))
    return rc;
  hawk_launch_scan_raw(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(), hs->counts.as<uint32_t>());
  hawk_launch_mscan(ctx->stream, hs->counts.as<uint32_t>(), ncnt, hs->partial.as<unsigned long long>(), nullptr,
                    hs->offsets.as<uint64_t>(), hs->totals.as<ScanTotals>());
  HIPCHK(hipGetLastError());
  ScanTotals tot;
  HIPCHK(hipMemcpyAsync(&tot, hs->totals.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<uint64_t> offs(ncnt + 1);
  HIPCHK(hipMemcpyAsync(offs.data(), hs->offsets.p, ncnt * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  offs[ncnt] = tot.n_keep;
  const uint64_t nf = offs[ncnt / 2], nr = tot.n_keep - nf;
  for (uint32_t h = 0; h <= hs->n_hap; ++h) {
    off_fwd[h] = offs[(size_t)h * sp.bph];                           // h == n_hap -> offs[ncnt/2] == nf
    off_rev[h] = offs[((size_t)hs->n_hap + h) * sp.bph] - nf;        // h == n_hap -> offs[ncnt] - nf == nr
  }
  if (nf > cap_fwd || nr > cap_rev || !hits_fwd || !hits_rev) return (nf || nr) ? HAWK_E_CAPACITY : HAWK_OK;
  uint32_t *d_f = nullptr, *d_r = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_f, std::max<uint64_t>(nf, 1) * 4);
  TEMPCHK(tmp, &d_r, std::max<uint64_t>(nr, 1) * 4);
  hawk_launch_emit_hits(ctx->stream, d, sp.bph, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(),
                        hs->offsets.as<uint64_t>(), nf, d_f, d_r);
  HIPCHK(hipGetLastError());
  if (nf) HIPCHK(hipMemcpyAsync(hits_fwd, d_f, nf * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (nr) HIPCHK(hipMemcpyAsync(hits_rev, d_r, nr * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_pam_scan_time(hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t reps, float* avg_ms,
                       uint64_t* scanned_positions) {
  if (!hs || !hs->has_meta || !avg_ms || !reps) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, pam_fwd, pam_rev, pamlen, 0, 0, false, &sp);
  if (rc) return rc;
  const HapSetDev d = make_dev(hs);
  const size_t words = (size_t)hs->n_hap * hs->S;
  const uint64_t ncnt = (uint64_t)hs->n_hap * 2 * sp.bph;
  if ((rc = hs->keepF.reserve(words * 4)) || (rc = hs->keepR.reserve(words * 4)) || (rc = hs->counts.reserve(ncnt * 4))) return rc;
  hawk_launch_scan_raw(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(), hs->counts.as<uint32_t>());  // warm-up
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
  for (uint32_t r = 0; r < reps; ++r)
    hawk_launch_scan_raw(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(), hs->counts.as<uint32_t>());
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  *avg_ms = ms / reps;
  if (scanned_positions) {
    uint64_t pos = 0;
    for (uint32_t h = 0; h < hs->n_hap; ++h) pos += (uint64_t)std::max(0, hs->scan_stop[h] - hs->scan_start[h]);
    *scanned_positions = pos;
  }
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- fused search
// ---------------------------------------------------------------------------- generic GBT over supplied features (RS3)
int hawk_gbt_predict(hawk_ctx* ctx, const double* feats, uint64_t n, uint32_t n_features, const hawk_gbt_model* m, int cast_f32,
                     double* out) {
  if (!ctx || !m || !m->tree_off || !m->feature || !m->left || !m->right || !m->threshold || !m->value || !n_features ||
      (n && (!feats || !out)))
    return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  for (uint32_t t = 0; t < m->n_trees; ++t) {  // every child index inside its tree and pointing forward, every feature in range
    const int32_t lo = m->tree_off[t], hi = m->tree_off[t + 1];
    if (lo < 0 || hi <= lo || (uint32_t)hi > m->n_nodes) return HAWK_E_INVALID;
    for (int32_t k = lo; k < hi; ++k) {
      if (m->feature[k] >= (int32_t)n_features) return HAWK_E_INVALID;
      if (m->feature[k] >= 0 && (m->left[k] <= k - lo || m->right[k] <= k - lo || m->left[k] >= hi - lo || m->right[k] >= hi - lo))
        return HAWK_E_INVALID;
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nn = m->n_nodes, nt = m->n_trees;
  double *d_x = nullptr, *d_th = nullptr, *d_v = nullptr, *d_o = nullptr;
  int32_t *d_off = nullptr, *d_f = nullptr, *d_l = nullptr, *d_r = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_x, n * n_features * 8); TEMPCHK(tmp, &d_off, (nt + 1) * 4); TEMPCHK(tmp, &d_f, nn * 4); TEMPCHK(tmp, &d_l, nn * 4); TEMPCHK(tmp, &d_r, nn * 4);
  TEMPCHK(tmp, &d_th, nn * 8); TEMPCHK(tmp, &d_v, nn * 8); TEMPCHK(tmp, &d_o, n * 8);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_x, feats, n * n_features * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, m->tree_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_f, m->feature, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_l, m->left, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r, m->right, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_th, m->threshold, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_v, m->value, nn * 8, hipMemcpyHostToDevice, st));
  hawk_launch_gbt(st, d_x, n, n_features, m->n_trees, d_off, d_f, d_l, d_r, d_th, d_v, m->init, m->learning_rate, cast_f32, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return HAWK_OK;
}

}  // extern "C"
// The columnar layout of a guide table: eight separate allocations (see GuideCols in hawk_device.h for the packed layout).
int hawk_reserve_cols(DevBuf (&b)[8], uint64_t cap, GuideCols* c) {
  int rc;
  const size_t sz[8] = {cap * 4, cap * 4, cap, cap * 8, cap * 8, cap, cap * 8, cap * 8 * HAWK_PLANES};
  for (int i = 0; i < 8; ++i) if ((rc = b[i].reserve(std::max<size_t>(sz[i], 16)))) return rc;
  c->hap = b[0].as<uint32_t>(); c->pos = b[1].as<uint32_t>(); c->strand = b[2].as<uint8_t>();
  c->start = b[3].as<int64_t>(); c->stop = b[4].as<int64_t>(); c->flags = b[5].as<uint8_t>();
  c->cfdon = b[6].as<double>(); c->win = b[7].as<uint64_t>(); c->cap = cap;
  c->rows = nullptr; c->startp = 0;
  return HAWK_OK;
}
extern "C" {

#define HAWK_RETRY_TEMPLATES (-100)  // private to this file: the template rows of a cluster search outgrew their reservation
static int hawk_search_once(hawk_hapset* hs, const hawk_search_params* p, hawk_table** out, hawk_timing* timing);
int hawk_search(hawk_hapset* hs, const hawk_search_params* p, hawk_table** out, hawk_timing* timing) {
  int rc = hawk_search_once(hs, p, out, timing);
  if (rc == HAWK_RETRY_TEMPLATES) rc = hawk_search_once(hs, p, out, timing);  // now reserved for the bound: cannot recur
  return rc == HAWK_RETRY_TEMPLATES ? HAWK_E_CAPACITY : rc;
}
static int hawk_search_once(hawk_hapset* hs, const hawk_search_params* p, hawk_table** out, hawk_timing* timing) {
  if (!hs || !p || !out || !hs->has_meta) return HAWK_E_INVALID;
  if (p->score_cfdon > 2) return HAWK_E_INVALID;
  if (p->score_cfdon && (p->right || !p->cfd_mm || !p->cfd_pam || p->pamlen < 2)) return HAWK_E_INVALID;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, p->pam_fwd, p->pam_rev, p->pamlen, p->guidelen, p->right, true, &sp);
  if (rc) return rc;
  const HapSetDev d = make_dev(hs);
  ++hs->cols_gen;  // the columns are about to be rewritten: earlier tables of this set become stale
  const uint64_t ntile = (uint64_t)hs->n_hap * sp.bph;
  // A view of a plan whose cluster dictionary is usable is searched per distinct cluster (hawk_csearch.hip): the scan then runs
  // over REF's tiles + one count per cluster instance.  HAWK_VIEW_SEARCH=words keeps the per-word search (hawk_vsearch.hip).
  bool by_cluster = hs->vplan && hs->vplan->cl.built && hs->vplan->cl.usable && hs->ref_index == 0;
  if (by_cluster) { const char* e = getenv("HAWK_VIEW_SEARCH"); if (e && e[0] == 'w') by_cluster = false; }
  // what the offset scan runs over: the plane kernels' tiles, then - for a cluster search - one entry per 64 consecutive cluster
  // instances (a wave of the count / emit kernels: its rows are one contiguous stretch of the table), else the view's tiles
  const uint64_t nscan = by_cluster ? (uint64_t)sp.bph + ((uint64_t)hs->vplan->cl.n_inst + 63) / 64 : ntile;
  if ((rc = hs->counts.reserve(nscan * 4)) || (rc = hs->offsets.reserve((nscan + 1) * 8)) ||
      (rc = hs->misc.reserve(512 * 8 + 64)) ||
      (rc = hs->cfd.reserve(336 * 8)) || (rc = hs->partial.reserve((nscan / 1024 + 2) * 8)))
    return rc;
  // hand-over lists (2 KB per tile): the count pass leaves each small tile's valid survivors for the emit pass.
  // HAWK_LIST_EMIT=0 keeps the recompute-everything emit pass (A/B measurements).
  static const bool list_emit_env = [] { const char* e = getenv("HAWK_LIST_EMIT"); return !(e && e[0] == '0'); }();
  // A view of an expansion plan (hawk_xplan_view) holds no planes: its REF row runs through the plane kernels below on the
  // plan's REF planes (without hand-over lists: a handful of tiles), every other row through hawk_vsearch.hip.
  const hawk_xplan* vx = hs->vplan;
  const bool list_emit = list_emit_env;
  const uint32_t plane_tiles = vx ? sp.bph * (hs->ref_index == 0 ? 1u : 0u) : (uint32_t)ntile;  // tiles the plane kernels take
  if (vx && (hs->ref_index != 0 || hs->n_ref_rows != 1)) return HAWK_E_INVALID;                 // a plan's rows: REF first, once
  uint32_t* d_lists = nullptr;
  unsigned long long* d_big = nullptr;
  if (list_emit) {
    // a REF tile takes one work-list entry per 512 survivors (<= 128 per tile), any other big tile one
    const uint64_t n_ref_tiles = (uint64_t)sp.bph * hs->n_ref_rows;
    if ((rc = hs->lists.reserve((size_t)plane_tiles * HAWK_LIST_CAP * 4 + 16)) || (rc = hs->big.reserve(((size_t)plane_tiles + 128 * n_ref_tiles) * 8 + 16))) return rc;
    d_lists = hs->lists.as<uint32_t>();
    d_big = hs->big.as<unsigned long long>();
  }
  if (p->score_cfdon) {  // the tables go up once; later searches with the same tables find them in HBM
    if (hs->cfd_host.size() != 336 || memcmp(hs->cfd_host.data(), p->cfd_mm, 320 * 8) != 0 ||
        memcmp(hs->cfd_host.data() + 320, p->cfd_pam, 16 * 8) != 0) {
      hs->cfd_host.assign(336, 0.0);
      memcpy(hs->cfd_host.data(), p->cfd_mm, 320 * 8);
      memcpy(hs->cfd_host.data() + 320, p->cfd_pam, 16 * 8);
      HIPCHK(hipMemcpyAsync(hs->cfd.p, hs->cfd_host.data(), 336 * 8, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  HIPCHK(hipMemsetAsync(hs->misc.p, 0, 512 * 8 + 64, ctx->stream));
  unsigned long long* d_shards = hs->misc.as<unsigned long long>();          // [256][2] candidate / hit partial sums
  int* d_status = reinterpret_cast<int*>(hs->misc.as<char>() + 512 * 8);
  uint32_t* d_big_count = reinterpret_cast<uint32_t*>(hs->misc.as<char>() + 512 * 8 + 16);  // zeroed with misc
  unsigned long long* d_tcount = reinterpret_cast<unsigned long long*>(hs->misc.as<char>() + 512 * 8 + 8);  // template rows handed out (hawk_csearch.hip)
  // status (4 B) | work-list count | totals share one 64-byte block: a single copy into page-locked memory per search
  static_assert(sizeof(ScanTotals) == 32, "status block layout");
  ScanTotals* d_totals = reinterpret_cast<ScanTotals*>(hs->misc.as<char>() + 512 * 8 + 32);
  const char* d_block = hs->misc.as<char>() + 512 * 8;
  char* h_block = static_cast<char*>(ctx->pinned);
  GuideParams gp;
  gp.pamlen = sp.pamlen; gp.guidelen = sp.guidelen; gp.right = sp.right; gp.L = sp.L;
  gp.score_cfdon = (int32_t)p->score_cfdon;  // 1: a non-ACGT base under a lookup is HAWK_E_CFD; 2: it scores NaN ("NA")
  gp.cfd_mm = hs->cfd.as<double>(); gp.cfd_pam = hs->cfd.as<double>() + 320; gp.bph = sp.bph;
  RefInfo ri;
  ri.index = hs->ref_index; ri.startp = hs->ref_startp;
  for (int s = 0; s < 2; ++s) {
    ri.lo[s] = 0; ri.hi[s] = 0;
    if (hs->ref_index >= 0) {  // same arithmetic as the kernel's phase A, for the REF haplotype
      const bool pamfirst = (sp.right != 0) != (s != 0);
      const int po = pamfirst ? 0 : sp.guidelen;
      const int haplen = (int)hs->hap_len[hs->ref_index];
      // where REF has guides a haplotype row can be grouped with: REF's own scan range, or - for a tile of a larger
      // region - the region's scan range as far as this tile's REF string reaches (hawk_hapset_set_ref_partner_range)
      const int rs = hs->has_partner ? hs->partner_start : hs->scan_start[hs->ref_index];
      const int re = hs->has_partner ? hs->partner_stop : hs->scan_stop[hs->ref_index];
      ri.lo[s] = std::max(rs - po, HAWK_PAD);
      ri.hi[s] = std::min(re - po, haplen - sp.L - HAWK_PAD + 1);
    }
  }
  ri.bits[0] = ri.bits[1] = nullptr;
  ri.n_bits = 0;
  if (hs->ref_index >= 0) {
    // REF's candidate windows as bitmaps (k_ref_bits): rebuilt only when the PAM / guide geometry or REF's range changed
    const uint64_t key[6] = {p->pam_fwd, p->pam_rev, ((uint64_t)p->pamlen << 32) | p->guidelen, (uint64_t)(p->right ? 1 : 0),
                             ((uint64_t)(uint32_t)ri.lo[0] << 32) | (uint32_t)ri.hi[0], ((uint64_t)(uint32_t)ri.lo[1] << 32) | (uint32_t)ri.hi[1]};
    if ((rc = hs->refbits.reserve((size_t)hs->S * 4 * 2))) return rc;
    ri.bits[0] = hs->refbits.as<uint32_t>();
    ri.bits[1] = hs->refbits.as<uint32_t>() + hs->S;
    ri.n_bits = hs->S * 32u;
    if (vx && ((rc = hs->refhp.reserve(((size_t)hs->S + 1) * 8 * 2)) || (rc = hs->vcnt0.reserve(by_cluster ? 16 : ntile * 4)))) return rc;
    if (!hs->refbits_valid || memcmp(hs->refbits_key, key, sizeof(key)) != 0) {
      hawk_launch_ref_bits(ctx->stream, d, sp, ri, hs->refbits.as<uint32_t>(), hs->refbits.as<uint32_t>() + hs->S);
      // REF's PAM hits + prefix counts: what the clean stretches of a plan's rows are counted from
      if (vx) hawk_launch_ref_hits(ctx->stream, d, sp, hs->ref_index, hs->refhp.p);
      HIPCHK(hipGetLastError());
      memcpy(hs->refbits_key, key, sizeof(key));
      hs->refbits_valid = true;
    }
  }
  VcArgs va;
  memset(&va, 0, sizeof(va));
  if (vx) {
    for (int pl = 0; pl < 4; ++pl) va.ref[pl] = vx->ref5[pl].as<uint32_t>();
    va.ref_S = hs->S;
    va.recs_ = vx->recs.p; va.alt_codes = vx->codes.as<uint8_t>(); va.hv_off = vx->off.as<uint64_t>(); va.tiles_ = vx->tiles.p;
    va.hp = hs->refhp.as<uint4>();
  }
  const uint32_t v_tiles = vx ? (uint32_t)ntile - plane_tiles : 0u;
  ClDict cd;
  memset(&cd, 0, sizeof(cd));
  uint64_t tcap = 0, t_rows_used = 0;
  if (by_cluster) {
    const auto& cl = vx->cl;
    cd.n_inst = cl.n_inst; cd.n_uniq = cl.n_uniq;
    cd.inst_uid = cl.inst_uid.as<uint32_t>(); cd.inst_o = cl.inst_o.as<int32_t>(); cd.inst_row = cl.inst_row.as<uint32_t>();
    cd.inst_pa = cl.inst_pa.as<int32_t>(); cd.inst_rb = cl.inst_rb.as<int32_t>();
    cd.u_rec = cl.u_rec.as<uint32_t>(); cd.u_n = cl.u_n.as<uint32_t>(); cd.u_row = cl.u_row.as<uint32_t>(); cd.u_o = cl.u_o.as<int32_t>();
    cd.u_seg = cl.u_seg.as<uint32_t>();
    // template rows: packed as the search produces them.  Their number is bounded by the window starts of the distinct clusters
    // (cl.slots: 2 strands x every start), but a PAM keeps a few per cent of those: reserve 16 rows per distinct cluster, and
    // if a search needs more it produces no table (k_cs_count sees the counter), says so and is rerun with the bound reserved
    const char* e0 = getenv("HAWK_CLUSTER_ROWS0");  // tests: a first reservation small enough to overflow
    const uint64_t first = e0 ? strtoull(e0, nullptr, 10) : 16ull * cl.n_uniq + 65536;
    tcap = std::max<uint64_t>(std::min<uint64_t>(cl.slots, std::max<uint64_t>(hs->cs_tcap, first)), 1);
    if ((rc = hs->cs_res.reserve((size_t)std::max<uint32_t>(cl.n_uniq, 1) * 16)) || (rc = hs->cs_tbase.reserve((size_t)std::max<uint32_t>(cl.n_uniq, 1) * 4)) ||
        (rc = hs->cs_trows.reserve((size_t)tcap * hawk_cs_row_bytes())) || (rc = hs->cs_itb.reserve((size_t)std::max<uint32_t>(cl.n_inst, 1) * 4)) ||
        (rc = hs->cs_icnt.reserve((size_t)std::max<uint32_t>(cl.n_inst, 1) * 4)))
      return rc;
  }
  // the view's share of the two passes: per dirty word of every row, or per distinct cluster + a copy per instance
  uint32_t* const d_counts_v = hs->counts.as<uint32_t>() + plane_tiles;
  auto view_count = [&]() {
    if (by_cluster) {
      hawk_launch_cs_templates(ctx->stream, d, va, cd, sp, gp, ri, hs->cs_res.p, hs->cs_tbase.as<uint32_t>(), hs->cs_trows.p, d_tcount, tcap, d_status);
      (void)hipEventRecord(ctx->ev[8], ctx->stream);
      hawk_launch_cs_count(ctx->stream, d, va, cd, sp, hs->cs_res.p, hs->cs_tbase.as<uint32_t>(), d_tcount, tcap, d_counts_v, hs->cs_icnt.as<uint32_t>(),
                           hs->cs_itb.as<uint32_t>(), d_shards);
    } else {
      hawk_launch_vsearch(ctx->stream, 0, d, va, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), hs->vcnt0.as<uint32_t>(), d_shards, nullptr,
                          GuideCols{}, d_status, plane_tiles, v_tiles);
    }
  };
  // the emit side.  Plane kernels (all rows of a set with planes; REF's rows of a view) and the per-word search of a view write
  // columns; the cluster search writes packed rows (k_cs_emit_rows), and REF's rows - staged as columns - are packed in front of them
  const uint64_t stage_cap = by_cluster ? 2ull * hs->hap_len[hs->ref_index] + 64 : 0;  // REF keeps at most every window start of both strands
  auto emit_all = [&](const GuideCols& cols, const GuideCols& packed) {
    hawk_launch_search(ctx->stream, 1, d, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), d_shards,
                       hs->offsets.as<uint64_t>(), cols, d_status, d_lists, d_big_count, d_big, ctx->ev[5], plane_tiles);
    if (!vx) return;
    (void)hipEventRecord(ctx->ev[7], ctx->stream);
    if (by_cluster) {
      hawk_launch_rows_pack(ctx->stream, cols, hs->offsets.as<uint64_t>() + plane_tiles, 0, std::min<uint64_t>(stage_cap, packed.cap), packed.rows,
                            packed.startp, d_status);
      (void)hipEventRecord(ctx->ev[9], ctx->stream);
      hawk_launch_cs_emit_rows(ctx->stream, cd, hs->cs_icnt.as<uint32_t>(), hs->cs_itb.as<uint32_t>(), hs->cs_trows.p, hs->offsets.as<uint64_t>() + plane_tiles,
                               d_tcount, tcap, packed.rows, packed.cap, d_status);
    } else {
      hawk_launch_vsearch(ctx->stream, 1, d, va, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), hs->vcnt0.as<uint32_t>(), d_shards,
                          hs->offsets.as<uint64_t>(), cols, d_status, plane_tiles, v_tiles);
    }
  };
  // reserve for `cap` rows: `cols` is what the column emitters write, `table` what the finished table is
  auto reserve_table = [&](uint64_t cap, GuideCols* cols, GuideCols* table) -> int {
    int r;
    if (!by_cluster) {
      if ((r = hawk_reserve_cols(hs->colsA, cap, cols))) return r;
      *table = *cols;
      return HAWK_OK;
    }
    if ((r = hawk_reserve_cols(hs->colsA, stage_cap, cols)) || (r = hs->rowsA.reserve(cap * 64))) return r;
    memset(table, 0, sizeof(*table));
    table->rows = hs->rowsA.as<uint4>(); table->cap = cap; table->startp = ri.startp;
    return HAWK_OK;
  };
  uint64_t& table_cap = by_cluster ? hs->rows_cap : hs->cols_cap;
  GuideCols none = {};
  hipEvent_t* ev = ctx->ev;
  HIPCHK(hipEventRecord(ev[0], ctx->stream));
  hawk_launch_search(ctx->stream, 0, d, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), d_shards, nullptr, none, d_status, d_lists, d_big_count, d_big,
                     nullptr, plane_tiles);
  if (vx) HIPCHK(hipEventRecord(ev[6], ctx->stream));
  if (vx) view_count();
  HIPCHK(hipEventRecord(ev[1], ctx->stream));
  hawk_launch_mscan(ctx->stream, hs->counts.as<uint32_t>(), nscan, hs->partial.as<unsigned long long>(), d_shards,
                    hs->offsets.as<uint64_t>(), d_totals);
  HIPCHK(hipEventRecord(ev[2], ctx->stream));
  HIPCHK(hipGetLastError());
  ScanTotals tot;
#ifdef HAWK_ABLATION  // measurement hook of the ablation builds only (tools/ab.sh): never compiled into the product library
  static const bool count_only = [] { const char* e = getenv("HAWK_COUNT_ONLY"); return e && e[0] == '1'; }();
#else
  const bool count_only = false;
#endif
  GuideCols ca, tc;
  int status = 0;
  uint64_t nrows = 0;
  bool emitted = false;
  auto template_overflow = [&](uint64_t tc_used) {  // the rerun reserves what this search asked for (+ 1/8), at most the plan's bound
    hs->cs_tcap = std::min<uint64_t>(vx->cl.slots, tc_used + tc_used / 8 + 64);
  };
  if (table_cap && !count_only) {
    // The table of an earlier search on this set is still reserved: launch the emit pass straight behind the offset
    // scan instead of waiting for the row count to cross PCIe (the kernels take their offsets from HBM and refuse to
    // write past the capacity).  If the table turns out larger, the normal path below runs after a reserve.
    if ((rc = reserve_table(table_cap, &ca, &tc))) return rc;
    HIPCHK(hipEventRecord(ev[3], ctx->stream));
    emit_all(ca, tc);
    HIPCHK(hipEventRecord(ev[4], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_block, d_block, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(&tot, h_block + 32, sizeof(tot));
    memcpy(&status, h_block, 4);
    if (by_cluster) { uint64_t tcu; memcpy(&tcu, h_block + 8, 8); if (tcu > tcap) { template_overflow(tcu); return HAWK_RETRY_TEMPLATES; } t_rows_used = tcu; }
    nrows = tot.n_keep;
    emitted = nrows <= table_cap;
    if (!emitted) {
      // only the capacity refusal of the emit pass is answered by emitting again; any other status was raised by the count side
      // (a strict-mode CFD error, an unsupported coordinate range) and stands - the kernels keep the FIRST status they raise
      if (status && status != HAWK_E_CAPACITY) return status;
      status = 0;
      HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
    }
  } else {
    HIPCHK(hipMemcpyAsync(h_block, d_block, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(&tot, h_block + 32, sizeof(tot));
    memcpy(&status, h_block, 4);
    if (by_cluster) { uint64_t tcu; memcpy(&tcu, h_block + 8, 8); if (tcu > tcap) { template_overflow(tcu); return HAWK_RETRY_TEMPLATES; } t_rows_used = tcu; }
    if (status) return status;
    nrows = tot.n_keep;
  }
  if (count_only) {  // measurement hook: time the count pass of an experimental build whose counts the emit pass cannot use
    if (getenv("HAWK_COUNT_VERBOSE")) fprintf(stderr, "[hawk] count pass: n_keep=%llu n_cand=%llu n_hits=%llu\n", (unsigned long long)tot.n_keep, (unsigned long long)tot.n_cand, (unsigned long long)tot.n_hits);
    nrows = 0;
  }
  if (!emitted) {
    if ((rc = reserve_table(std::max<uint64_t>(std::max<uint64_t>(nrows, 1), table_cap), &ca, &tc))) return rc;
    table_cap = tc.cap;
    HIPCHK(hipEventRecord(ev[3], ctx->stream));
    if (nrows) emit_all(ca, tc);
    HIPCHK(hipEventRecord(ev[4], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_block, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(&status, h_block, 4);
  }
  if (timing) {
    memset(timing, 0, sizeof(*timing));
    (void)hipEventElapsedTime(&timing->count_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&timing->offsets_ms, ev[1], ev[2]);
    (void)hipEventElapsedTime(&timing->emit_ms, ev[3], ev[4]);
    if (nrows && d_lists) (void)hipEventElapsedTime(&timing->emit_list_ms, ev[3], ev[5]);
    (void)hipEventElapsedTime(&timing->total_ms, ev[0], ev[4]);
    if (vx) {
      (void)hipEventElapsedTime(&timing->v_count_ms, ev[6], ev[1]);
      if (nrows) (void)hipEventElapsedTime(&timing->v_emit_ms, ev[7], ev[4]);
      timing->v_path = by_cluster ? 2u : 1u;
      if (by_cluster) (void)hipEventElapsedTime(&timing->v_templates_ms, ev[6], ev[8]);
      if (by_cluster && nrows) (void)hipEventElapsedTime(&timing->v_emit_rows_ms, ev[9], ev[4]);
    }
    uint64_t pos = 0;
    for (uint32_t h = 0; h < hs->n_hap; ++h) pos += (uint64_t)std::max(0, hs->scan_stop[h] - hs->scan_start[h]);
    timing->scanned_positions = pos;
  }
  if (status) return status;
  hawk_table* t = new (std::nothrow) hawk_table();
  if (!t) return HAWK_E_INVALID;
  t->hs = hs; t->ctx = ctx; t->gen = hs->cols_gen;
  t->n_rows = nrows; t->n_cand = tot.n_cand; t->n_hits = tot.n_hits; t->cols = tc; t->cap = tc.cap;
  t->guidelen = p->guidelen; t->pamlen = p->pamlen; t->right = p->right ? 1 : 0; t->n_groups = 0; t->collapsed = false;
  t->by_cluster = by_cluster; t->plane_tiles = plane_tiles; t->t_rows = by_cluster ? t_rows_used : 0;
  *out = t;
  return HAWK_OK;
}

void hawk_table_destroy(hawk_table* t) {  // columns live in the hapset's workspace, or in own[] for a merged table
  if (!t) return;
  if (!t->hs) {
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    for (auto& b : t->own) b.release();
  }
  delete t;
}

int hawk_table_counts(const hawk_table* t, uint64_t* n_rows, uint64_t* n_candidates, uint64_t* n_hits) {
  if (!t) return HAWK_E_INVALID;
  if (n_rows) *n_rows = t->n_rows;
  if (n_candidates) *n_candidates = t->n_cand;
  if (n_hits) *n_hits = t->n_hits;
  return HAWK_OK;
}

int hawk_table_download(hawk_table* t, uint32_t* hap, uint32_t* pos, uint8_t* strand, int64_t* start, int64_t* stop,
                        uint8_t* flags, double* cfdon, uint64_t* win) {
  if (!t || hawk_table_stale(t)) return HAWK_E_INVALID;
  hawk_ctx* ctx = t->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows;
  if (!n) return HAWK_OK;
  GuideCols c = t->cols;
  PoolScope tmp;
  if (c.rows) {  // packed rows: the asked-for columns are cut out on the device first
    GuideCols u;
    memset(&u, 0, sizeof(u));
    u.cap = n;
    if (hap) TEMPCHK(tmp, &u.hap, n * 4);
    if (pos) TEMPCHK(tmp, &u.pos, n * 4);
    if (strand) TEMPCHK(tmp, &u.strand, n);
    if (start) TEMPCHK(tmp, &u.start, n * 8);
    if (stop) TEMPCHK(tmp, &u.stop, n * 8);
    if (flags) TEMPCHK(tmp, &u.flags, n);
    if (cfdon) TEMPCHK(tmp, &u.cfdon, n * 8);
    if (win) TEMPCHK(tmp, &u.win, n * 8 * HAWK_PLANES);
    hawk_launch_rows_unpack(ctx->stream, c.rows, n, c.startp, u);
    HIPCHK(hipGetLastError());
    c = u;
  }
  if (hap) HIPCHK(hipMemcpyAsync(hap, c.hap, n * 4, hipMemcpyDefault, ctx->stream));
  if (pos) HIPCHK(hipMemcpyAsync(pos, c.pos, n * 4, hipMemcpyDefault, ctx->stream));
  if (strand) HIPCHK(hipMemcpyAsync(strand, c.strand, n, hipMemcpyDefault, ctx->stream));
  if (start) HIPCHK(hipMemcpyAsync(start, c.start, n * 8, hipMemcpyDefault, ctx->stream));
  if (stop) HIPCHK(hipMemcpyAsync(stop, c.stop, n * 8, hipMemcpyDefault, ctx->stream));
  if (flags) HIPCHK(hipMemcpyAsync(flags, c.flags, n, hipMemcpyDefault, ctx->stream));
  if (cfdon) HIPCHK(hipMemcpyAsync(cfdon, c.cfdon, n * 8, hipMemcpyDefault, ctx->stream));
  if (win)
    for (int p = 0; p < HAWK_PLANES; ++p)
      HIPCHK(hipMemcpyAsync(win + (size_t)p * n, c.win + (size_t)p * c.cap, n * 8, hipMemcpyDefault, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_table_layout(const hawk_table* t, uint32_t* layout, int64_t* startp) {
  if (!t || !layout) return HAWK_E_INVALID;
  *layout = t->cols.rows ? HAWK_LAYOUT_ROWS : HAWK_LAYOUT_COLUMNS;
  if (startp) *startp = t->cols.rows ? t->cols.startp : 0;
  return HAWK_OK;
}

int hawk_table_download_rows(hawk_table* t, void* rows64) {
  if (!t || hawk_table_stale(t) || !rows64) return HAWK_E_INVALID;
  if (!t->cols.rows) return HAWK_E_UNSUPPORTED;
  HIPCHK(hipSetDevice(t->ctx->device));
  if (t->n_rows) HIPCHK(hipMemcpyAsync(rows64, t->cols.rows, t->n_rows * 64, hipMemcpyDefault, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  return HAWK_OK;
}

int hawk_table_device_rows(hawk_table* t, void** rows64, int64_t* startp) {
  if (!t || hawk_table_stale(t) || !rows64) return HAWK_E_INVALID;
  if (!t->cols.rows) return HAWK_E_UNSUPPORTED;
  *rows64 = t->cols.rows;
  if (startp) *startp = t->cols.startp;
  return HAWK_OK;
}

int hawk_table_device_columns(hawk_table* t, void** hap, void** pos, void** strand, void** start, void** stop,
                              void** flags, void** cfdon, void** win, uint64_t* win_plane_stride) {
  if (!t || hawk_table_stale(t)) return HAWK_E_INVALID;
  const GuideCols& c = t->cols;
  if (c.rows) return HAWK_E_UNSUPPORTED;  // packed rows: hawk_table_device_rows
  if (hap) *hap = c.hap;
  if (pos) *pos = c.pos;
  if (strand) *strand = c.strand;
  if (start) *start = c.start;
  if (stop) *stop = c.stop;
  if (flags) *flags = c.flags;
  if (cfdon) *cfdon = c.cfdon;
  if (win) *win = c.win;
  if (win_plane_stride) *win_plane_stride = c.cap;  // plane p of the window slices starts at win + p * stride
  return HAWK_OK;
}

static int collapse_rows(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms);
static int collapse_by_templates(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms);
int hawk_table_collapse_ex(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms) {
  if (!t || !n_groups || !t->hs || hawk_table_stale(t)) return HAWK_E_INVALID;
  if (flank_up > HAWK_PAD || flank_down > HAWK_PAD) return HAWK_E_UNSUPPORTED;
  // A table the cluster search wrote is grouped on its template rows (HAWK_COLLAPSE_TEMPLATES=0: on its own rows, as any table)
  const char* et = getenv("HAWK_COLLAPSE_TEMPLATES");
  if (t->by_cluster && t->n_rows && !(et && et[0] == '0')) return collapse_by_templates(t, flank_up, flank_down, n_groups, kernel_ms);
  return collapse_rows(t, flank_up, flank_down, n_groups, kernel_ms);
}
static int collapse_rows(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms) {
  hawk_hapset* hs = t->hs;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows;
  t->collapsed = false;
  *n_groups = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  hs->collapse_gen = t->gen;
  if (n == 0) { t->n_groups = 0; t->collapsed = true; return HAWK_OK; }
  if (n > 0xffffffffull || hs->max_gen - hs->min_gen > 0xffffffffll) return HAWK_E_UNSUPPORTED;
  unsigned end_bit = 32;
  while (end_bit < 64 && ((uint64_t)(hs->max_gen - hs->min_gen) >> (end_bit - 32)) != 0) ++end_bit;
  // low hash bits the sort leaves out: start bits + strand + hash fill whole 8-bit passes, at least 24 hash bits stay
  const unsigned pos_bits = end_bit - 31;  // start - base, strand
  const unsigned hash_bits = std::min(31u, (pos_bits + 24 + 7) / 8 * 8 - pos_bits);
  const unsigned begin_bit = 31 - hash_bits;
  const size_t temp_bytes = hawk_collapse_temp_bytes(n, begin_bit, end_bit);
  int rc;
  if ((rc = hs->ckeys.reserve(2 * n * 8)) || (rc = hs->cvals.reserve(2 * n * 4)) || (rc = hs->cflags.reserve(n * 4)) ||
      (rc = hs->cgidx.reserve(n * 4)) || (rc = hs->ctemp.reserve(temp_bytes + 16)) || (rc = hs->cgoff.reserve((n + 1) * 8)) ||
      (rc = hs->cgc.reserve(2 * n)) || (rc = hs->ccnt.reserve(32)))
    return rc;
  // Rows are grouped on (start, strand, 63 hash bits of the rest) and the grouping is then VERIFIED: every member against
  // its group's first member on the full key (k_collapse_verify).  A mismatch - two different rows under one hash, never
  // seen - sends the call through the exact path (full 64-byte keys compared neighbour by neighbour), which
  // HAWK_COLLAPSE_EXACT=1 forces from the start; HAWK_COLLAPSE_VERIFY=0 switches the check off (A/B timing only).
  const char* ex = getenv("HAWK_COLLAPSE_EXACT");
  const char* vf = getenv("HAWK_COLLAPSE_VERIFY");
  const char* wk = getenv("HAWK_COLLAPSE_WEAK_HASH");  // test knob: every row hashes alike, so the verify pass HAS to catch it
  const bool verify = !(vf && vf[0] == '0'), weak = wk && wk[0] == '1';
  bool exact = ex && ex[0] == '1';
  for (int round = 0; round < 2; ++round, exact = true) {
  if (exact && (rc = hs->cfull.reserve(hawk_collapse_full_bytes(n)))) return rc;
  // ---- grouping through a hash table (hawk_collapse.hip) when groups are expected to be far fewer than rows: the sort
  // below then only orders (group number, row).  HAWK_COLLAPSE_MODE=sort / hash overrides the choice; a table that turns
  // out too small, or an unlucky seed twice, falls through to the sort.
  bool verified_bad = false;
  {
    const char* md = getenv("HAWK_COLLAPSE_MODE");
    const bool force_hash = md && md[0] == 'h', force_sort = md && md[0] == 's';
    const uint64_t g_est = hs->last_groups ? hs->last_groups + hs->last_groups / 4 : n / 16;
    uint64_t C = 1024;
    while (C < 2 * g_est) C <<= 1;  // at most half full (with the 25 % head room of g_est)
    const bool fits = C <= (1ull << 26) && hs->max_gen - hs->min_gen < 0xffffffffll;
    const bool want = !exact && !weak && !force_sort && fits && (force_hash || (n >= (1u << 20) && (hs->last_groups == 0 || hs->last_groups * 8 <= n)));
    if (want) {
      const size_t tb = hawk_collapse_hash_temp_bytes(n, (uint32_t)C);
      if ((rc = hs->ctable.reserve(C * 16)) || (rc = hs->cocc.reserve(C * 4)) || (rc = hs->cdense.reserve(C * 4)) ||
          (rc = hs->cgkey.reserve(2 * C * 8)) || (rc = hs->cgslot.reserve(2 * C * 4)) || (rc = hs->ctemp.reserve(std::max(tb, temp_bytes) + 16)) ||
          (rc = hs->ccnt.reserve(32)))
        return rc;
      for (int attempt = 0; attempt < 2; ++attempt) {
        unsigned long long hc[4] = {0, 0, 0, 0};
        HIPCHK(hipMemsetAsync(hs->ccnt.p, 0, 32, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
        if (hawk_launch_collapse_hash1(ctx->stream, t->cols, hs->d_is_ref, n, (int)t->guidelen, (int)t->pamlen, (int)flank_up, (int)flank_down,
                                       hs->min_gen, 0x9e3779b97f4a7c15ull * (uint64_t)(attempt + 1), hs->ctemp.p, tb, hs->ctable.p, (uint32_t)C,
                                       hs->cocc.as<uint32_t>(), hs->cdense.as<uint32_t>(), hs->cgkey.as<uint64_t>(), hs->cgslot.as<uint32_t>(),
                                       hs->cflags.as<uint32_t>(), hs->ccnt.as<unsigned long long>()))
          return HAWK_E_HIP;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hc, hs->ccnt.p, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (hc[1]) { hs->last_groups = n; if (hs->plan_groups) *hs->plan_groups = n; break; }   // no room: sort now, a larger table (or the sort) next time
        if (hc[0]) continue;                         // two identities under one key: another seed
        const uint64_t ng = hc[2];
        if (hawk_launch_collapse_hash2(ctx->stream, t->cols, n, (uint32_t)ng, (int)t->guidelen, (int)t->pamlen, (int)t->right, end_bit, hs->ctemp.p, tb,
                                       hs->cgkey.as<uint64_t>(), hs->cgslot.as<uint32_t>(), (uint32_t)C, hs->cocc.as<uint32_t>(),
                                       hs->cflags.as<uint32_t>(), hs->ckeys.as<uint32_t>(), hs->cvals.as<uint32_t>(), hs->cgoff.as<uint64_t>(),
                                       hs->cgc.as<uint8_t>(), hs->cgc.as<uint8_t>() + n))
          return HAWK_E_HIP;
        if (verify) {  // slot_of_row = cflags, slot -> group number = cocc (hawk_launch_collapse_hash2's arguments above)
          if ((rc = hs->cfull.reserve(ng * 64 + 64))) return rc;
          hawk_launch_collapse_verify_rows(ctx->stream, t->cols, hs->d_is_ref, n, (uint32_t)ng, (int)t->guidelen, (int)t->pamlen, (int)flank_up,
                                           (int)flank_down, hs->cvals.as<uint32_t>() + n, hs->cflags.as<uint32_t>(), hs->cocc.as<uint32_t>(),
                                           hs->cgoff.as<uint64_t>(), hs->cfull.p, hs->ccnt.as<unsigned long long>() + 3);
        }
        HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hs->cgoff.as<uint64_t>() + ng, &n, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(&hc[3], hs->ccnt.as<unsigned long long>() + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (kernel_ms) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); *kernel_ms += ms; }
        if (hc[3]) { verified_bad = true; break; }  // two different rows under one group: the exact path decides
        hs->last_groups = ng;
        if (hs->plan_groups) *hs->plan_groups = ng;
        t->n_groups = ng; t->collapsed = true;
        *n_groups = ng;
        return HAWK_OK;
      }
    }
  }
  if (verified_bad) continue;
  unsigned long long cnt[3] = {0, 0, 0};
  for (int attempt = 0; attempt < 4; ++attempt) {  // a new seed whenever two different rows collide in the hash bits
    HIPCHK(hipMemsetAsync(hs->ccnt.p, 0, 32, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    if (hawk_launch_collapse(ctx->stream, t->cols, hs->d_is_ref, n, (int)t->guidelen, (int)t->pamlen, (int)t->right, (int)flank_up,
                             (int)flank_down, hs->min_gen, begin_bit, end_bit, 0x9e3779b97f4a7c15ull * (uint64_t)(attempt + 1), hs->ctemp.p,
                             temp_bytes, hs->ckeys.as<uint64_t>(), hs->cvals.as<uint32_t>(), hs->cflags.as<uint32_t>(),
                             hs->cgidx.as<uint32_t>(), hs->ccnt.as<unsigned long long>(), hs->cgoff.as<uint64_t>(), hs->cgc.as<uint8_t>(),
                             hs->cgc.as<uint8_t>() + n, hs->cgidx.as<uint32_t>(), exact ? hs->cfull.p : nullptr, weak && !exact))
      return HAWK_E_HIP;
    if (verify && !exact)  // group of sorted position j: exclusive scan of the head flags + its own flag - 1
      hawk_launch_collapse_verify(ctx->stream, t->cols, hs->d_is_ref, n, (int)t->guidelen, (int)t->pamlen, (int)flank_up, (int)flank_down,
                                  hs->cvals.as<uint32_t>() + n, hs->cgidx.as<uint32_t>(), hs->cflags.as<uint32_t>(), hs->cgoff.as<uint64_t>(),
                                  hs->ccnt.as<unsigned long long>() + 2);
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cnt, hs->ccnt.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); *kernel_ms += ms; }
    if (cnt[0] == cnt[1]) break;
  }
  if (cnt[0] != cnt[1]) { if (!exact) continue; return HAWK_E_UNSUPPORTED; }
  if (cnt[2]) continue;  // the verify pass found a group holding two different rows: once more, exactly
  const uint64_t ng = cnt[1];
  HIPCHK(hipMemcpyAsync(hs->cgoff.as<uint64_t>() + ng, &n, 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hs->last_groups = ng;
  if (hs->plan_groups) *hs->plan_groups = ng;
  t->n_groups = ng; t->collapsed = true;
  *n_groups = ng;
  return HAWK_OK;
  }
  return HAWK_E_UNSUPPORTED;
}

// The collapse of a table the cluster search wrote.  Every non-REF row is a copy of one of the search's template rows in all the
// grouping compares (start, stop, strand, origin, windows), so the grouping - hashing, sorting, the exact verification - runs on
// REF's rows + the template rows (C3: 2.8 x 10^5 instead of 2.8 x 10^7; a C4 tile 10^6 instead of 10^8), and the table's rows
// only inherit their template row's group number before the one sort that orders them by (group, row).  Same groups, same
// order, same members as collapse_rows on the table itself (tests/test_gpu_vsearch.py, tools/stress_views.py).
static int collapse_by_templates(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms) {
  hawk_hapset* hs = t->hs;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const hawk_xplan* vx = hs->vplan;
  if (!vx || !vx->cl.usable) return HAWK_E_INVALID;
  const uint64_t n = t->n_rows;
  if (n > 0xffffffffull) return HAWK_E_UNSUPPORTED;
  t->collapsed = false;
  *n_groups = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  // the template rows of the search: every distinct cluster's rows [tbase[u], tbase[u] + n0 + n1) of the template array (its
  // reservation may be longer: kept window starts that turned out to repeat REF); numbered densely here
  const uint32_t nu = vx->cl.n_uniq;
  PoolScope tmp;
  uint32_t* d_ucnt;
  uint64_t* d_moff;
  unsigned long long *d_partial, *d_shards;
  ScanTotals* d_tot;
  TEMPCHK(tmp, &d_ucnt, (size_t)std::max<uint32_t>(nu, 1) * 4);
  TEMPCHK(tmp, &d_moff, ((size_t)nu + 2) * 8);
  TEMPCHK(tmp, &d_partial, ((size_t)nu / 1024 + 2) * 8);
  TEMPCHK(tmp, &d_shards, 512 * 8);
  TEMPCHK(tmp, &d_tot, sizeof(ScanTotals));
  HIPCHK(hipMemsetAsync(d_shards, 0, 512 * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(d_tot, 0, sizeof(ScanTotals), ctx->stream));
  uint64_t r0 = 0;  // REF's rows come first: as many as the first cluster instance's offset says
  ScanTotals tot;
  memset(&tot, 0, sizeof(tot));
  if (nu) {
    hawk_launch_cc_ucnt(ctx->stream, hs->cs_res.p, nu, d_ucnt);
    hawk_launch_mscan(ctx->stream, d_ucnt, nu, d_partial, d_shards, d_moff, d_tot);
    HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipMemcpyAsync(&r0, hs->offsets.as<uint64_t>() + t->plane_tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  const uint64_t t_live = tot.n_keep;
  const uint64_t nm = r0 + t_live;
  if (r0 > n || nm == 0 || nm > 0xffffffffull || r0 + 0 > n) return HAWK_E_INVALID;
  int rc;
  GuideCols mini;
  if ((rc = hawk_reserve_cols(hs->cmini, nm, &mini)) || (rc = hs->cm_gid.reserve(nm * 4))) return rc;
  float ms_a = 0.f, ms_b = 0.f, ms_c = 0.f;
  HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
  hawk_launch_cc_mini(ctx->stream, t->cols, r0, hs->cs_trows.p, t_live, d_moff, hs->cs_tbase.as<uint32_t>(), nu, hs->ref_startp, mini);
  HIPCHK(hipEventRecord(ctx->ev[9], ctx->stream));
  HIPCHK(hipGetLastError());
  hawk_table tm;  // the mini table borrows the set's collapse workspace like any table of the set
  tm.hs = hs; tm.ctx = ctx; tm.n_rows = nm; tm.n_cand = tm.n_hits = 0; tm.cap = mini.cap; tm.cols = mini;
  tm.guidelen = t->guidelen; tm.pamlen = t->pamlen; tm.right = t->right; tm.n_groups = 0; tm.collapsed = false; tm.gen = t->gen;
  uint64_t G = 0;
  if ((rc = collapse_rows(&tm, flank_up, flank_down, &G, &ms_b))) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipEventElapsedTime(&ms_a, ctx->ev[8], ctx->ev[9]);
  // every mini row's group number, then every table row's; the table's rows sorted by (group, row)
  unsigned gbits = 1;
  while (gbits < 32 && (G >> gbits) != 0) ++gbits;
  const size_t tb = hawk_collapse_expand_temp_bytes(n);
  HIPCHK(hipEventRecord(ctx->ev[8], ctx->stream));
  hawk_launch_cc_gidm(ctx->stream, hs->cvals.as<uint32_t>() + nm, hs->cgoff.as<uint64_t>(), nm, G, hs->cm_gid.as<uint32_t>());
  if ((rc = hs->ckeys.reserve(2 * n * 4)) || (rc = hs->cvals.reserve(2 * n * 4)) || (rc = hs->cgoff.reserve((std::max<uint64_t>(G, nm) + 1) * 8)) ||
      (rc = hs->cgc.reserve(2 * n)) || (rc = hs->ctemp.reserve(tb + 16)) || (rc = hs->cflags.reserve(n * 4)))
    return rc;
  ClDict cd;
  memset(&cd, 0, sizeof(cd));
  cd.n_inst = vx->cl.n_inst; cd.n_uniq = vx->cl.n_uniq; cd.inst_uid = vx->cl.inst_uid.as<uint32_t>();
  hawk_launch_cs_gid(ctx->stream, cd, hs->cs_res.p, d_moff, hs->offsets.as<uint64_t>() + t->plane_tiles, r0, n,
                     hs->cm_gid.as<uint32_t>(), hs->ckeys.as<uint32_t>(), hs->cvals.as<uint32_t>());
  if (hawk_launch_collapse_expand(ctx->stream, t->cols, n, gbits, (int)t->guidelen, (int)t->pamlen, (int)t->right, hs->ctemp.p, tb, hs->ckeys.as<uint32_t>(),
                                  hs->cvals.as<uint32_t>(), hs->cgoff.as<uint64_t>(), hs->cgc.as<uint8_t>(), hs->cgc.as<uint8_t>() + n))
    return HAWK_E_HIP;
  HIPCHK(hipMemcpyAsync(hs->cgoff.as<uint64_t>() + G, &n, 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->ev[9], ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipEventElapsedTime(&ms_c, ctx->ev[8], ctx->ev[9]);
  if (kernel_ms) *kernel_ms = ms_a + ms_b + ms_c;
  hs->collapse_gen = t->gen;
  hs->last_groups = G;
  if (hs->plan_groups) *hs->plan_groups = G;
  t->n_groups = G; t->collapsed = true;
  *n_groups = G;
  return HAWK_OK;
}

int hawk_table_collapse(hawk_table* t, uint64_t* n_groups, float* kernel_ms) {
  return hawk_table_collapse_ex(t, 0, 0, n_groups, kernel_ms);
}

// the collapse results live in the set's workspace: valid for the table that was collapsed last, until the next search
static bool collapse_valid(const hawk_table* t) {
  return t && t->collapsed && t->hs && !hawk_table_stale(t) && t->hs->collapse_gen == t->gen;
}

int hawk_table_collapse_download(hawk_table* t, uint32_t* perm, uint64_t* group_off, uint8_t* gc_num, uint8_t* gc_den) {
  if (!collapse_valid(t)) return HAWK_E_INVALID;
  hawk_hapset* hs = t->hs;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows, ng = t->n_groups;
  if (n == 0) { if (group_off) group_off[0] = 0; return HAWK_OK; }
  if (perm) HIPCHK(hipMemcpyAsync(perm, hs->cvals.as<uint32_t>() + n, n * 4, hipMemcpyDefault, ctx->stream));
  if (group_off) HIPCHK(hipMemcpyAsync(group_off, hs->cgoff.p, (ng + 1) * 8, hipMemcpyDefault, ctx->stream));
  if (gc_num) HIPCHK(hipMemcpyAsync(gc_num, hs->cgc.p, ng, hipMemcpyDefault, ctx->stream));
  if (gc_den) HIPCHK(hipMemcpyAsync(gc_den, hs->cgc.as<uint8_t>() + n, ng, hipMemcpyDefault, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_table_collapse_export(hawk_table* t, uint32_t* rep_row, uint32_t* pos, uint8_t* strand, int64_t* start, int64_t* stop,
                               uint8_t* flags, double* cfdon, uint64_t* win, uint32_t* member_hap, float* kernel_ms) {
  if (!collapse_valid(t)) return HAWK_E_INVALID;
  hawk_hapset* hs = t->hs;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows, ng = t->n_groups;
  if (kernel_ms) *kernel_ms = 0.f;
  if (n == 0) return HAWK_OK;
  // the representatives' columns and the member list reuse collapse workspace that is dead by now:
  // ckeys (2n u64: the sort's key ping-pong) holds the rep columns when they fit, cflags (n u32) the members
  GuideCols rep;
  int rc = hawk_reserve_cols(hs->crep, ng, &rep);
  if (rc) return rc;
  uint32_t* d_mem = hs->cflags.as<uint32_t>();
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
  hawk_launch_collapse_export(ctx->stream, t->cols, n, ng, hs->cvals.as<uint32_t>() + n, hs->cgoff.as<uint64_t>(), rep, d_mem);
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  HIPCHK(hipGetLastError());
  hipStream_t st = ctx->stream;
  if (rep_row) HIPCHK(hipMemcpyAsync(rep_row, rep.hap, ng * 4, hipMemcpyDefault, st));
  if (pos) HIPCHK(hipMemcpyAsync(pos, rep.pos, ng * 4, hipMemcpyDefault, st));
  if (strand) HIPCHK(hipMemcpyAsync(strand, rep.strand, ng, hipMemcpyDefault, st));
  if (start) HIPCHK(hipMemcpyAsync(start, rep.start, ng * 8, hipMemcpyDefault, st));
  if (stop) HIPCHK(hipMemcpyAsync(stop, rep.stop, ng * 8, hipMemcpyDefault, st));
  if (flags) HIPCHK(hipMemcpyAsync(flags, rep.flags, ng, hipMemcpyDefault, st));
  if (cfdon) HIPCHK(hipMemcpyAsync(cfdon, rep.cfdon, ng * 8, hipMemcpyDefault, st));
  if (win)
    for (int p = 0; p < HAWK_PLANES; ++p)
      HIPCHK(hipMemcpyAsync(win + (size_t)p * ng, rep.win + (size_t)p * rep.cap, ng * 8, hipMemcpyDefault, st));
  if (member_hap) HIPCHK(hipMemcpyAsync(member_hap, d_mem, n * 4, hipMemcpyDefault, st));
  HIPCHK(hipStreamSynchronize(st));
  if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- K7 off-targets
int hawk_genome_finalize(hawk_hapset* rows) {
  if (!rows || rows->vplan) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(rows->ctx->device));
  hawk_launch_ot_onehot(rows->ctx->stream, rows->plane, (uint64_t)rows->n_hap * rows->S);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(rows->ctx->stream));
  return HAWK_OK;
}

int hawk_offtarget_scan(hawk_hapset* hs, const hawk_ot_params* p, const uint64_t* guides2, uint32_t n_guides,
                        uint32_t* out_guide, uint32_t* out_row, uint32_t* out_q, uint8_t* out_strand, uint8_t* out_mm,
                        uint64_t* out_code, uint32_t* out_nmask, uint64_t cap, uint64_t* n_out, hawk_ot_timing* timing) {
  if (!hs || !p || !hs->has_meta || !n_out || (n_guides && !guides2)) return HAWK_E_INVALID;
  if (hs->vplan) return HAWK_E_INVALID;  // a plan view holds no planes
  if (p->guidelen + p->pamlen > 32 || p->guidelen == 0) return HAWK_E_UNSUPPORTED;  // window code = 2 bits x 32
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, p->pam_fwd, p->pam_rev, p->pamlen, p->guidelen, p->right, false, &sp);
  if (rc) return rc;
  // windows are indexed by their start q: strand 0 stores the + strand as the guide reads it,
  // strand 1 the mirror image (same convention as the search, search_guides.py:538)
  sp.poF = p->right ? 0 : (int32_t)p->guidelen;
  sp.poR = p->right ? (int32_t)p->guidelen : 0;
  const HapSetDev d = make_dev(hs);
  const size_t words = (size_t)hs->n_hap * hs->S;
  const uint64_t ncnt = (uint64_t)hs->n_hap * 2 * sp.bph;
  if ((rc = hs->keepF.reserve(words * 4)) || (rc = hs->keepR.reserve(words * 4)) || (rc = hs->counts.reserve(ncnt * 4)) ||
      (rc = hs->offsets.reserve((ncnt + 1) * 8)) || (rc = hs->totals.reserve(sizeof(ScanTotals))) ||
      (rc = hs->partial.reserve((ncnt / 1024 + 2) * 8)) || (rc = hs->misc.reserve(512 * 8 + 64)) ||
      (rc = hs->guides.reserve(std::max<size_t>((size_t)n_guides * 8, 16))))
    return rc;
  // Pigeonhole seeds when they pay: enough guides to bucket, and blocks of at least two bases.  HAWK_OT_ALLPAIRS=1
  // keeps the all-pairs kernel (A/B measurements, and the parity test runs both).
  static const bool force_allpairs = [] { const char* e = getenv("HAWK_OT_ALLPAIRS"); return e && e[0] == '1'; }();
  const int G = (int)p->guidelen, nb = (int)p->max_mm + 1;
  const bool seeded = !force_allpairs && n_guides >= 64 && nb <= OT_MAX_BLOCKS && nb * 2 <= G;
  OtSeeds sd;
  memset(&sd, 0, sizeof(sd));
  // LDS variant: guides in chunks of OT_LDS_CHUNK, 4 key bases per block (nb * 8.5 KB of LDS must leave room for a few
  // workgroups per CU); HAWK_OT_SEED_GLOBAL=1 keeps the single-table global-gather kernel
  static const bool seed_global = [] { const char* e = getenv("HAWK_OT_SEED_GLOBAL"); return e && e[0] == '1'; }();
  const bool seed_lds = seeded && !seed_global && nb <= 6;
  const uint32_t chunk = seed_lds ? OT_LDS_CHUNK : n_guides;
  const uint32_t n_chunks = seeded ? (n_guides + chunk - 1) / chunk : 0;
  if (seeded) {
    sd.nb = nb;
    const int kmax = seed_lds ? 4 : 6;
    int startb = 0;
    for (int b = 0; b < nb; ++b) {
      const int len = G / nb + (b < G % nb ? 1 : 0), kl = std::min(len, kmax);
      sd.start[b] = startb; sd.klen[b] = kl;
      for (int t = 0; t < kl; ++t) sd.pmask2[b] |= 1ull << (2 * (startb + t));
      startb += len;
    }
    // tables per (chunk, block): bucket offsets (inside the chunk), codes and guide ids in bucket order
    std::vector<uint32_t> goff;
    std::vector<uint64_t> gcode((size_t)n_chunks * nb * chunk, 0);
    std::vector<uint32_t> gid((size_t)n_chunks * nb * chunk, 0);
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const uint32_t g0 = c * chunk, ng = std::min<uint32_t>(chunk, n_guides - g0);
      for (int b = 0; b < nb; ++b) {
        const uint32_t nkeys = seed_lds ? OT_LDS_KEYS : (1u << (2 * sd.klen[b])), kmask = (1u << (2 * sd.klen[b])) - 1u;
        if (c == 0) sd.off_base[b] = (uint32_t)goff.size();  // global variant: one chunk, per-block table sizes differ
        std::vector<uint32_t> cnt(nkeys + 1, 0);
        for (uint32_t g = 0; g < ng; ++g) ++cnt[((uint32_t)(guides2[g0 + g] >> (2 * sd.start[b])) & kmask) + 1];
        for (uint32_t v = 0; v < nkeys; ++v) cnt[v + 1] += cnt[v];
        goff.insert(goff.end(), cnt.begin(), cnt.end());
        std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
        const size_t base = ((size_t)c * nb + b) * chunk;
        for (uint32_t g = 0; g < ng; ++g) {  // counting sort: guides of one bucket stay in input order
          const uint32_t slot = cur[(uint32_t)(guides2[g0 + g] >> (2 * sd.start[b])) & kmask]++;
          gcode[base + slot] = guides2[g0 + g];
          gid[base + slot] = g0 + g;
        }
      }
    }
    if ((rc = hs->otoff.reserve(goff.size() * 4)) || (rc = hs->otcode.reserve(gcode.size() * 8)) || (rc = hs->otid.reserve(gid.size() * 4)))
      return rc;
    HIPCHK(hipMemcpyAsync(hs->otoff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otcode.p, gcode.data(), gcode.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(hs->otid.p, gid.data(), gid.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope
  }
  hipEvent_t* ev = ctx->ev;
  if (n_guides) HIPCHK(hipMemcpyAsync(hs->guides.p, guides2, (size_t)n_guides * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(hs->misc.p, 0, 64, ctx->stream));
  HIPCHK(hipEventRecord(ev[0], ctx->stream));
  hawk_launch_scan_raw(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(), hs->counts.as<uint32_t>());
  hawk_launch_mscan(ctx->stream, hs->counts.as<uint32_t>(), ncnt, hs->partial.as<unsigned long long>(), nullptr,
                    hs->offsets.as<uint64_t>(), hs->totals.as<ScanTotals>());
  HIPCHK(hipEventRecord(ev[1], ctx->stream));
  HIPCHK(hipGetLastError());
  ScanTotals tot;
  HIPCHK(hipMemcpyAsync(&tot, hs->totals.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const uint64_t nsites = tot.n_keep;
  if ((rc = hs->sites.reserve(std::max<uint64_t>(nsites, 1) * sizeof(OtSite))) ||
      (rc = hs->hits.reserve(std::max<uint64_t>(cap, 1) * sizeof(OtHit))))
    return rc;
  unsigned long long* d_nhits = hs->misc.as<unsigned long long>();
  HIPCHK(hipEventRecord(ev[2], ctx->stream));
  if (nsites) hawk_launch_ot_sites(ctx->stream, d, sp, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(),
                                   hs->offsets.as<uint64_t>(), hs->sites.as<OtSite>());
  HIPCHK(hipEventRecord(ev[3], ctx->stream));
  if (seeded) {
    if (seed_lds)
      hawk_launch_ot_match_seeded_lds(ctx->stream, hs->sites.as<OtSite>(), nsites, sd, hs->otoff.as<uint32_t>(),
                                      hs->otcode.as<uint64_t>(), hs->otid.as<uint32_t>(), n_guides, n_chunks, G,
                                      p->right ? (int)p->pamlen : 0, (int)p->max_mm, hs->hits.as<OtHit>(), cap, d_nhits);
    else
      hawk_launch_ot_match_seeded(ctx->stream, hs->sites.as<OtSite>(), nsites, sd, hs->otoff.as<uint32_t>(), hs->otcode.as<uint64_t>(),
                                  hs->otid.as<uint32_t>(), n_guides, G, p->right ? (int)p->pamlen : 0, (int)p->max_mm,
                                  hs->hits.as<OtHit>(), cap, d_nhits);
  } else {
    hawk_launch_ot_match(ctx->stream, hs->sites.as<OtSite>(), nsites, hs->guides.as<uint64_t>(), n_guides, (int)p->guidelen,
                         p->right ? (int)p->pamlen : 0, (int)p->max_mm, hs->hits.as<OtHit>(), cap, d_nhits);
  }
  HIPCHK(hipEventRecord(ev[4], ctx->stream));
  HIPCHK(hipGetLastError());
  unsigned long long nh = 0;
  HIPCHK(hipMemcpyAsync(&nh, d_nhits, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n_out = nh;
  if (timing) {
    memset(timing, 0, sizeof(*timing));
    (void)hipEventElapsedTime(&timing->scan_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&timing->sites_ms, ev[2], ev[3]);
    (void)hipEventElapsedTime(&timing->match_ms, ev[3], ev[4]);
    (void)hipEventElapsedTime(&timing->total_ms, ev[0], ev[4]);
    timing->n_sites = nsites;
    uint64_t pos = 0;
    for (uint32_t h = 0; h < hs->n_hap; ++h) pos += (uint64_t)std::max(0, hs->scan_stop[h] - hs->scan_start[h]);
    timing->scanned_positions = pos;
  }
  if (nh > cap) return HAWK_E_CAPACITY;
  if (!nh) return HAWK_OK;
  std::vector<OtHit> hh(nh);
  HIPCHK(hipMemcpy(hh.data(), hs->hits.p, nh * sizeof(OtHit), hipMemcpyDeviceToHost));
  // the sites of the hits: gathered into a compact array on the device, one download
  std::vector<OtSite> ss(nh);
  if ((rc = hs->othit.reserve(nh * sizeof(OtSite)))) return rc;
  hawk_launch_ot_gather(ctx->stream, hs->sites.as<OtSite>(), hs->hits.as<OtHit>(), nh, hs->othit.as<OtSite>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ss.data(), hs->othit.p, nh * sizeof(OtSite), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (uint64_t i = 0; i < nh; ++i) {
    if (out_guide) out_guide[i] = hh[i].guide;
    if (out_row) out_row[i] = ss[i].row;
    if (out_q) out_q[i] = ss[i].q & 0x7fffffffu;
    if (out_strand) out_strand[i] = (uint8_t)(ss[i].q >> 31);
    if (out_mm) out_mm[i] = (uint8_t)hh[i].mm;
    if (out_code) out_code[i] = ss[i].code;
    if (out_nmask) out_nmask[i] = ss[i].nmask;
  }
  return HAWK_OK;
}

int hawk_cfd(hawk_ctx* ctx, const char* wt, const char* sg, uint32_t len, const char* pam2, uint64_t n,
             const double* cfd_mm, const double* cfd_pam, double* out) {
  if (!ctx || !cfd_mm || !cfd_pam || (n && (!wt || !sg || !pam2 || !out)) || len == 0) return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  char *d_wt = nullptr, *d_sg = nullptr, *d_p = nullptr;
  double *d_tab = nullptr, *d_out = nullptr;
  int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_wt, n * len); TEMPCHK(tmp, &d_sg, n * len); TEMPCHK(tmp, &d_p, n * 2);
  TEMPCHK(tmp, &d_tab, 336 * 8); TEMPCHK(tmp, &d_out, n * 8); TEMPCHK(tmp, &d_status, 4);
  HIPCHK(hipMemcpyAsync(d_wt, wt, n * len, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_sg, sg, n * len, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_p, pam2, n * 2, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_tab, cfd_mm, 320 * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_tab + 320, cfd_pam, 16 * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
  hawk_launch_cfd(ctx->stream, d_wt, d_sg, len, d_p, n, d_tab, d_tab + 320, d_out, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return status;
}

// ---------------------------------------------------------------------------- f1 haplotype expansion
// An expansion plan keeps everything hawk_hapset_expand needs in HBM - the variant table, the carried-variant lists, the
// per-workgroup variant ranges and (after hawk_xplan_set_meta) the metadata of the rows it produces - so that running
// it is device work only: the per-tile loop of a whole-contig search re-expands its tiles without touching the host.
void hawk_xplan_destroy(hawk_xplan* x) {
  if (!x) return;
  (void)hipSetDevice(x->ctx->device);
  (void)hipStreamSynchronize(x->ctx->stream);
  for (auto& p : x->ref_plane) hawk_pool_free(p);
  for (auto& b : x->ref5) b.release();
  DevBuf* bufs[] = {&x->recs, &x->tiles, &x->codes, &x->off, &x->hlen, &x->hash,
                    &x->m_is_ref, &x->m_ss, &x->m_se, &x->m_seg_off, &x->m_seg_rel, &x->m_seg_gen, &x->m_tile,
                    &x->cl.inst_uid, &x->cl.inst_o, &x->cl.inst_row, &x->cl.inst_pa, &x->cl.inst_rb, &x->cl.u_rec, &x->cl.u_n, &x->cl.u_row,
                    &x->cl.u_o, &x->cl.u_seg};
  for (auto* b : bufs) b->release();
  delete x;
}

// The device half of plan creation: copies of REF's planes, the variant table, one record per carried variant, the tile
// index.  The carried-variant lists come from the host (hv_idx / hv_o, uploaded into temporaries) or are already in HBM
// (d_idx / d_o: the genotype inversion left them there, hawk_xplan_create_gt).
static int xplan_build(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span, const uint32_t* v_alt_off,
                       const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len, uint32_t n_hap, const uint64_t* hv_off,
                       const uint32_t* hv_idx, const int32_t* hv_o, const uint32_t* d_idx, const int32_t* d_o, const uint32_t* hap_len,
                       uint32_t maxlen, hawk_xplan** out) {
  hawk_ctx* ctx = ref_set->ctx;
  const uint32_t ref_len = ref_set->hap_len[0];
  const uint64_t ncar = hv_off[n_hap];
  HIPCHK(hipSetDevice(ctx->device));
  hawk_xplan* x = new (std::nothrow) hawk_xplan();
  if (!x) return HAWK_E_INVALID;
  x->ctx = ctx; x->n_var = n_var; x->n_hap = n_hap; x->ref_len = ref_len; x->ncar = ncar;
  x->hap_len.assign(hap_len, hap_len + n_hap);
  x->has_meta = false; x->nseg = 0; x->ref_index = -1; x->ref_startp = 0; x->min_gen = 0; x->max_gen = 0;
  for (auto& p : x->ref_plane) p = nullptr;
  x->ref_S = ref_set->S;
  x->S = ((maxlen + 31) / 32 + 2 + 3) / 4 * 4;  // the stride hapset_create_impl will choose
  x->bph = (x->S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;
  const size_t nwg = (size_t)n_hap * hawk_hx_tiles_per_row(x->S);
  const size_t nv = std::max<size_t>(n_var, 1), nc = std::max<size_t>(ncar, 1);
  int rc = HAWK_OK;
  for (int p = 0; p < 4 && !rc; ++p) rc = hawk_pool_alloc((void**)&x->ref_plane[p], (size_t)x->ref_S * 4);
  // the per-variant and per-carried-variant tables only feed the record / tile kernels: temporaries of this call
  DevBuf t_r0, t_span, t_ao, t_al, t_am, t_idx, t_o;
  DevBuf* temps[] = {&t_r0, &t_span, &t_ao, &t_al, &t_am, &t_idx, &t_o};
  if (!rc) rc = t_r0.reserve(nv * 4);
  if (!rc) rc = t_span.reserve(nv * 4);
  if (!rc) rc = t_ao.reserve(nv * 4);
  if (!rc) rc = t_al.reserve(nv * 4);
  if (!rc) rc = t_am.reserve(nv * 16);
  if (!rc && !d_idx) rc = t_idx.reserve(nc * 4);
  if (!rc && !d_idx) rc = t_o.reserve(nc * 4);
  if (!rc) rc = x->recs.reserve(nc * hawk_hx_record_bytes());
  if (!rc) rc = x->tiles.reserve(nwg * hawk_hx_tile_bytes());
  if (!rc) rc = x->codes.reserve(std::max<size_t>(alt_codes_len, 1));
  if (!rc) rc = x->off.reserve((size_t)(n_hap + 1) * 8);
  if (!rc) rc = x->hlen.reserve((size_t)n_hap * 4);
  if (!rc) rc = x->hash.reserve((size_t)n_hap * 16);
  for (int p = 0; p < HAWK_PLANES && !rc; ++p) rc = x->ref5[p].reserve((size_t)x->S * 4);
  if (rc) { for (auto* b : temps) b->release(); hawk_xplan_destroy(x); return rc; }
  hipStream_t st = ctx->stream;
  hipError_t e = hipSuccess;
  for (int p = 0; p < HAWK_PLANES && e == hipSuccess; ++p) {  // REF at the rows' stride (S >= ref_S iff no row is shorter ... either way: copy what fits)
    e = hipMemsetAsync(x->ref5[p].p, 0, (size_t)x->S * 4, st);
    if (p < 4 && e == hipSuccess)
      e = hipMemcpyAsync(x->ref5[p].p, ref_set->plane[p], (size_t)std::min(x->S, x->ref_S) * 4, hipMemcpyDeviceToDevice, st);
  }
  // every variant's first 32 alt bases as plane bits (A, C, G, T): the build kernel shifts them into place instead of
  // walking the allele text (which only insertions longer than a word still need)
  std::vector<uint32_t> am(nv * 4, 0);
  for (uint32_t i = 0; i < n_var; ++i)
    for (uint32_t j = 0; j < v_alt_len[i] && j < 32; ++j) {
      const uint8_t c = alt_codes[v_alt_off[i] + j];
      for (int pl = 0; pl < 4; ++pl) am[(size_t)i * 4 + pl] |= (uint32_t)((c >> pl) & 1u) << j;
    }
  for (int p = 0; p < 4 && e == hipSuccess; ++p)
    e = hipMemcpyAsync(x->ref_plane[p], ref_set->plane[p], (size_t)x->ref_S * 4, hipMemcpyDeviceToDevice, st);
  if (n_var && e == hipSuccess) {
    e = hipMemcpyAsync(t_r0.p, v_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t_span.p, v_span, (size_t)n_var * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t_ao.p, v_alt_off, (size_t)n_var * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t_al.p, v_alt_len, (size_t)n_var * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(x->codes.p, alt_codes, alt_codes_len, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t_am.p, am.data(), (size_t)n_var * 16, hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(x->off.p, hv_off, (size_t)(n_hap + 1) * 8, hipMemcpyHostToDevice, st);
  if (ncar && e == hipSuccess && !d_idx) {
    e = hipMemcpyAsync(t_idx.p, hv_idx, ncar * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(t_o.p, hv_o, ncar * 4, hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(x->hlen.p, hap_len, (size_t)n_hap * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hawk_launch_hx_prepare(st, x->off.as<uint64_t>(), d_idx ? d_idx : t_idx.as<uint32_t>(), d_o ? d_o : t_o.as<int32_t>(), ncar, t_r0.as<uint32_t>(),
                           t_span.as<uint32_t>(), t_ao.as<uint32_t>(), t_al.as<uint32_t>(), t_am.p, x->hlen.as<uint32_t>(), n_hap, x->S,
                           x->recs.p, x->tiles.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  for (auto* b : temps) b->release();
  if (e != hipSuccess) {
    snprintf(g_hip_err, sizeof(g_hip_err), "hawk_xplan_create: %s", hipGetErrorString(e));
    hawk_xplan_destroy(x);
    return HAWK_E_HIP;
  }
  *out = x;
  return HAWK_OK;
}

int hawk_xplan_create(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                      const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len,
                      uint32_t n_hap, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o,
                      const uint32_t* hap_len, hawk_xplan** out) {
  if (!ref_set || !out || !n_hap || !hv_off || !hap_len || (n_var && (!v_r0 || !v_span || !v_alt_off || !v_alt_len || !alt_codes)))
    return HAWK_E_INVALID;
  const uint32_t ref_len = ref_set->hap_len[0];
  const uint64_t ncar = hv_off[n_hap];
  if (ncar && (!hv_idx || !hv_o)) return HAWK_E_INVALID;
  // validate everything the kernel will index with, on the host
  for (uint32_t i = 0; i < n_var; ++i) {
    if ((uint64_t)v_r0[i] + v_span[i] > ref_len || v_span[i] == 0 || v_alt_len[i] == 0) return HAWK_E_INVALID;
    if ((uint64_t)v_alt_off[i] + v_alt_len[i] > alt_codes_len) return HAWK_E_INVALID;
    if (i && v_r0[i] < v_r0[i - 1]) return HAWK_E_INVALID;  // sorted by position (alleles of one site may share it)
  }
  uint32_t maxlen = 0;
  for (uint32_t h = 0; h < n_hap; ++h) {
    if (hv_off[h + 1] < hv_off[h]) return HAWK_E_INVALID;
    int64_t off = 0;
    uint32_t prev = 0;
    for (uint64_t k = hv_off[h]; k < hv_off[h + 1]; ++k) {
      const uint32_t vi = hv_idx[k];
      if (vi >= n_var || (k > hv_off[h] && (vi <= prev || v_r0[vi] < v_r0[prev] + v_span[prev]))) return HAWK_E_INVALID;  // ascending, non-overlapping within a row
      if ((int64_t)hv_o[k] != (int64_t)v_r0[vi] + off) return HAWK_E_INVALID;  // exclusive prefix of the length changes
      off += (int64_t)v_alt_len[vi] - (int64_t)v_span[vi];
      prev = vi;
    }
    if ((int64_t)hap_len[h] != (int64_t)ref_len + off) return HAWK_E_INVALID;
    if (hap_len[h] >= (1u << 31) - 256) return HAWK_E_UNSUPPORTED;
    maxlen = std::max(maxlen, hap_len[h]);
  }
  return xplan_build(ref_set, n_var, v_r0, v_span, v_alt_off, v_alt_len, alt_codes, alt_codes_len, n_hap, hv_off, hv_idx, hv_o, nullptr, nullptr,
                     hap_len, maxlen, out);
}

int hawk_xplan_set_meta(hawk_xplan* x, const uint8_t* is_ref, const int32_t* scan_start, const int32_t* scan_stop,
                        const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, int32_t ref_index) {
  if (!x) return HAWK_E_INVALID;
  hawk_ctx* ctx = x->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n = x->n_hap;
  std::vector<TileMeta> t0;
  int64_t mn, mx;
  int rc = meta_build(n, x->hap_len, x->bph, is_ref, scan_start, scan_stop, seg_off, seg_rel, seg_gen, ref_index, &t0, &mn, &mx);
  if (rc) return rc;
  const uint32_t nseg = seg_off[n];
  if ((rc = x->m_is_ref.reserve(n)) || (rc = x->m_ss.reserve((size_t)n * 4)) || (rc = x->m_se.reserve((size_t)n * 4)) ||
      (rc = x->m_seg_off.reserve((size_t)(n + 1) * 4)) || (rc = x->m_seg_rel.reserve((size_t)nseg * 4)) ||
      (rc = x->m_seg_gen.reserve((size_t)nseg * 8)) || (rc = x->m_tile.reserve(t0.size() * sizeof(TileMeta))))
    return rc;
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(x->m_is_ref.p, is_ref, n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_ss.p, scan_start, (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_se.p, scan_stop, (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_seg_off.p, seg_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_seg_rel.p, seg_rel, (size_t)nseg * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_seg_gen.p, seg_gen, (size_t)nseg * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_tile.p, t0.data(), t0.size() * sizeof(TileMeta), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  x->nseg = nseg; x->ref_index = ref_index; x->ref_startp = ref_index >= 0 ? seg_gen[seg_off[ref_index]] : 0;
  x->min_gen = mn; x->max_gen = mx;
  x->scan_start.assign(scan_start, scan_start + n);
  x->scan_stop.assign(scan_stop, scan_stop + n);
  x->n_ref_rows = 0;
  for (uint32_t h = 0; h < n; ++h) x->n_ref_rows += is_ref[h] ? 1u : 0u;
  x->has_meta = true; x->cl.built = false; x->cl.usable = false;
  return HAWK_OK;
}

int hawk_xplan_set_ref_partner_range(hawk_xplan* x, int32_t start, int32_t stop) {
  if (!x || !x->has_meta || x->ref_index < 0) return HAWK_E_INVALID;
  if (start < 0 || stop > (int32_t)x->hap_len[x->ref_index] || stop < start) return HAWK_E_INVALID;
  x->has_partner = true; x->partner_start = start; x->partner_stop = stop;
  return HAWK_OK;
}

// the rows' metadata of a plan into a set of its rows (hawk_xplan_run's, or a view), device to device
static int xplan_install(const hawk_xplan* x, hawk_hapset* hs) {
  if (!x->has_meta || hs->n_hap != x->n_hap || hs->S != x->S) return HAWK_E_INVALID;
  hipStream_t st = x->ctx->stream;
  const uint32_t n = x->n_hap;
  hawk_pool_free(hs->d_seg_rel); hs->d_seg_rel = nullptr;
  hawk_pool_free(hs->d_seg_gen); hs->d_seg_gen = nullptr;
  int rc = hawk_pool_alloc((void**)&hs->d_seg_rel, (size_t)x->nseg * 4);
  if (!rc) rc = hawk_pool_alloc((void**)&hs->d_seg_gen, (size_t)x->nseg * 8);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(hs->d_is_ref, x->m_is_ref.p, n, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_scan_start, x->m_ss.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_scan_stop, x->m_se.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_seg_off, x->m_seg_off.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_seg_rel, x->m_seg_rel.p, (size_t)x->nseg * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_seg_gen, x->m_seg_gen.p, (size_t)x->nseg * 8, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(hs->d_tile_meta, x->m_tile.p, (size_t)n * x->bph * sizeof(TileMeta), hipMemcpyDeviceToDevice, st));
  hs->ref_startp = x->ref_startp; hs->min_gen = x->min_gen; hs->max_gen = x->max_gen;
  hs->scan_start = x->scan_start; hs->scan_stop = x->scan_stop;
  hs->ref_index = x->ref_index; hs->has_meta = true; hs->n_ref_rows = x->n_ref_rows;
  hs->plan_groups = x->groups; hs->last_groups = *x->groups;
  hs->has_partner = x->has_partner; hs->partner_start = x->partner_start; hs->partner_stop = x->partner_stop;
  hs->refbits_valid = false;
  ++hs->cols_gen;
  return HAWK_OK;
}

int hawk_xplan_run(hawk_xplan* x, hawk_hapset** out, uint64_t* hash_out, float* kernel_ms) {
  if (!x || !out) return HAWK_E_INVALID;
  hawk_ctx* ctx = x->ctx;
  hawk_hapset* hs = nullptr;
  int rc = hapset_create_impl(ctx, x->n_hap, x->hap_len.data(), false, &hs);  // the build kernel writes every word of every row
  if (rc) return rc;
  if (hs->S != x->S) { hawk_hapset_destroy(hs); return HAWK_E_INVALID; }
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemsetAsync(x->hash.p, 0, (size_t)x->n_hap * 16, st);
  if (e == hipSuccess) e = hipEventRecord(ctx->ev[0], st);
  if (e == hipSuccess) {
    hawk_launch_hx_build(st, x->ref_plane, x->ref_S, x->recs.p, x->codes.as<uint8_t>(), x->off.as<uint64_t>(), hs->d_hap_len, x->n_hap,
                         hs->S, hs->plane, x->tiles.p);
    if (hash_out) hawk_launch_hx_hash(st, hs->plane, x->n_hap, hs->S, x->hash.as<unsigned long long>());
    e = hipEventRecord(ctx->ev[1], st);
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess && hash_out) e = hipMemcpyAsync(hash_out, x->hash.p, (size_t)x->n_hap * 16, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && x->has_meta) {  // install the rows' metadata, device to device
    rc = xplan_install(x, hs);
    if (rc) { hawk_hapset_destroy(hs); return rc; }
  }
  if (e == hipSuccess && (hash_out || kernel_ms)) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    snprintf(g_hip_err, sizeof(g_hip_err), "hawk_xplan_run: %s", hipGetErrorString(e));
    hawk_hapset_destroy(hs);
    return HAWK_E_HIP;
  }
  if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
  *out = hs;
  return HAWK_OK;
}

// The cluster dictionary of a plan (hawk_csearch.hip): which rows carry which distinct variant cluster.  Built once per plan,
// from the records and the rows' scan bounds; a search of a view then does the per-window work once per distinct cluster.
// Not usable (the per-word search of hawk_vsearch.hip takes the plan instead) when a chain of variants is longer than the
// builder accepts, when two different clusters share a hash, or when sharing is too thin to pay for the template rows.
static int xplan_build_dict(hawk_xplan* x) {
  auto& cl = x->cl;
  if (cl.built) return HAWK_OK;
  cl.built = true; cl.usable = false; cl.status = 0; cl.n_inst = cl.n_uniq = 0; cl.slots = 0; cl.build_ms = 0.f;
  hawk_ctx* ctx = x->ctx;
  hipStream_t st = ctx->stream;
  const uint32_t n = x->n_hap;
  if (x->ncar == 0 || x->ncar >= (1ull << 32) - 2 || n < 2) { cl.status = 4; return HAWK_OK; }
  PoolScope tmp;
  uint32_t *d_cnt, *d_off, *d_status;
  TEMPCHK(tmp, &d_cnt, (size_t)n * 4);
  TEMPCHK(tmp, &d_off, (size_t)(n + 1) * 4);
  TEMPCHK(tmp, &d_status, 64);
  HIPCHK(hipMemsetAsync(d_status, 0, 64, st));
  HIPCHK(hipEventRecord(ctx->ev[8], st));
  hawk_launch_cl_count(st, x->recs.p, x->off.as<uint64_t>(), x->m_is_ref.as<uint8_t>(), x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), n, d_cnt);
  hawk_launch_scan_u32(st, d_cnt, n, d_off);
  uint32_t n_inst = 0, n_head = 0;  // n_head: the instances of the first 48 rows (hawk_launch_cl_insert)
  HIPCHK(hipMemcpyAsync(&n_inst, d_off + n, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&n_head, d_off + std::min<uint32_t>(n, 49), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (n_inst == 0) { cl.status = 4; return HAWK_OK; }
  int rc;
  if ((rc = cl.inst_uid.reserve((size_t)n_inst * 4)) || (rc = cl.inst_o.reserve((size_t)n_inst * 4)) || (rc = cl.inst_row.reserve((size_t)n_inst * 4)) ||
      (rc = cl.inst_pa.reserve((size_t)n_inst * 4)) || (rc = cl.inst_rb.reserve((size_t)n_inst * 4)))
    return rc;
  // built in (row, position) order, then laid out stretch by stretch of REF (k_cl_permute)
  // HAWK_CLUSTER_ORDER=stretch lays the instances out stretch by stretch of REF (32 kb each: the template rows a search copies
  // then stay in L2) at the price of a table that is no longer haplotype-major; the default keeps (row, position) order
  uint32_t bshift = 31;
  { const char* eo = getenv("HAWK_CLUSTER_ORDER"); if (eo && eo[0] == 's') { bshift = 15; while (((x->ref_len >> bshift) + 1) > 1024) ++bshift; } }
  const uint32_t n_bkt = (x->ref_len >> bshift) + 1;
  uint32_t *t_uid, *t_row, *d_cnt_br, *d_first_rb;
  int32_t *t_o, *t_pa, *t_rb;
  uint16_t* d_bkt;
  uint64_t* d_base_br;
  const bool in_place = n_bkt == 1;  // one stretch: the order the instances are built in is the order they stay in
  if (in_place) {
    t_uid = cl.inst_uid.as<uint32_t>(); t_row = cl.inst_row.as<uint32_t>(); t_o = cl.inst_o.as<int32_t>();
    t_pa = cl.inst_pa.as<int32_t>(); t_rb = cl.inst_rb.as<int32_t>();
  } else {
    TEMPCHK(tmp, &t_uid, (size_t)n_inst * 4); TEMPCHK(tmp, &t_row, (size_t)n_inst * 4); TEMPCHK(tmp, &t_o, (size_t)n_inst * 4);
    TEMPCHK(tmp, &t_pa, (size_t)n_inst * 4); TEMPCHK(tmp, &t_rb, (size_t)n_inst * 4);
  }
  TEMPCHK(tmp, &d_bkt, (size_t)n_inst * 2);
  TEMPCHK(tmp, &d_cnt_br, (size_t)n * n_bkt * 4); TEMPCHK(tmp, &d_first_rb, (size_t)n * n_bkt * 4); TEMPCHK(tmp, &d_base_br, ((size_t)n * n_bkt + 1) * 8);
  uint32_t *d_rec, *d_n, *d_slot, *d_flag, *d_trep;
  void* d_slot_uid;  // 16 bytes per table slot: the representative's descriptor (k_cl_assign -> k_cl_uid)
  uint64_t *d_key, *d_rank;
  uint8_t* d_cls;
  unsigned long long *d_tkey, *d_partial, *d_shards;
  ScanTotals* d_tot;
  // the hash table of distinct clusters: at least two slots per instance would always do, but on a shared panel the distinct
  // clusters are a small fraction of the instances and clearing 12 bytes x 2^25 slots costs as much as a kernel of this build
  // (C3: 0.08 ms) - so the first attempt takes four slots per distinct cluster EXPECTED (the last build's count, else an eighth
  // of the instances), gives up after 64 probes (status bit 8), and the insert is repeated with the full size
  uint32_t tsize = 1024;
  while (tsize < 2u * n_inst && tsize < (1u << 31)) tsize <<= 1;
  uint32_t tsmall = std::min<uint32_t>(1u << 16, tsize);
  { const uint64_t expect = cl.last_uniq ? (uint64_t)cl.last_uniq * 4 : (uint64_t)n_inst / 2; while (tsmall < expect && tsmall < tsize) tsmall <<= 1; }
  TEMPCHK(tmp, &d_rec, (size_t)n_inst * 4);
  TEMPCHK(tmp, &d_n, (size_t)n_inst * 4);
  TEMPCHK(tmp, &d_slot, (size_t)n_inst * 4);
  TEMPCHK(tmp, &d_flag, (size_t)n_inst * 4);
  TEMPCHK(tmp, &d_key, (size_t)n_inst * 8);
  TEMPCHK(tmp, &d_rank, ((size_t)n_inst + 1) * 8);
  TEMPCHK(tmp, &d_cls, (size_t)n_inst);
  TEMPCHK(tmp, &d_tkey, (size_t)tsize * 8);
  TEMPCHK(tmp, &d_trep, (size_t)tsize * 4);
  TEMPCHK(tmp, &d_partial, ((size_t)std::max<uint64_t>(n_inst, (uint64_t)n * n_bkt) / 1024 + 2) * 8);
  TEMPCHK(tmp, &d_shards, 512 * 8);
  TEMPCHK(tmp, &d_tot, sizeof(ScanTotals) * 2);
  HIPCHK(hipMemsetAsync(d_tkey, 0, (size_t)tsmall * 8, st));
  HIPCHK(hipMemsetAsync(d_trep, 0xff, (size_t)tsmall * 4, st));
  HIPCHK(hipMemsetAsync(d_shards, 0, 512 * 8, st));
  hawk_launch_cl_fill(st, x->recs.p, x->off.as<uint64_t>(), x->hlen.as<uint32_t>(), x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), n, d_off,
                      t_o, t_row, t_pa, t_rb, d_rec, d_n, d_key, d_cls, d_bkt, bshift, n_bkt, d_cnt_br, d_first_rb, d_status);
  hawk_launch_cl_insert(st, n_inst, n_head, d_key, d_cls, d_tkey, d_trep, tsmall - 1, d_slot, d_status, tsmall < tsize ? 64u : 0xffffffffu, tsmall < tsize ? 8u : 2u);
  hawk_launch_cl_flag(st, n_inst, d_cls, d_slot, d_trep, d_flag);
  hawk_launch_mscan(st, d_flag, n_inst, d_partial, d_shards, d_rank, d_tot);
  ScanTotals tot;
  uint32_t st_now = 0;
  HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&st_now, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  uint32_t tused = tsmall;
  if (st_now & 8u) {  // the small table filled up: once more with two slots per instance
    tused = tsize;
    st_now &= ~8u;
    HIPCHK(hipMemcpyAsync(d_status, &st_now, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_tkey, 0, (size_t)tsize * 8, st));
    HIPCHK(hipMemsetAsync(d_trep, 0xff, (size_t)tsize * 4, st));
    HIPCHK(hipMemsetAsync(d_shards, 0, 512 * 8, st));
    HIPCHK(hipMemsetAsync(d_tot, 0, sizeof(ScanTotals), st));
    hawk_launch_cl_insert(st, n_inst, n_head, d_key, d_cls, d_tkey, d_trep, tsize - 1, d_slot, d_status, 0xffffffffu, 2u);
    hawk_launch_cl_flag(st, n_inst, d_cls, d_slot, d_trep, d_flag);
    hawk_launch_mscan(st, d_flag, n_inst, d_partial, d_shards, d_rank, d_tot);
    HIPCHK(hipMemcpyAsync(&tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
  }
  const uint32_t n_uniq = (uint32_t)tot.n_keep;
  TEMPCHK(tmp, &d_slot_uid, (size_t)tused * 16);
  cl.n_inst = n_inst; cl.n_uniq = n_uniq; cl.last_uniq = n_uniq;
  if (n_uniq) {
    uint32_t* d_span2;
    TEMPCHK(tmp, &d_span2, (size_t)n_uniq * 4);
    if ((rc = cl.u_rec.reserve((size_t)n_uniq * 4)) || (rc = cl.u_n.reserve((size_t)n_uniq * 4)) || (rc = cl.u_row.reserve((size_t)n_uniq * 4)) ||
        (rc = cl.u_o.reserve((size_t)n_uniq * 4)) || (rc = cl.u_seg.reserve((size_t)n_uniq * 4)))
      return rc;
    uint64_t* d_slot64;
    TEMPCHK(tmp, &d_slot64, ((size_t)n_uniq + 1) * 8);
    hawk_launch_cl_assign(st, n_inst, d_flag, d_rank, d_slot, d_trep, x->recs.p, t_o, t_row, t_pa, t_rb, d_rec, d_n, d_key, d_cls,
                          x->m_seg_off.as<uint32_t>(), x->m_seg_rel.as<uint32_t>(), d_slot_uid, cl.u_rec.as<uint32_t>(), cl.u_n.as<uint32_t>(),
                          cl.u_row.as<uint32_t>(), cl.u_o.as<int32_t>(), cl.u_seg.as<uint32_t>(), d_span2, t_uid, d_status);
    hawk_launch_mscan(st, d_span2, n_uniq, d_partial, d_shards, d_slot64, d_tot + 1);
    HIPCHK(hipMemcpyAsync(&tot, d_tot + 1, sizeof(tot), hipMemcpyDeviceToHost, st));
  } else {
    HIPCHK(hipMemsetAsync(t_uid, 0xff, (size_t)n_inst * 4, st));
    tot.n_keep = 0;
  }
  if (!in_place) {
    hawk_launch_mscan(st, d_cnt_br, (uint64_t)n * n_bkt, d_partial, d_shards, d_base_br, d_tot);
    hawk_launch_cl_permute(st, n_inst, n, n_bkt, d_bkt, d_base_br, d_first_rb, t_uid, t_o, t_row, t_pa, t_rb, cl.inst_uid.as<uint32_t>(),
                           cl.inst_o.as<int32_t>(), cl.inst_row.as<uint32_t>(), cl.inst_pa.as<int32_t>(), cl.inst_rb.as<int32_t>());
  }
  uint32_t status = 0;
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(ctx->ev[9], st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  (void)hipEventElapsedTime(&cl.build_ms, ctx->ev[8], ctx->ev[9]);
  cl.slots = tot.n_keep;
  cl.status = status;
  // worth it when clusters are shared (the template rows are extra traffic otherwise) and the templates fit a sane budget
  // (HAWK_CLUSTER_MAX_SLOTS template rows, default 2^27 = 10 GB; HAWK_CLUSTER_MIN_SHARE instances per distinct cluster, default 3:
  // read per call so that tests can send small panels down this path)
  const char* e1 = getenv("HAWK_CLUSTER_MAX_SLOTS");
  const char* e2 = getenv("HAWK_CLUSTER_MIN_SHARE");
  const uint64_t max_slots = e1 ? strtoull(e1, nullptr, 10) : (1ull << 27);
  const double min_share = e2 ? atof(e2) : 3.0;
  if (!status && (cl.slots > max_slots || (double)n_inst < min_share * (double)std::max<uint32_t>(n_uniq, 1))) cl.status = 4;
  cl.usable = cl.status == 0;
  return HAWK_OK;
}

int hawk_xplan_cluster_stats(const hawk_xplan* x, uint32_t* usable, uint32_t* n_instances, uint32_t* n_distinct, uint64_t* template_slots,
                             float* build_ms, uint32_t* status) {
  if (!x) return HAWK_E_INVALID;
  if (usable) *usable = x->cl.built && x->cl.usable ? 1u : 0u;
  if (n_instances) *n_instances = x->cl.n_inst;
  if (n_distinct) *n_distinct = x->cl.n_uniq;
  if (template_slots) *template_slots = x->cl.slots;
  if (build_ms) *build_ms = x->cl.build_ms;
  if (status) *status = x->cl.built ? x->cl.status : 0xffffffffu;
  return HAWK_OK;
}

int hawk_xplan_cluster_rebuild(hawk_xplan* x) {
  if (!x || !x->has_meta || x->ref_index != 0) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(x->ctx->device));
  x->cl.built = false;
  return xplan_build_dict(x);
}

int hawk_xplan_view(hawk_xplan* x, hawk_hapset** out) {
  if (!x || !out || !x->has_meta || x->ref_index != 0) return HAWK_E_INVALID;
  hawk_ctx* ctx = x->ctx;
  hawk_hapset* hs = nullptr;
  int rc = hapset_create_impl(ctx, x->n_hap, x->hap_len.data(), false, &hs, false);
  if (rc) return rc;
  if (hs->S != x->S) { hawk_hapset_destroy(hs); return HAWK_E_INVALID; }
  hs->vplan = x;
  for (int p = 0; p < HAWK_PLANES; ++p) hs->plane[p] = x->ref5[p].as<uint32_t>();  // row 0 = REF; no other row is ever read
  rc = xplan_install(x, hs);
  if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = HAWK_E_HIP;
  if (!rc) rc = xplan_build_dict(x);
  if (rc) { hs->vplan = nullptr; for (int p = 0; p < HAWK_PLANES; ++p) hs->plane[p] = nullptr; hawk_hapset_destroy(hs); return rc; }
  *out = hs;
  return HAWK_OK;
}

int hawk_xplan_install_meta(hawk_xplan* x, hawk_hapset* hs) {
  if (!x || !hs || hs->vplan) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(x->ctx->device));
  int rc = xplan_install(x, hs);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(x->ctx->stream));
  return HAWK_OK;
}

int hawk_xplan_create_gt(hawk_hapset* ref_set, hawk_gt* g, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                         const uint32_t* v_alt_off, const uint32_t* v_alt_len, const int32_t* v_chain, const uint8_t* alt_codes,
                         uint32_t alt_codes_len, int64_t startp, int check_clamp, int64_t rev_g0, int64_t rev_g1, uint32_t* n_hap_out,
                         hawk_xplan** out) {
  if (!ref_set || !g || !out || !n_hap_out || !g->d_col_off || g->n_var != n_var || !n_var || !v_r0 || !v_span || !v_alt_off || !v_alt_len ||
      !v_chain || !alt_codes)
    return HAWK_E_INVALID;
  hawk_ctx* ctx = ref_set->ctx;
  if (g->ctx != ctx) return HAWK_E_INVALID;
  const uint32_t ref_len = ref_set->hap_len[0];
  for (uint32_t i = 0; i < n_var; ++i) {  // the variant table: everything the kernels index with (as hawk_xplan_create)
    if ((uint64_t)v_r0[i] + v_span[i] > ref_len || v_span[i] == 0 || v_alt_len[i] == 0) return HAWK_E_INVALID;
    if ((uint64_t)v_alt_off[i] + v_alt_len[i] > alt_codes_len) return HAWK_E_INVALID;
    if (i && v_r0[i] < v_r0[i - 1]) return HAWK_E_INVALID;
    if ((int64_t)v_chain[i] != (int64_t)v_alt_len[i] - (int64_t)v_span[i]) return HAWK_E_INVALID;
  }
  // rows: REF, then every chromosome copy (column) that carries something, in column order.  Columns without a variant
  // contribute no entry, so the rows' lists are the inversion's list array as it stands.
  const uint32_t n_cols = 2 * g->n_samples;
  std::vector<uint64_t> hv_off(2, 0), ioff(2, 0);
  std::vector<uint32_t> hap_len(1, ref_len);
  uint32_t maxlen = ref_len;
  for (uint32_t c = 0; c < n_cols; ++c) {
    if (g->h_off[c + 1] == g->h_off[c]) continue;
    const int64_t len = (int64_t)ref_len + g->h_delta[c];
    if (len <= 0 || len >= (int64_t)((1u << 31) - 256)) return HAWK_E_UNSUPPORTED;
    hv_off.push_back(g->h_off[c + 1]);
    ioff.push_back(g->h_ioff[c + 1]);
    hap_len.push_back((uint32_t)len);
    maxlen = std::max(maxlen, (uint32_t)len);
  }
  const uint32_t n_hap = (uint32_t)hap_len.size();
  *n_hap_out = n_hap;
  hawk_xplan* x = nullptr;
  int rc = xplan_build(ref_set, n_var, v_r0, v_span, v_alt_off, v_alt_len, alt_codes, alt_codes_len, n_hap, hv_off.data(), nullptr, nullptr,
                       g->d_idx, g->d_o, hap_len.data(), maxlen, &x);
  if (rc) return rc;
  // ---- checks over every list entry, position-map segments, the two reverse look-ups: device work over the lists in place
  hipStream_t st = ctx->stream;
  DevBuf t_r0, t_span, t_ch, t_ioff, t_cnt, t_status, t_rev;
  DevBuf* temps[] = {&t_r0, &t_span, &t_ch, &t_ioff, &t_cnt, &t_status, &t_rev};
  auto done = [&](int code) { for (auto* b : temps) b->release(); if (code) hawk_xplan_destroy(x); return code; };
  if ((rc = t_r0.reserve((size_t)n_var * 4)) || (rc = t_span.reserve((size_t)n_var * 4)) || (rc = t_ch.reserve((size_t)n_var * 4)) ||
      (rc = t_ioff.reserve((size_t)(n_hap + 1) * 8)) || (rc = t_cnt.reserve((size_t)n_hap * 4)) || (rc = t_status.reserve(16)) ||
      (rc = t_rev.reserve((size_t)n_hap * 16)) || (rc = x->m_seg_off.reserve((size_t)(n_hap + 1) * 4)))
    return done(rc);
#define HIPCHK_X(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return done(HAWK_E_HIP);                                                                 \
    }                                                                                          \
  } while (0)
  HIPCHK_X(hipMemcpyAsync(t_r0.p, v_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK_X(hipMemcpyAsync(t_span.p, v_span, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK_X(hipMemcpyAsync(t_ch.p, v_chain, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK_X(hipMemcpyAsync(t_ioff.p, ioff.data(), (size_t)(n_hap + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK_X(hipMemsetAsync(t_status.p, 0, 16, st));
  hawk_launch_list_check(st, x->off.as<uint64_t>(), n_hap, g->d_idx, g->d_o, t_r0.as<int32_t>(), t_span.as<int32_t>(), t_ch.as<int32_t>(), n_var,
                         ref_len, check_clamp, t_status.as<uint32_t>());
  hawk_launch_segments(st, t_ioff.as<uint64_t>(), g->d_indel, g->d_idx, g->d_o, t_r0.as<int32_t>(), t_ch.as<int32_t>(), x->hlen.as<uint32_t>(), n_hap,
                       startp, t_cnt.as<uint32_t>(), x->m_seg_off.as<uint32_t>(), nullptr, nullptr);
  HIPCHK_X(hipGetLastError());
  uint32_t status = 0, nseg = 0;
  HIPCHK_X(hipMemcpyAsync(&status, t_status.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK_X(hipMemcpyAsync(&nseg, x->m_seg_off.as<uint32_t>() + n_hap, 4, hipMemcpyDeviceToHost, st));
  HIPCHK_X(hipStreamSynchronize(st));
  if (status & 4u) return done(HAWK_E_INVALID);      // a list out of order: not what the inversion writes
  if (status & 1u) return done(HAWK_E_OVERLAP);      // a chromosome copy carries overlapping variants (haplotype.py:214-252 raises)
  if (status & 2u) return done(HAWK_E_CLAMP);        // an indel beyond the region's original length (haplotype.py:199-201)
  if ((rc = x->m_seg_rel.reserve((size_t)nseg * 4)) || (rc = x->m_seg_gen.reserve((size_t)nseg * 8))) return done(rc);
  hawk_launch_segments(st, t_ioff.as<uint64_t>(), g->d_indel, g->d_idx, g->d_o, t_r0.as<int32_t>(), t_ch.as<int32_t>(), x->hlen.as<uint32_t>(), n_hap,
                       startp, t_cnt.as<uint32_t>(), x->m_seg_off.as<uint32_t>(), x->m_seg_rel.as<uint32_t>(), x->m_seg_gen.as<int64_t>());
  hawk_launch_rev_lookup(st, x->m_seg_off.as<uint32_t>(), x->m_seg_rel.as<uint32_t>(), x->m_seg_gen.as<int64_t>(), x->hlen.as<uint32_t>(), n_hap,
                         rev_g0, rev_g1, t_rev.as<int64_t>(), t_rev.as<int64_t>() + n_hap);
  HIPCHK_X(hipGetLastError());
  x->rev0.resize(n_hap); x->rev1.resize(n_hap);
  HIPCHK_X(hipMemcpyAsync(x->rev0.data(), t_rev.p, (size_t)n_hap * 8, hipMemcpyDeviceToHost, st));
  HIPCHK_X(hipMemcpyAsync(x->rev1.data(), t_rev.as<int64_t>() + n_hap, (size_t)n_hap * 8, hipMemcpyDeviceToHost, st));
  HIPCHK_X(hipStreamSynchronize(st));
#undef HIPCHK_X
  x->nseg = nseg; x->ref_startp = startp;
  x->min_gen = startp; x->max_gen = startp + (int64_t)ref_len;  // every row's positions map into REF's range
  (void)done(HAWK_OK);
  *out = x;
  return HAWK_OK;
}

int hawk_xplan_rows(hawk_xplan* x, uint32_t* hap_len, int64_t* rev0, int64_t* rev1) {
  if (!x) return HAWK_E_INVALID;
  if (hap_len) memcpy(hap_len, x->hap_len.data(), (size_t)x->n_hap * 4);
  if ((rev0 || rev1) && x->rev0.size() != x->n_hap) return HAWK_E_INVALID;
  if (rev0) memcpy(rev0, x->rev0.data(), (size_t)x->n_hap * 8);
  if (rev1) memcpy(rev1, x->rev1.data(), (size_t)x->n_hap * 8);
  return HAWK_OK;
}

int hawk_xplan_finish_meta(hawk_xplan* x, const int32_t* scan_start, const int32_t* scan_stop) {
  if (!x || !scan_start || !scan_stop || !x->nseg) return HAWK_E_INVALID;
  hawk_ctx* ctx = x->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n = x->n_hap;
  for (uint32_t h = 0; h < n; ++h)
    if (scan_start[h] < 0 || scan_stop[h] > (int32_t)x->hap_len[h]) return HAWK_E_INVALID;
  int rc;
  if ((rc = x->m_is_ref.reserve(n)) || (rc = x->m_ss.reserve((size_t)n * 4)) || (rc = x->m_se.reserve((size_t)n * 4)) ||
      (rc = x->m_tile.reserve((size_t)n * x->bph * sizeof(TileMeta))))
    return rc;
  hipStream_t st = ctx->stream;
  std::vector<uint8_t> is_ref(n, 0);
  is_ref[0] = 1;
  HIPCHK(hipMemcpyAsync(x->m_is_ref.p, is_ref.data(), n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_ss.p, scan_start, (size_t)n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(x->m_se.p, scan_stop, (size_t)n * 4, hipMemcpyHostToDevice, st));
  hawk_launch_tile_meta(st, x->m_seg_off.as<uint32_t>(), x->m_seg_rel.as<uint32_t>(), x->hlen.as<uint32_t>(), x->m_is_ref.as<uint8_t>(),
                        x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), n, x->bph, x->m_tile.as<TileMeta>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  x->scan_start.assign(scan_start, scan_start + n);
  x->scan_stop.assign(scan_stop, scan_stop + n);
  x->ref_index = 0; x->n_ref_rows = 1; x->has_meta = true; x->cl.built = false; x->cl.usable = false;
  return HAWK_OK;
}

int hawk_xplan_segments(hawk_xplan* x, uint32_t* seg_off, uint32_t* seg_rel, int64_t* seg_gen, uint64_t cap, uint64_t* n_seg) {
  if (!x || !x->nseg) return HAWK_E_INVALID;
  if (n_seg) *n_seg = x->nseg;
  if (!seg_off && !seg_rel && !seg_gen) return HAWK_OK;
  if (cap < x->nseg) return HAWK_E_CAPACITY;
  hawk_ctx* ctx = x->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  if (seg_off) HIPCHK(hipMemcpyAsync(seg_off, x->m_seg_off.p, (size_t)(x->n_hap + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (seg_rel) HIPCHK(hipMemcpyAsync(seg_rel, x->m_seg_rel.p, (size_t)x->nseg * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (seg_gen) HIPCHK(hipMemcpyAsync(seg_gen, x->m_seg_gen.p, (size_t)x->nseg * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_hapset_expand(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                       const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len,
                       uint32_t n_hap, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o,
                       const uint32_t* hap_len, hawk_hapset** out, uint64_t* hash_out, float* kernel_ms) {
  hawk_xplan* x = nullptr;
  int rc = hawk_xplan_create(ref_set, n_var, v_r0, v_span, v_alt_off, v_alt_len, alt_codes, alt_codes_len, n_hap, hv_off, hv_idx, hv_o,
                             hap_len, &x);
  if (rc) return rc;
  std::vector<uint64_t> tmp;
  if (!hash_out) { tmp.resize((size_t)n_hap * 2); hash_out = tmp.data(); }  // run synchronously either way
  rc = hawk_xplan_run(x, out, hash_out, kernel_ms);
  hawk_xplan_destroy(x);
  return rc;
}

// ---------------------------------------------------------------------------- f3: VCF genotypes

int hawk_gt_parse(hawk_ctx* ctx, const uint8_t* text, uint64_t text_len, const uint64_t* line_off, const uint64_t* gt_off,
                  uint64_t n_lines, uint32_t n_samples, hawk_gt** out, float* kernel_ms) {
  if (!ctx || !out || !n_samples || (n_lines && (!text || !line_off || !gt_off))) return HAWK_E_INVALID;
  // every offset the kernel dereferences is checked here
  for (uint64_t i = 0; i < n_lines; ++i)
    if (line_off[i + 1] > text_len || line_off[i] >= line_off[i + 1] || gt_off[i] < line_off[i] || gt_off[i] > line_off[i + 1])
      return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_gt* g = new (std::nothrow) hawk_gt();
  if (!g) return HAWK_E_INVALID;
  g->ctx = ctx; g->n_lines = n_lines; g->n_samples = n_samples; g->n_var = 0; g->n_entries = 0;
  g->d_codes = nullptr; g->d_flags = nullptr; g->d_col_off = nullptr; g->d_idx = nullptr; g->d_o = nullptr; g->d_delta = nullptr;
  if (kernel_ms) *kernel_ms = 0.f;
  const size_t ncode = std::max<size_t>((size_t)n_lines * 2 * n_samples, 1);
  POOLCHK(&g->d_codes, ncode); POOLCHK(&g->d_flags, std::max<size_t>(n_lines, 1));
  if (n_lines) {
    uint8_t* d_text = nullptr; uint64_t *d_lo = nullptr, *d_go = nullptr;
    PoolScope tmp;
    TEMPCHK(tmp, &d_text, text_len); TEMPCHK(tmp, &d_lo, (n_lines + 1) * 8); TEMPCHK(tmp, &d_go, n_lines * 8);
    hipStream_t st = ctx->stream;
    HIPCHK(hipMemcpyAsync(d_text, text, text_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_lo, line_off, (n_lines + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_go, gt_off, n_lines * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(g->d_codes, 0xff, ncode, st));  // samples a short record does not reach read as missing
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    hawk_launch_gt_parse(st, d_text, d_lo, d_go, n_lines, n_samples, g->d_codes, g->d_flags);
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
  }
  *out = g;
  return HAWK_OK;
}

int hawk_gt_from_codes(hawk_ctx* ctx, const uint8_t* codes, uint64_t n_lines, uint32_t n_samples, hawk_gt** out) {
  if (!ctx || !out || !n_samples || (n_lines && !codes)) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_gt* g = new (std::nothrow) hawk_gt();
  if (!g) return HAWK_E_INVALID;
  g->ctx = ctx; g->n_lines = n_lines; g->n_samples = n_samples; g->n_var = 0; g->n_entries = 0;
  g->d_codes = nullptr; g->d_flags = nullptr; g->d_col_off = nullptr; g->d_idx = nullptr; g->d_o = nullptr; g->d_delta = nullptr;
  const size_t ncode = std::max<size_t>((size_t)n_lines * 2 * n_samples, 1);
  POOLCHK(&g->d_codes, ncode); POOLCHK(&g->d_flags, std::max<size_t>(n_lines, 1));
  if (n_lines) {
    HIPCHK(hipMemcpyAsync(g->d_codes, codes, (size_t)n_lines * 2 * n_samples, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(g->d_flags, 0, n_lines, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  *out = g;
  return HAWK_OK;
}

void hawk_gt_destroy(hawk_gt* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  hawk_pool_free(g->d_codes); hawk_pool_free(g->d_flags);
  if (g->d_col_off) hawk_pool_free(g->d_col_off);
  if (g->d_idx) hawk_pool_free(g->d_idx);
  if (g->d_o) hawk_pool_free(g->d_o);
  if (g->d_delta) hawk_pool_free(g->d_delta);
  if (g->d_indel) hawk_pool_free(g->d_indel);
  if (g->d_ioff) hawk_pool_free(g->d_ioff);
  delete g;
}

int hawk_gt_codes(hawk_gt* g, uint8_t* codes, uint8_t* line_flags) {
  if (!g) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (codes && g->n_lines) HIPCHK(hipMemcpyAsync(codes, g->d_codes, (size_t)g->n_lines * 2 * g->n_samples, hipMemcpyDefault, g->ctx->stream));
  if (line_flags && g->n_lines) HIPCHK(hipMemcpyAsync(line_flags, g->d_flags, g->n_lines, hipMemcpyDefault, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

int hawk_gt_lists(hawk_gt* g, const uint32_t* var_line, const uint8_t* var_allele, const int32_t* var_r0, const int32_t* var_chain,
                  uint32_t n_var, uint64_t* col_off, int64_t* col_delta, float* kernel_ms) {
  if (!g || !col_off || (n_var && (!var_line || !var_allele || !var_r0 || !var_chain))) return HAWK_E_INVALID;
  for (uint32_t j = 0; j < n_var; ++j)
    if (var_line[j] >= g->n_lines || var_allele[j] == 0 || var_allele[j] == 255) return HAWK_E_INVALID;
  hawk_ctx* ctx = g->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n_cols = 2 * g->n_samples, n_chunk = (n_var + 63u) / 64u;
  hipStream_t st = ctx->stream;
  if (g->d_col_off) { hawk_pool_free(g->d_col_off); g->d_col_off = nullptr; }
  if (g->d_idx) { hawk_pool_free(g->d_idx); g->d_idx = nullptr; }
  if (g->d_o) { hawk_pool_free(g->d_o); g->d_o = nullptr; }
  if (g->d_delta) { hawk_pool_free(g->d_delta); g->d_delta = nullptr; }
  if (g->d_indel) { hawk_pool_free(g->d_indel); g->d_indel = nullptr; }
  if (g->d_ioff) { hawk_pool_free(g->d_ioff); g->d_ioff = nullptr; }
  g->h_off.assign(n_cols + 1, 0); g->h_ioff.assign(n_cols + 1, 0); g->h_delta.assign(n_cols, 0);
  g->n_var = n_var; g->n_entries = 0; g->n_indel = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  std::vector<uint64_t> off(n_cols + 1, 0);
  if (n_var == 0) {
    memcpy(col_off, off.data(), (n_cols + 1) * 8);
    if (col_delta) memset(col_delta, 0, (size_t)n_cols * 8);
    return HAWK_OK;
  }
  uint32_t *d_vl = nullptr, *d_cnt = nullptr; uint8_t* d_va = nullptr; int32_t *d_r0 = nullptr, *d_ch = nullptr;
  unsigned long long* d_bal = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_vl, (size_t)n_var * 4); TEMPCHK(tmp, &d_va, n_var); TEMPCHK(tmp, &d_r0, (size_t)n_var * 4);
  TEMPCHK(tmp, &d_ch, (size_t)n_var * 4); TEMPCHK(tmp, &d_cnt, (size_t)n_cols * 2 * 4);
  POOLCHK(&g->d_ioff, (size_t)(n_cols + 1) * 8);
  uint64_t* d_ioff = g->d_ioff;
  TEMPCHK(tmp, &d_bal, (size_t)n_cols * n_chunk * 8);
  POOLCHK(&g->d_col_off, (size_t)(n_cols + 1) * 8); POOLCHK(&g->d_delta, (size_t)n_cols * 8);
  HIPCHK(hipMemcpyAsync(d_vl, var_line, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_va, var_allele, n_var, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r0, var_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ch, var_chain, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hawk_launch_gt_count(st, g->d_codes, n_cols, d_vl, d_va, d_ch, n_var, d_bal, d_cnt);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  std::vector<uint32_t> cnt(2 * (size_t)n_cols);
  HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n_cols * 2 * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<uint64_t> ioff(n_cols + 1, 0);
  for (uint32_t c = 0; c < n_cols; ++c) {  // 2 * n_samples values: host prefix sums
    off[c + 1] = off[c] + cnt[c];
    ioff[c + 1] = ioff[c] + cnt[n_cols + c];
  }
  const uint64_t ne = off[n_cols], ni = ioff[n_cols];
  POOLCHK(&g->d_idx, std::max<size_t>(ne, 1) * 4); POOLCHK(&g->d_o, std::max<size_t>(ne, 1) * 4);
  POOLCHK(&g->d_indel, std::max<size_t>(ni, 1) * 4);
  HIPCHK(hipMemcpyAsync(g->d_col_off, off.data(), (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ioff, ioff.data(), (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  hawk_launch_gt_fill(st, n_cols, d_r0, d_ch, n_var, d_bal, g->d_col_off, g->d_idx, g->d_o, g->d_delta, d_ioff, g->d_indel);
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(g->h_delta.data(), g->d_delta, (size_t)n_cols * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (col_delta) memcpy(col_delta, g->h_delta.data(), (size_t)n_cols * 8);
  g->h_off = off; g->h_ioff = ioff;
  if (kernel_ms) {
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]); (void)hipEventElapsedTime(&b, ctx->ev[2], ctx->ev[3]);
    *kernel_ms = a + b;
  }
  memcpy(col_off, off.data(), (size_t)(n_cols + 1) * 8);
  g->n_entries = ne;
  g->n_indel = ni;
  return HAWK_OK;
}

int hawk_gt_lists_indels(hawk_gt* g, uint32_t* entry_idx, uint64_t cap, uint64_t* n_indel) {
  if (!g || !n_indel) return HAWK_E_INVALID;
  *n_indel = g->n_indel;
  const uint64_t k = std::min<uint64_t>(cap, g->n_indel);
  if (!k || !entry_idx) return HAWK_OK;
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(entry_idx, g->d_indel, k * 4, hipMemcpyDefault, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

int hawk_gt_lists_download(hawk_gt* g, uint32_t* hv_idx, int32_t* hv_o) {
  if (!g) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (g->n_entries) {
    if (hv_idx) HIPCHK(hipMemcpyAsync(hv_idx, g->d_idx, g->n_entries * 4, hipMemcpyDefault, g->ctx->stream));
    if (hv_o) HIPCHK(hipMemcpyAsync(hv_o, g->d_o, g->n_entries * 4, hipMemcpyDefault, g->ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- K6 DeepCpf1
int hawk_deepcpf1(hawk_ctx* ctx, const char* seqs34, uint64_t n, const float* weights, float* out) {
  if (!ctx || !weights || (n && (!seqs34 || !out))) return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nw = HAWK_DEEPCPF1_NPARAMS;
  char* d_s = nullptr; float *d_w = nullptr, *d_o = nullptr; int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_s, n * 34); TEMPCHK(tmp, &d_w, nw * 4); TEMPCHK(tmp, &d_o, n * 4); TEMPCHK(tmp, &d_status, 4);
  HIPCHK(hipMemcpyAsync(d_s, seqs34, n * 34, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_w, weights, nw * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
  hawk_launch_deepcpf1(ctx->stream, d_s, n, d_w, d_o, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return status;
}

// ---------------------------------------------------------------------------- K5 Azimuth
int hawk_tm_nn(hawk_ctx* ctx, const char* seqs, uint32_t len, uint64_t n, double* out) {
  if (!ctx || (n && (!seqs || !out))) return HAWK_E_INVALID;
  if (len < 2 || len > 32) return HAWK_E_UNSUPPORTED;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  PoolScope tmp;
  char* d_s; double* d_o; int* d_status;
  TEMPCHK(tmp, &d_s, n * len); TEMPCHK(tmp, &d_o, n * 8); TEMPCHK(tmp, &d_status, 4);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_s, seqs, n * len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, st));
  hawk_launch_tm_nn(st, d_s, len, n, d_o, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return status;
}

int hawk_azimuth(hawk_ctx* ctx, const char* seqs30, uint64_t n, const hawk_gbt_model* m, double* out, double* feats_out) {
  if (!ctx || !m || !m->tree_off || !m->feature || !m->left || !m->right || !m->threshold || !m->value ||
      (n && (!seqs30 || !out)))
    return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  // validate the trees on the host: every child index inside its tree, every feature < 627
  for (uint32_t t = 0; t < m->n_trees; ++t) {
    const int32_t lo = m->tree_off[t], hi = m->tree_off[t + 1];
    if (lo < 0 || hi <= lo || (uint32_t)hi > m->n_nodes) return HAWK_E_INVALID;
    for (int32_t k = lo; k < hi; ++k) {
      if (m->feature[k] >= 627) return HAWK_E_INVALID;
      if (m->feature[k] >= 0 && (m->left[k] <= k - lo || m->right[k] <= k - lo || m->left[k] >= hi - lo || m->right[k] >= hi - lo))
        return HAWK_E_INVALID;  // children must point forward inside the tree: traversal terminates
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nn = m->n_nodes, nt = m->n_trees;
  char* d_s = nullptr; int32_t *d_off = nullptr, *d_f = nullptr, *d_l = nullptr, *d_r = nullptr;
  double *d_th = nullptr, *d_v = nullptr, *d_o = nullptr, *d_fo = nullptr; int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_s, n * 30); TEMPCHK(tmp, &d_off, (nt + 1) * 4); TEMPCHK(tmp, &d_f, nn * 4);
  TEMPCHK(tmp, &d_l, nn * 4); TEMPCHK(tmp, &d_r, nn * 4); TEMPCHK(tmp, &d_th, nn * 8); TEMPCHK(tmp, &d_v, nn * 8);
  TEMPCHK(tmp, &d_o, n * 8); TEMPCHK(tmp, &d_status, 4);
  if (feats_out) TEMPCHK(tmp, &d_fo, n * 627 * 8);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_s, seqs30, n * 30, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, m->tree_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_f, m->feature, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_l, m->left, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r, m->right, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_th, m->threshold, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_v, m->value, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, st));
  hawk_launch_azimuth(st, d_s, n, m->n_trees, d_off, d_f, d_l, d_r, d_th, d_v, m->init, m->learning_rate, d_o, d_fo, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  if (feats_out) HIPCHK(hipMemcpyAsync(feats_out, d_fo, n * 627 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return status;
}

}  // extern "C"
