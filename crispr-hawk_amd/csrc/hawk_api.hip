// hawk_api.hip — the C ABI of include/hawk.h over the kernels in hawk_kernels.hip.
// Host-side responsibilities only: HBM allocation, H2D/D2H staging, launch order on one HIP
// stream, HIP-event timing, error mapping.  No compute happens on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/hawk.h"
#include "hawk_device.h"

static thread_local char g_hip_err[256] = "";

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return HAWK_E_HIP;                                                                     \
    }                                                                                        \
  } while (0)

struct hawk_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev[8];
};

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return HAWK_OK;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    size_t want = need + need / 8 + 256;
    HIPCHK(hipMalloc(&p, want));
    bytes = want;
    return HAWK_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct hawk_hapset {
  hawk_ctx* ctx;
  uint32_t n_hap, S;
  uint64_t total_len;
  std::vector<uint32_t> hap_len;
  std::vector<int32_t> scan_start, scan_stop;
  uint32_t* plane[HAWK_PLANES];
  uint32_t* d_hap_len;
  uint8_t* d_is_ref;
  int32_t *d_scan_start, *d_scan_stop;
  uint32_t *d_seg_off, *d_seg_rel;
  int64_t* d_seg_gen;
  int32_t ref_index;
  bool has_meta;
  uint32_t bph;            // workgroups (tiles of 1024 words) per haplotype row
  TileMeta* d_tile_meta;   // [n_hap * bph] per-tile record (haplotype scalars + first position-map segment)
  int64_t ref_startp;
  int64_t min_gen, max_gen;  // range of genomic positions the position maps reach (collapse sort key)
  uint64_t cols_cap = 0;      // rows the guide-table columns currently hold (0: never reserved)
  std::vector<double> cfd_host;  // the CFD tables resident in `cfd`
  // workspace reused across searches
  DevBuf keepF, keepR, counts, offsets, totals, misc, cfd, partial, sites, hits, guides, lists;
  DevBuf ckeys, cvals, cflags, cgidx, ctemp, cgoff, cgc, ccnt, cfull;  // hawk_table_collapse
  DevBuf otoff, otcode, otid, othit;  // hawk_offtarget_scan: bucketed guides, gathered hit sites
  DevBuf colsA[8];
};

struct hawk_table {
  hawk_hapset* hs;
  uint64_t n_rows, n_cand, n_hits, cap;
  GuideCols cols;  // points into hs->colsA or colsB
  uint32_t guidelen, pamlen, right;
  uint64_t n_groups;  // valid after hawk_table_collapse
  bool collapsed;
};

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

int hawk_init(int device, hawk_ctx** out) {
  if (!out) return HAWK_E_INVALID;
  int c = 0;
  hawk_device_count(&c);
  if (c <= 0) return HAWK_E_NODEVICE;
  if (device < 0 || device >= c) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(device));
  hawk_ctx* ctx = new (std::nothrow) hawk_ctx();
  if (!ctx) return HAWK_E_INVALID;
  ctx->device = device;
  HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (auto& e : ctx->ev) HIPCHK(hipEventCreate(&e));
  *out = ctx;
  return HAWK_OK;
}

void hawk_destroy(hawk_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void* hawk_stream(hawk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int hawk_sync(hawk_ctx* ctx) {
  if (!ctx) return HAWK_E_INVALID;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- hapset
int hawk_hapset_create(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, hawk_hapset** out) {
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
  for (int p = 0; p < HAWK_PLANES; ++p) {
    hs->plane[p] = nullptr;
    HIPCHK(hipMalloc(&hs->plane[p], words * 4));
    HIPCHK(hipMemsetAsync(hs->plane[p], 0, words * 4, ctx->stream));
  }
  HIPCHK(hipMalloc(&hs->d_hap_len, n_hap * 4));
  HIPCHK(hipMemcpyAsync(hs->d_hap_len, hap_len, n_hap * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMalloc(&hs->d_is_ref, n_hap));
  HIPCHK(hipMalloc(&hs->d_scan_start, n_hap * 4));
  HIPCHK(hipMalloc(&hs->d_scan_stop, n_hap * 4));
  HIPCHK(hipMalloc(&hs->d_seg_off, (n_hap + 1) * 4));
  hs->d_seg_rel = nullptr;
  hs->d_seg_gen = nullptr;
  hs->bph = (hs->S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hs->d_tile_meta = nullptr;
  HIPCHK(hipMalloc(&hs->d_tile_meta, (size_t)n_hap * hs->bph * sizeof(TileMeta)));
  hs->ref_startp = 0;
  hs->ref_index = -1;
  hs->has_meta = false;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *out = hs;
  return HAWK_OK;
}

void hawk_hapset_destroy(hawk_hapset* hs) {
  if (!hs) return;
  (void)hipSetDevice(hs->ctx->device);
  (void)hipStreamSynchronize(hs->ctx->stream);
  for (int p = 0; p < HAWK_PLANES; ++p) (void)hipFree(hs->plane[p]);
  (void)hipFree(hs->d_hap_len); (void)hipFree(hs->d_is_ref); (void)hipFree(hs->d_scan_start);
  (void)hipFree(hs->d_scan_stop); (void)hipFree(hs->d_seg_off);
  if (hs->d_seg_rel) (void)hipFree(hs->d_seg_rel);
  if (hs->d_seg_gen) (void)hipFree(hs->d_seg_gen);
  (void)hipFree(hs->d_tile_meta);
  DevBuf* bufs[] = {&hs->keepF, &hs->keepR, &hs->counts, &hs->offsets, &hs->totals, &hs->misc, &hs->cfd, &hs->partial,
                    &hs->sites, &hs->hits, &hs->guides, &hs->lists, &hs->ckeys, &hs->cvals, &hs->cflags, &hs->cgidx,
                    &hs->ctemp, &hs->cgoff, &hs->cgc, &hs->ccnt, &hs->cfull, &hs->otoff, &hs->otcode, &hs->otid, &hs->othit};
  for (auto* b : bufs) b->release();
  for (auto& b : hs->colsA) b.release();
  delete hs;
}

int hawk_hapset_stride(const hawk_hapset* hs, uint32_t* stride_words) {
  if (!hs || !stride_words) return HAWK_E_INVALID;
  *stride_words = hs->S;
  return HAWK_OK;
}

int hawk_hapset_pack_ascii(hawk_hapset* hs, const char* seqs, const uint64_t* seq_off, uint64_t* bad_index) {
  if (!hs || !seqs || !seq_off) return HAWK_E_INVALID;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  for (uint32_t h = 0; h < hs->n_hap; ++h)
    if (seq_off[h + 1] - seq_off[h] != hs->hap_len[h]) return HAWK_E_INVALID;
  uint64_t* d_off = nullptr;
  unsigned long long* d_bad = nullptr;
  HIPCHK(hipMalloc(&d_off, (hs->n_hap + 1) * 8));
  HIPCHK(hipMalloc(&d_bad, 8));
  HIPCHK(hipMemcpyAsync(d_off, seq_off, (hs->n_hap + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_bad, 0xff, 8, ctx->stream));
  // stage the ASCII in batches of whole haplotypes (<= 256 MiB of HBM staging)
  const uint64_t kStage = 256ull << 20;
  uint64_t maxlen = 0;
  for (auto l : hs->hap_len) maxlen = std::max<uint64_t>(maxlen, l);
  const uint64_t stage_bytes = std::max(kStage, maxlen);
  uint8_t* d_stage = nullptr;
  HIPCHK(hipMalloc(&d_stage, std::min<uint64_t>(stage_bytes, std::max<uint64_t>(hs->total_len, 1))));
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
  (void)hipFree(d_stage); (void)hipFree(d_off); (void)hipFree(d_bad);
  if (bad != ~0ull) {
    if (bad_index) *bad_index = bad;
    return HAWK_E_IUPAC;
  }
  return HAWK_OK;
}

int hawk_hapset_set_meta(hawk_hapset* hs, const uint8_t* is_ref, const int32_t* scan_start, const int32_t* scan_stop,
                         const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, int32_t ref_index) {
  if (!hs || !is_ref || !scan_start || !scan_stop || !seg_off || !seg_rel || !seg_gen) return HAWK_E_INVALID;
  if (ref_index >= (int32_t)hs->n_hap) return HAWK_E_INVALID;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n = hs->n_hap;
  for (uint32_t h = 0; h < n; ++h) {
    // the kernels trust these: validate on the host before anything is launched
    if (seg_off[h + 1] <= seg_off[h] || seg_rel[seg_off[h]] != 0) return HAWK_E_INVALID;
    if (scan_start[h] < 0 || scan_stop[h] > (int32_t)hs->hap_len[h]) return HAWK_E_INVALID;
    for (uint32_t k = seg_off[h] + 1; k < seg_off[h + 1]; ++k)
      if (seg_rel[k] <= seg_rel[k - 1]) return HAWK_E_INVALID;
  }
  if (ref_index >= 0 && seg_off[ref_index + 1] - seg_off[ref_index] != 1) return HAWK_E_INVALID;
  const uint32_t nseg = seg_off[n];
  if (hs->d_seg_rel) { (void)hipFree(hs->d_seg_rel); hs->d_seg_rel = nullptr; }
  if (hs->d_seg_gen) { (void)hipFree(hs->d_seg_gen); hs->d_seg_gen = nullptr; }
  HIPCHK(hipMalloc(&hs->d_seg_rel, (size_t)nseg * 4));
  HIPCHK(hipMalloc(&hs->d_seg_gen, (size_t)nseg * 8));
  HIPCHK(hipMemcpyAsync(hs->d_is_ref, is_ref, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_scan_start, scan_start, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_scan_stop, scan_stop, n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_off, seg_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_rel, seg_rel, (size_t)nseg * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(hs->d_seg_gen, seg_gen, (size_t)nseg * 8, hipMemcpyHostToDevice, ctx->stream));
  {
    // first segment each tile needs: the last one starting at or before the tile's first base
    std::vector<TileMeta> t0((size_t)n * hs->bph);
    for (uint32_t h = 0; h < n; ++h) {
      const uint32_t* b = seg_rel + seg_off[h];
      const uint32_t* e = seg_rel + seg_off[h + 1];
      for (uint32_t blk = 0; blk < hs->bph; ++blk) {
        const uint32_t q0 = blk * HAWK_BLOCK * 128u;
        const uint32_t* it = std::upper_bound(b, e, q0);  // first seg_rel > q0
        TileMeta& t = t0[(size_t)h * hs->bph + blk];
        t.h = h; t.blk = blk; t.hap_len = hs->hap_len[h];
        t.scan_start = scan_start[h]; t.scan_stop = scan_stop[h]; t.is_ref = is_ref[h] ? 1u : 0u;
        t.seg0 = (uint32_t)((it - seg_rel) - 1); t.seg_end = seg_off[h + 1];
      }
    }
    HIPCHK(hipMemcpyAsync(hs->d_tile_meta, t0.data(), t0.size() * sizeof(TileMeta), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  hs->ref_startp = ref_index >= 0 ? seg_gen[seg_off[ref_index]] : 0;
  hs->min_gen = INT64_MAX; hs->max_gen = INT64_MIN;
  for (uint32_t h = 0; h < n; ++h)
    for (uint32_t k = seg_off[h]; k < seg_off[h + 1]; ++k) {
      const uint32_t end = k + 1 < seg_off[h + 1] ? seg_rel[k + 1] : hs->hap_len[h];
      hs->min_gen = std::min(hs->min_gen, seg_gen[k]);
      hs->max_gen = std::max(hs->max_gen, seg_gen[k] + (int64_t)(end - seg_rel[k]));
    }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hs->scan_start.assign(scan_start, scan_start + n);
  hs->scan_stop.assign(scan_stop, scan_stop + n);
  hs->ref_index = ref_index;
  hs->has_meta = true;
  return HAWK_OK;
}

int hawk_hapset_download_plane(hawk_hapset* hs, int plane, uint32_t* out_words) {
  if (!hs || plane < 0 || plane >= HAWK_PLANES || !out_words) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(hs->ctx->device));
  HIPCHK(hipStreamSynchronize(hs->ctx->stream));
  HIPCHK(hipMemcpy(out_words, hs->plane[plane], (size_t)hs->n_hap * hs->S * 4, hipMemcpyDeviceToHost));
  return HAWK_OK;
}

int hawk_hapset_upload_planes(hawk_hapset* hs, const uint32_t* planes) {
  if (!hs || !planes) return HAWK_E_INVALID;
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
  HIPCHK(hipMalloc(&d_f, std::max<uint64_t>(nf, 1) * 4));
  HIPCHK(hipMalloc(&d_r, std::max<uint64_t>(nr, 1) * 4));
  hawk_launch_emit_hits(ctx->stream, d, sp.bph, hs->keepF.as<uint32_t>(), hs->keepR.as<uint32_t>(),
                        hs->offsets.as<uint64_t>(), nf, d_f, d_r);
  HIPCHK(hipGetLastError());
  if (nf) HIPCHK(hipMemcpyAsync(hits_fwd, d_f, nf * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (nr) HIPCHK(hipMemcpyAsync(hits_rev, d_r, nr * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_f); (void)hipFree(d_r);
  return HAWK_OK;
}

int hawk_pam_scan_time(hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t reps, float* avg_ms,
                       uint64_t* scanned_positions) {
  if (!hs || !hs->has_meta || !avg_ms || !reps) return HAWK_E_INVALID;
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
static int reserve_cols(DevBuf (&b)[8], uint64_t cap, GuideCols* c) {
  int rc;
  const size_t sz[8] = {cap * 4, cap * 4, cap, cap * 8, cap * 8, cap, cap * 8, cap * 8 * HAWK_PLANES};
  for (int i = 0; i < 8; ++i) if ((rc = b[i].reserve(std::max<size_t>(sz[i], 16)))) return rc;
  c->hap = b[0].as<uint32_t>(); c->pos = b[1].as<uint32_t>(); c->strand = b[2].as<uint8_t>();
  c->start = b[3].as<int64_t>(); c->stop = b[4].as<int64_t>(); c->flags = b[5].as<uint8_t>();
  c->cfdon = b[6].as<double>(); c->win = b[7].as<uint64_t>(); c->cap = cap;
  return HAWK_OK;
}

int hawk_search(hawk_hapset* hs, const hawk_search_params* p, hawk_table** out, hawk_timing* timing) {
  if (!hs || !p || !out || !hs->has_meta) return HAWK_E_INVALID;
  if (p->score_cfdon && (p->right || !p->cfd_mm || !p->cfd_pam || p->pamlen < 2)) return HAWK_E_INVALID;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  ScanParams sp;
  int rc = make_scan_params(hs, p->pam_fwd, p->pam_rev, p->pamlen, p->guidelen, p->right, true, &sp);
  if (rc) return rc;
  const HapSetDev d = make_dev(hs);
  const uint64_t ntile = (uint64_t)hs->n_hap * sp.bph;
  if ((rc = hs->counts.reserve(ntile * 4)) || (rc = hs->offsets.reserve((ntile + 1) * 8)) ||
      (rc = hs->totals.reserve(sizeof(ScanTotals))) || (rc = hs->misc.reserve(512 * 8 + 64)) ||
      (rc = hs->cfd.reserve(336 * 8)) || (rc = hs->partial.reserve((ntile / 1024 + 2) * 8)))
    return rc;
  // hand-over lists (2 KB per tile): the count pass leaves each small tile's valid survivors for the emit pass.
  // HAWK_LIST_EMIT=0 keeps the recompute-everything emit pass (A/B measurements).
  static const bool list_emit = [] { const char* e = getenv("HAWK_LIST_EMIT"); return !(e && e[0] == '0'); }();
  uint32_t* d_lists = nullptr;
  if (list_emit) {
    if ((rc = hs->lists.reserve(ntile * HAWK_LIST_CAP * 4))) return rc;
    d_lists = hs->lists.as<uint32_t>();
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
  GuideParams gp;
  gp.pamlen = sp.pamlen; gp.guidelen = sp.guidelen; gp.right = sp.right; gp.L = sp.L;
  gp.score_cfdon = p->score_cfdon ? 1 : 0;
  gp.cfd_mm = hs->cfd.as<double>(); gp.cfd_pam = hs->cfd.as<double>() + 320; gp.bph = sp.bph;
  RefInfo ri;
  ri.index = hs->ref_index; ri.startp = hs->ref_startp;
  for (int s = 0; s < 2; ++s) {
    ri.lo[s] = 0; ri.hi[s] = 0;
    if (hs->ref_index >= 0) {  // same arithmetic as the kernel's phase A, for the REF haplotype
      const bool pamfirst = (sp.right != 0) != (s != 0);
      const int po = pamfirst ? 0 : sp.guidelen;
      const int haplen = (int)hs->hap_len[hs->ref_index];
      ri.lo[s] = std::max(hs->scan_start[hs->ref_index] - po, HAWK_PAD);
      ri.hi[s] = std::min(hs->scan_stop[hs->ref_index] - po, haplen - sp.L - HAWK_PAD + 1);
    }
  }
  GuideCols none = {};
  hipEvent_t* ev = ctx->ev;
  HIPCHK(hipEventRecord(ev[0], ctx->stream));
  hawk_launch_search(ctx->stream, 0, d, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), d_shards, nullptr, none, d_status, d_lists);
  HIPCHK(hipEventRecord(ev[1], ctx->stream));
  hawk_launch_mscan(ctx->stream, hs->counts.as<uint32_t>(), ntile, hs->partial.as<unsigned long long>(), d_shards,
                    hs->offsets.as<uint64_t>(), hs->totals.as<ScanTotals>());
  HIPCHK(hipEventRecord(ev[2], ctx->stream));
  HIPCHK(hipGetLastError());
  ScanTotals tot;
  static const bool count_only = [] { const char* e = getenv("HAWK_COUNT_ONLY"); return e && e[0] == '1'; }();
  GuideCols ca;
  int status = 0;
  uint64_t nrows = 0;
  bool emitted = false;
  if (hs->cols_cap && !count_only) {
    // Columns from an earlier search on this set are still reserved: launch the emit pass straight behind the offset
    // scan instead of waiting for the row count to cross PCIe (the kernels take their offsets from HBM and refuse to
    // write past the capacity).  If the table turns out larger, the normal path below runs after a reserve.
    if ((rc = reserve_cols(hs->colsA, hs->cols_cap, &ca))) return rc;
    HIPCHK(hipEventRecord(ev[3], ctx->stream));
    hawk_launch_search(ctx->stream, 1, d, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), d_shards,
                       hs->offsets.as<uint64_t>(), ca, d_status, d_lists, ev[5]);
    HIPCHK(hipEventRecord(ev[4], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&tot, hs->totals.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    nrows = tot.n_keep;
    emitted = nrows <= hs->cols_cap;
    if (!emitted) { status = 0; HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream)); }
  } else {
    HIPCHK(hipMemcpyAsync(&tot, hs->totals.p, sizeof(tot), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    nrows = tot.n_keep;
  }
  if (count_only) {  // measurement hook: time the count pass of an experimental build whose counts the emit pass cannot use
    if (getenv("HAWK_COUNT_VERBOSE")) fprintf(stderr, "[hawk] count pass: n_keep=%llu n_cand=%llu n_hits=%llu\n", (unsigned long long)tot.n_keep, (unsigned long long)tot.n_cand, (unsigned long long)tot.n_hits);
    nrows = 0;
  }
  if (!emitted) {
    const uint64_t want = std::max<uint64_t>(nrows, 1);
    if ((rc = reserve_cols(hs->colsA, std::max<uint64_t>(want, hs->cols_cap), &ca))) return rc;
    hs->cols_cap = ca.cap;
    HIPCHK(hipEventRecord(ev[3], ctx->stream));
    if (nrows) hawk_launch_search(ctx->stream, 1, d, sp, gp, ri, hs->d_tile_meta, hs->counts.as<uint32_t>(), d_shards,
                                  hs->offsets.as<uint64_t>(), ca, d_status, d_lists, ev[5]);
    HIPCHK(hipEventRecord(ev[4], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (timing) {
    memset(timing, 0, sizeof(*timing));
    (void)hipEventElapsedTime(&timing->count_ms, ev[0], ev[1]);
    (void)hipEventElapsedTime(&timing->offsets_ms, ev[1], ev[2]);
    (void)hipEventElapsedTime(&timing->emit_ms, ev[3], ev[4]);
    if (nrows && d_lists) (void)hipEventElapsedTime(&timing->emit_list_ms, ev[3], ev[5]);
    (void)hipEventElapsedTime(&timing->total_ms, ev[0], ev[4]);
    uint64_t pos = 0;
    for (uint32_t h = 0; h < hs->n_hap; ++h) pos += (uint64_t)std::max(0, hs->scan_stop[h] - hs->scan_start[h]);
    timing->scanned_positions = pos;
  }
  if (status) return status;
  hawk_table* t = new (std::nothrow) hawk_table();
  if (!t) return HAWK_E_INVALID;
  t->hs = hs; t->n_rows = nrows; t->n_cand = tot.n_cand; t->n_hits = tot.n_hits; t->cols = ca; t->cap = ca.cap;
  t->guidelen = p->guidelen; t->pamlen = p->pamlen; t->right = p->right ? 1 : 0; t->n_groups = 0; t->collapsed = false;
  *out = t;
  return HAWK_OK;
}

void hawk_table_destroy(hawk_table* t) { delete t; }  // columns live in the hapset's workspace

int hawk_table_counts(const hawk_table* t, uint64_t* n_rows, uint64_t* n_candidates, uint64_t* n_hits) {
  if (!t) return HAWK_E_INVALID;
  if (n_rows) *n_rows = t->n_rows;
  if (n_candidates) *n_candidates = t->n_cand;
  if (n_hits) *n_hits = t->n_hits;
  return HAWK_OK;
}

int hawk_table_download(hawk_table* t, uint32_t* hap, uint32_t* pos, uint8_t* strand, int64_t* start, int64_t* stop,
                        uint8_t* flags, double* cfdon, uint64_t* win) {
  if (!t) return HAWK_E_INVALID;
  hawk_ctx* ctx = t->hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows;
  if (!n) return HAWK_OK;
  const GuideCols& c = t->cols;
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

int hawk_table_device_columns(hawk_table* t, void** hap, void** pos, void** strand, void** start, void** stop,
                              void** flags, void** cfdon, void** win, uint64_t* win_plane_stride) {
  if (!t) return HAWK_E_INVALID;
  const GuideCols& c = t->cols;
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

int hawk_table_collapse(hawk_table* t, uint64_t* n_groups, float* kernel_ms) {
  if (!t || !n_groups) return HAWK_E_INVALID;
  hawk_hapset* hs = t->hs;
  hawk_ctx* ctx = hs->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint64_t n = t->n_rows;
  t->collapsed = false;
  *n_groups = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  if (n == 0) { t->n_groups = 0; t->collapsed = true; return HAWK_OK; }
  if (n > 0xffffffffull || hs->max_gen - hs->min_gen > 0xffffffffll) return HAWK_E_UNSUPPORTED;
  unsigned end_bit = 32;
  while (end_bit < 64 && ((uint64_t)(hs->max_gen - hs->min_gen) >> (end_bit - 32)) != 0) ++end_bit;
  const size_t temp_bytes = hawk_collapse_temp_bytes(n, end_bit);
  int rc;
  if ((rc = hs->ckeys.reserve(2 * n * 8)) || (rc = hs->cvals.reserve(2 * n * 4)) || (rc = hs->cflags.reserve(n * 4)) ||
      (rc = hs->cgidx.reserve(n * 4)) || (rc = hs->ctemp.reserve(temp_bytes + 16)) || (rc = hs->cgoff.reserve((n + 1) * 8)) ||
      (rc = hs->cgc.reserve(2 * n)) || (rc = hs->ccnt.reserve(16)) || (rc = hs->cfull.reserve(hawk_collapse_full_bytes(n))))
    return rc;
  unsigned long long cnt[2] = {0, 0};
  for (int attempt = 0; attempt < 4; ++attempt) {  // a new seed whenever two different rows collide in the hash bits
    HIPCHK(hipMemsetAsync(hs->ccnt.p, 0, 16, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    if (hawk_launch_collapse(ctx->stream, t->cols, hs->d_is_ref, n, (int)t->guidelen, (int)t->pamlen, (int)t->right, hs->min_gen,
                             end_bit, 0x9e3779b97f4a7c15ull * (uint64_t)(attempt + 1), hs->ctemp.p, temp_bytes, hs->ckeys.as<uint64_t>(),
                             hs->cvals.as<uint32_t>(), hs->cflags.as<uint32_t>(), hs->cgidx.as<uint32_t>(),
                             hs->ccnt.as<unsigned long long>(), hs->cgoff.as<uint64_t>(), hs->cgc.as<uint8_t>(),
                             hs->cgc.as<uint8_t>() + n, hs->cfull.p))
      return HAWK_E_HIP;
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(cnt, hs->ccnt.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]); *kernel_ms += ms; }
    if (cnt[0] == cnt[1]) break;
  }
  if (cnt[0] != cnt[1]) return HAWK_E_UNSUPPORTED;
  const uint64_t ng = cnt[1];
  HIPCHK(hipMemcpyAsync(hs->cgoff.as<uint64_t>() + ng, &n, 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  t->n_groups = ng; t->collapsed = true;
  *n_groups = ng;
  return HAWK_OK;
}

int hawk_table_collapse_download(hawk_table* t, uint32_t* perm, uint64_t* group_off, uint8_t* gc_num, uint8_t* gc_den) {
  if (!t || !t->collapsed) return HAWK_E_INVALID;
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

// ---------------------------------------------------------------------------- K7 off-targets
int hawk_genome_finalize(hawk_hapset* rows) {
  if (!rows) return HAWK_E_INVALID;
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
  HIPCHK(hipMalloc(&d_wt, n * len)); HIPCHK(hipMalloc(&d_sg, n * len)); HIPCHK(hipMalloc(&d_p, n * 2));
  HIPCHK(hipMalloc(&d_tab, 336 * 8)); HIPCHK(hipMalloc(&d_out, n * 8)); HIPCHK(hipMalloc(&d_status, 4));
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
  (void)hipFree(d_wt); (void)hipFree(d_sg); (void)hipFree(d_p); (void)hipFree(d_tab); (void)hipFree(d_out); (void)hipFree(d_status);
  return status;
}

// ---------------------------------------------------------------------------- f1 haplotype expansion
int hawk_hapset_expand(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                       const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len,
                       uint32_t n_hap, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o,
                       const uint32_t* hap_len, hawk_hapset** out, uint64_t* hash_out, float* kernel_ms) {
  if (!ref_set || !out || !n_hap || !hv_off || !hap_len || (n_var && (!v_r0 || !v_span || !v_alt_off || !v_alt_len || !alt_codes)))
    return HAWK_E_INVALID;
  hawk_ctx* ctx = ref_set->ctx;
  const uint32_t ref_len = ref_set->hap_len[0];
  const uint64_t ncar = hv_off[n_hap];
  if (ncar && (!hv_idx || !hv_o)) return HAWK_E_INVALID;
  // validate everything the kernel will index with, on the host
  for (uint32_t i = 0; i < n_var; ++i) {
    if ((uint64_t)v_r0[i] + v_span[i] > ref_len || v_span[i] == 0 || v_alt_len[i] == 0) return HAWK_E_INVALID;
    if ((uint64_t)v_alt_off[i] + v_alt_len[i] > alt_codes_len) return HAWK_E_INVALID;
    if (i && v_r0[i] < v_r0[i - 1]) return HAWK_E_INVALID;  // sorted by position (alleles of one site may share it)
  }
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
  }
  hawk_hapset* hs = nullptr;
  int rc = hawk_hapset_create(ctx, n_hap, hap_len, &hs);
  if (rc) return rc;
  uint32_t *d_r0 = nullptr, *d_span = nullptr, *d_ao = nullptr, *d_al = nullptr, *d_idx = nullptr;
  uint8_t* d_codes = nullptr; uint64_t* d_off = nullptr; int32_t* d_o = nullptr; unsigned long long* d_hash = nullptr;
  int32_t* d_wk0 = nullptr; uint32_t* d_wn = nullptr;
  const size_t nwg = (size_t)n_hap * ((hs->S + HAWK_BLOCK - 1) / HAWK_BLOCK);
  const size_t nv = std::max<size_t>(n_var, 1), nc = std::max<size_t>(ncar, 1);
  HIPCHK(hipMalloc(&d_r0, nv * 4)); HIPCHK(hipMalloc(&d_span, nv * 4)); HIPCHK(hipMalloc(&d_ao, nv * 4)); HIPCHK(hipMalloc(&d_al, nv * 4));
  HIPCHK(hipMalloc(&d_codes, std::max<size_t>(alt_codes_len, 1))); HIPCHK(hipMalloc(&d_off, (size_t)(n_hap + 1) * 8));
  HIPCHK(hipMalloc(&d_idx, nc * 4)); HIPCHK(hipMalloc(&d_o, nc * 4)); HIPCHK(hipMalloc(&d_hash, (size_t)n_hap * 16));
  HIPCHK(hipMalloc(&d_wk0, nwg * 4)); HIPCHK(hipMalloc(&d_wn, nwg * 4));
  hipStream_t st = ctx->stream;
  if (n_var) {
    HIPCHK(hipMemcpyAsync(d_r0, v_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_span, v_span, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ao, v_alt_off, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_al, v_alt_len, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_codes, alt_codes, alt_codes_len, hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(d_off, hv_off, (size_t)(n_hap + 1) * 8, hipMemcpyHostToDevice, st));
  if (ncar) {
    HIPCHK(hipMemcpyAsync(d_idx, hv_idx, ncar * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_o, hv_o, ncar * 4, hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemsetAsync(d_hash, 0, (size_t)n_hap * 16, st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hawk_launch_hx_build(st, ref_set->plane, d_r0, d_span, d_ao, d_al, d_codes, d_off, d_idx, d_o, hs->d_hap_len, n_hap, hs->S, hs->plane, d_wk0, d_wn);
  hawk_launch_hx_hash(st, hs->plane, n_hap, hs->S, d_hash);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  HIPCHK(hipGetLastError());
  if (hash_out) HIPCHK(hipMemcpyAsync(hash_out, d_hash, (size_t)n_hap * 16, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
  (void)hipFree(d_r0); (void)hipFree(d_span); (void)hipFree(d_ao); (void)hipFree(d_al); (void)hipFree(d_codes); (void)hipFree(d_off);
  (void)hipFree(d_idx); (void)hipFree(d_o); (void)hipFree(d_hash); (void)hipFree(d_wk0); (void)hipFree(d_wn);
  *out = hs;
  return HAWK_OK;
}

// ---------------------------------------------------------------------------- f3: VCF genotypes
struct hawk_gt {
  hawk_ctx* ctx;
  uint64_t n_lines;
  uint32_t n_samples, n_var;
  uint8_t* d_codes;   // [n_lines][2 * n_samples]
  uint8_t* d_flags;   // [n_lines]
  uint64_t n_entries; // carried-variant entries over all columns (valid after hawk_gt_lists)
  uint64_t* d_col_off; uint32_t* d_idx; int32_t* d_o; int64_t* d_delta;
};

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
  HIPCHK(hipMalloc(&g->d_codes, ncode)); HIPCHK(hipMalloc(&g->d_flags, std::max<size_t>(n_lines, 1)));
  if (n_lines) {
    uint8_t* d_text = nullptr; uint64_t *d_lo = nullptr, *d_go = nullptr;
    HIPCHK(hipMalloc(&d_text, text_len)); HIPCHK(hipMalloc(&d_lo, (n_lines + 1) * 8)); HIPCHK(hipMalloc(&d_go, n_lines * 8));
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
    (void)hipFree(d_text); (void)hipFree(d_lo); (void)hipFree(d_go);
  }
  *out = g;
  return HAWK_OK;
}

void hawk_gt_destroy(hawk_gt* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipFree(g->d_codes); (void)hipFree(g->d_flags);
  if (g->d_col_off) (void)hipFree(g->d_col_off);
  if (g->d_idx) (void)hipFree(g->d_idx);
  if (g->d_o) (void)hipFree(g->d_o);
  if (g->d_delta) (void)hipFree(g->d_delta);
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
  if (g->d_col_off) { (void)hipFree(g->d_col_off); g->d_col_off = nullptr; }
  if (g->d_idx) { (void)hipFree(g->d_idx); g->d_idx = nullptr; }
  if (g->d_o) { (void)hipFree(g->d_o); g->d_o = nullptr; }
  if (g->d_delta) { (void)hipFree(g->d_delta); g->d_delta = nullptr; }
  g->n_var = n_var; g->n_entries = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  std::vector<uint64_t> off(n_cols + 1, 0);
  if (n_var == 0) {
    memcpy(col_off, off.data(), (n_cols + 1) * 8);
    if (col_delta) memset(col_delta, 0, (size_t)n_cols * 8);
    return HAWK_OK;
  }
  uint32_t *d_vl = nullptr, *d_cnt = nullptr; uint8_t* d_va = nullptr; int32_t *d_r0 = nullptr, *d_ch = nullptr;
  unsigned long long* d_bal = nullptr;
  HIPCHK(hipMalloc(&d_vl, (size_t)n_var * 4)); HIPCHK(hipMalloc(&d_va, n_var)); HIPCHK(hipMalloc(&d_r0, (size_t)n_var * 4));
  HIPCHK(hipMalloc(&d_ch, (size_t)n_var * 4)); HIPCHK(hipMalloc(&d_cnt, (size_t)n_cols * 4));
  HIPCHK(hipMalloc(&d_bal, (size_t)n_cols * n_chunk * 8));
  HIPCHK(hipMalloc(&g->d_col_off, (size_t)(n_cols + 1) * 8)); HIPCHK(hipMalloc(&g->d_delta, (size_t)n_cols * 8));
  HIPCHK(hipMemcpyAsync(d_vl, var_line, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_va, var_allele, n_var, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r0, var_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ch, var_chain, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hawk_launch_gt_count(st, g->d_codes, n_cols, d_vl, d_va, n_var, d_bal, d_cnt);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  std::vector<uint32_t> cnt(n_cols);
  HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n_cols * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (uint32_t c = 0; c < n_cols; ++c) off[c + 1] = off[c] + cnt[c];  // 2 * n_samples values: a host prefix sum
  const uint64_t ne = off[n_cols];
  HIPCHK(hipMalloc(&g->d_idx, std::max<size_t>(ne, 1) * 4)); HIPCHK(hipMalloc(&g->d_o, std::max<size_t>(ne, 1) * 4));
  HIPCHK(hipMemcpyAsync(g->d_col_off, off.data(), (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  hawk_launch_gt_fill(st, n_cols, d_r0, d_ch, n_var, d_bal, g->d_col_off, g->d_idx, g->d_o, g->d_delta);
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipGetLastError());
  if (col_delta) HIPCHK(hipMemcpyAsync(col_delta, g->d_delta, (size_t)n_cols * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (kernel_ms) {
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]); (void)hipEventElapsedTime(&b, ctx->ev[2], ctx->ev[3]);
    *kernel_ms = a + b;
  }
  memcpy(col_off, off.data(), (size_t)(n_cols + 1) * 8);
  g->n_entries = ne;
  (void)hipFree(d_vl); (void)hipFree(d_va); (void)hipFree(d_r0); (void)hipFree(d_ch); (void)hipFree(d_cnt); (void)hipFree(d_bal);
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
  HIPCHK(hipMalloc(&d_s, n * 34)); HIPCHK(hipMalloc(&d_w, nw * 4)); HIPCHK(hipMalloc(&d_o, n * 4)); HIPCHK(hipMalloc(&d_status, 4));
  HIPCHK(hipMemcpyAsync(d_s, seqs34, n * 34, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_w, weights, nw * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
  hawk_launch_deepcpf1(ctx->stream, d_s, n, d_w, d_o, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_s); (void)hipFree(d_w); (void)hipFree(d_o); (void)hipFree(d_status);
  return status;
}

// ---------------------------------------------------------------------------- K5 Azimuth
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
  HIPCHK(hipMalloc(&d_s, n * 30)); HIPCHK(hipMalloc(&d_off, (nt + 1) * 4)); HIPCHK(hipMalloc(&d_f, nn * 4));
  HIPCHK(hipMalloc(&d_l, nn * 4)); HIPCHK(hipMalloc(&d_r, nn * 4)); HIPCHK(hipMalloc(&d_th, nn * 8)); HIPCHK(hipMalloc(&d_v, nn * 8));
  HIPCHK(hipMalloc(&d_o, n * 8)); HIPCHK(hipMalloc(&d_status, 4));
  if (feats_out) HIPCHK(hipMalloc(&d_fo, n * 627 * 8));
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
  (void)hipFree(d_s); (void)hipFree(d_off); (void)hipFree(d_f); (void)hipFree(d_l); (void)hipFree(d_r); (void)hipFree(d_th);
  (void)hipFree(d_v); (void)hipFree(d_o); (void)hipFree(d_status); if (d_fo) (void)hipFree(d_fo);
  return status;
}

}  // extern "C"
