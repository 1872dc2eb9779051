// hawk_api_hapset.hip - C ABI: haplotype sets (create / pack / metadata / planes) and the raw PAM scan (include/hawk.h)
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

extern "C" {

// ---------------------------------------------------------------------------- hapset
extern "C++" int hapset_create_impl(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, bool zero_planes, hawk_hapset** out, bool alloc_planes) {
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
extern "C++" int meta_build(uint32_t n, const std::vector<uint32_t>& hap_len, uint32_t bph, const uint8_t* is_ref, const int32_t* scan_start,
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

extern "C++" HapSetDev make_dev(const hawk_hapset* hs) {
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

extern "C++" int make_scan_params(const hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t guidelen,
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

}  // extern "C"
