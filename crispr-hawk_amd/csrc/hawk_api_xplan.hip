// hawk_api_xplan.hip - C ABI: expansion plans, plan views and the cluster dictionary (f1)
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
  DevBuf* bufs[] = {&x->recs, &x->heads, &x->tiles, &x->codes, &x->off, &x->hlen, &x->hash,
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
  if (!rc) rc = x->heads.reserve(nc * 16);
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
    hawk_launch_hx_heads(st, x->recs.p, d_idx ? d_idx : t_idx.as<uint32_t>(), ncar, x->heads.p);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  for (auto* b : temps) b->release();
  if (e != hipSuccess) {
    snprintf(hawk_hip_err_buf(), 256, "hawk_xplan_create: %s", hipGetErrorString(e));
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
    snprintf(hawk_hip_err_buf(), 256, "hawk_xplan_run: %s", hipGetErrorString(e));
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
  cl.built = true; cl.usable = false; cl.status = 0; cl.n_inst = cl.n_uniq = cl.n_real = 0; cl.slots = 0; cl.build_ms = 0.f;
  hawk_ctx* ctx = x->ctx;
  hipStream_t st = ctx->stream;
  const uint32_t n = x->n_hap;
  if (x->ncar == 0 || x->ncar >= (1ull << 32) - 2 || n < 2) { cl.status = 4; return HAWK_OK; }
  PoolScope tmp;
  // Sizes first - nothing below waits for a count from the device.  An instance starts at a record or closes a row, so records + rows
  // bounds their number: the instance arrays, the list and the grids of the passes are sized by the bound, the passes read the true
  // counts on the device, and the host learns them together with the number of distinct clusters, once, at the end.
  const uint32_t n_var = x->n_var;
  const uint32_t ch_bound = hawk_cl_chunk_bound(x->ncar, n);  // the rows' records in chunks (hawk_csearch.hip)
  const uint32_t inst_bound = (uint32_t)x->ncar + n;
  const uint32_t bm_words = n_var / 32 + 1;
  // The table of the clusters that are more than their variant: at least two slots per listed instance would always do, but the
  // distinct clusters are a small fraction of the instances and clearing 32 bytes x 2^25 slots costs more than every kernel of this
  // build - so the first attempt takes two slots per distinct cluster EXPECTED (the last build's count, else an eighth of the
  // instances), gives up after 64 probes (status bit 8), and the pass is repeated with the full size
  uint32_t tsize = 1024;
  while (tsize < 2u * inst_bound && tsize < (1u << 30)) tsize <<= 1;
  uint32_t tsmall = std::min<uint32_t>(1u << 16, tsize);
  { const uint64_t expect = cl.last_uniq ? (uint64_t)cl.last_uniq * 2 : (uint64_t)inst_bound / 8; while (tsmall < expect && tsmall < tsize) tsmall <<= 1; }
  const uint32_t u_bound0 = n_var + std::min<uint32_t>(tsmall, inst_bound);  // cluster numbers of the first attempt: the variants, then what its table can hold
  // everything that starts from zero, in ONE block cleared by ONE memset: status, counters, the chunks' counts, the variants' bitmap
  // and describers, the clusters' template-row bounds, the first attempt's table
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t z_status = 0, z_counters = 256, z_results = 512, z_cnt = 768,
               z_claim = z_cnt + up(((size_t)ch_bound + 4) * 8), z_vdesc = z_claim + up((size_t)bm_words * 4),
               z_span2 = z_vdesc + up((size_t)std::max<uint32_t>(n_var, 1) * 8), z_tab = z_span2 + up((size_t)u_bound0 * 4),
               z_end = z_tab + (size_t)tsmall * hawk_cl_slot_bytes();
  char* d_zero;
  TEMPCHK(tmp, &d_zero, z_end);
  uint32_t* const d_status = reinterpret_cast<uint32_t*>(d_zero + z_status);
  uint32_t* const d_counters = reinterpret_cast<uint32_t*>(d_zero + z_counters);
  unsigned long long* const d_results = reinterpret_cast<unsigned long long*>(d_zero + z_results);
  uint32_t* const d_cnt = reinterpret_cast<uint32_t*>(d_zero + z_cnt);
  uint32_t* const d_lcnt = d_cnt + ((ch_bound + 3) & ~3u);  // (16-byte aligned: k_scan2_u32 loads four counts at a time)
  uint32_t* const d_claim = reinterpret_cast<uint32_t*>(d_zero + z_claim);
  void* const d_vdesc = d_zero + z_vdesc;
  uint32_t *d_ch_off, *d_ch_row, *d_base, *d_lbase, *d_state;
  void* d_list;
  TEMPCHK(tmp, &d_ch_off, (size_t)(n + 1) * 4);
  TEMPCHK(tmp, &d_ch_row, (size_t)ch_bound * 4);
  TEMPCHK(tmp, &d_base, (size_t)(ch_bound + 1) * 4);
  TEMPCHK(tmp, &d_lbase, (size_t)(ch_bound + 1) * 4);
  TEMPCHK(tmp, &d_list, (size_t)inst_bound * hawk_cl_listed_bytes());
  TEMPCHK(tmp, &d_state, (size_t)inst_bound * 4);
  int rc;
  if ((rc = cl.inst_uid.reserve((size_t)inst_bound * 4)) || (rc = cl.inst_o.reserve((size_t)inst_bound * 4)) ||
      (rc = cl.inst_row.reserve((size_t)inst_bound * 4)) || (rc = cl.inst_pa.reserve((size_t)inst_bound * 4)) ||
      (rc = cl.inst_rb.reserve((size_t)inst_bound * 4)))
    return rc;
  uint32_t* const t_uid = cl.inst_uid.as<uint32_t>();   // the instances stay in the order they are built in: (row, position)
  uint32_t* const t_row = cl.inst_row.as<uint32_t>();
  int32_t* const t_o = cl.inst_o.as<int32_t>();
  int32_t* const t_pa = cl.inst_pa.as<int32_t>();
  int32_t* const t_rb = cl.inst_rb.as<int32_t>();
  HIPCHK(hipEventRecord(ctx->ev[8], st));
  HIPCHK(hipMemsetAsync(d_zero, 0, z_end, st));
  // chunks per row -> their offsets and rows, then the instances every chunk opens and how many of them go on the list of the
  // clusters that are more than their variant (the number of chunks is only known on the device - rows that scan nothing have
  // none: the count pass and its scan run over the bound; chunks beyond the last one open nothing)
  hawk_launch_cl_chunks(st, x->off.as<uint64_t>(), x->m_is_ref.as<uint8_t>(), x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), n, d_ch_off, d_ch_row);
  hawk_launch_cl_count(st, x->heads.p, x->off.as<uint64_t>(), x->hlen.as<uint32_t>(), x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), d_ch_off, d_ch_row, n,
                       n_var, ch_bound, d_cnt, d_lcnt);
  hawk_launch_scan2_u32(st, d_cnt, d_lcnt, ch_bound, d_base, d_lbase);
  const uint32_t* const d_n_inst = d_base + ch_bound;
  const uint32_t* const d_n_list = d_lbase + ch_bound;
  // One pass = cut the rows (a one-record shareable instance is its variant; the rest goes on the list), the listed instances through
  // the table, the distinct clusters' descriptions, the listed instances that share a cluster - queued without a read-back in
  // between.  The host reads the counts once, at the end, and repeats the pass if the small table gave up.
  unsigned long long res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t status = 0, n_inst = 0, n_uniq = 0, n_real = 0;
  auto pass = [&](uint32_t tsz, uint32_t max_probe, uint32_t fail_bit, bool first) -> int {
    const uint32_t u_bound = n_var + std::min<uint32_t>(tsz, inst_bound);  // the variants, then what the table can hold
    uint32_t* d_span2 = reinterpret_cast<uint32_t*>(d_zero + z_span2);
    void* d_tab = d_zero + z_tab;
    if (!first) {  // the repeat: its own, larger table and bounds; everything cleared again
      TEMPCHK(tmp, &d_span2, (size_t)u_bound * 4);
      TEMPCHK(tmp, &d_tab, (size_t)tsz * hawk_cl_slot_bytes());
      HIPCHK(hipMemsetAsync(d_tab, 0, (size_t)tsz * hawk_cl_slot_bytes(), st));
      HIPCHK(hipMemsetAsync(d_span2, 0, (size_t)u_bound * 4, st));
      HIPCHK(hipMemsetAsync(d_zero + z_counters, 0, z_cnt - z_counters, st));               // counters, results
      HIPCHK(hipMemsetAsync(d_zero + z_claim, 0, z_span2 - z_claim, st));                   // the variants' bitmap and describers
    }
    int rc2;
    if ((rc2 = cl.u_rec.reserve((size_t)u_bound * 4)) || (rc2 = cl.u_n.reserve((size_t)u_bound * 4)) || (rc2 = cl.u_row.reserve((size_t)u_bound * 4)) ||
        (rc2 = cl.u_o.reserve((size_t)u_bound * 4)) || (rc2 = cl.u_seg.reserve((size_t)u_bound * 4)))
      return rc2;
    hawk_launch_cl_fill(st, x->heads.p, x->off.as<uint64_t>(), x->hlen.as<uint32_t>(), x->m_ss.as<int32_t>(), x->m_se.as<int32_t>(), n, d_ch_off, d_ch_row,
                        ch_bound, d_base, d_lbase, t_o, t_row, t_pa, t_rb, t_uid, d_vdesc, d_claim, n_var, d_list, d_status);
    hawk_launch_cl_finish(st, inst_bound, d_n_inst, d_n_list, d_counters, d_results, n_var, u_bound, d_tab, tsz - 1, max_probe, fail_bit, d_list, d_state, d_vdesc, x->heads.p,
                          t_uid, t_row, x->m_seg_off.as<uint32_t>(), x->m_seg_rel.as<uint32_t>(), cl.u_rec.as<uint32_t>(), cl.u_n.as<uint32_t>(),
                          cl.u_row.as<uint32_t>(), cl.u_o.as<int32_t>(), cl.u_seg.as<uint32_t>(), d_span2, d_status);
    HIPCHK(hipMemcpyAsync(res, d_results, 64, hipMemcpyDeviceToHost, st));  // {instances, the table's clusters, the variants that are clusters, template rows, status}
    HIPCHK(hipEventRecord(ctx->ev[9], st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    n_inst = (uint32_t)res[0]; status = (uint32_t)res[4];
    n_uniq = std::min<uint32_t>(n_var + (uint32_t)res[1], u_bound);  // the range of the numbers
    n_real = (uint32_t)(res[1] + res[2]);
    return HAWK_OK;
  };
  if ((rc = pass(tsmall, tsmall < tsize ? 64u : 0xffffffffu, tsmall < tsize ? 8u : 2u, true))) return rc;
  if (status & 8u) {  // the small table filled up: once more with two slots per instance
    status &= ~8u;
    HIPCHK(hipMemcpyAsync(d_status, &status, 4, hipMemcpyHostToDevice, st));
    if ((rc = pass(tsize, 0xffffffffu, 2u, false))) return rc;
  }
  if (n_inst == 0) { cl.status = 4; return HAWK_OK; }
  if (n_inst > inst_bound) { snprintf(hawk_hip_err_buf(), 256, "hawk_xplan_view: instance count beyond its bound"); return HAWK_E_HIP; }
  cl.n_inst = n_inst; cl.n_uniq = n_uniq; cl.n_real = n_real; cl.last_uniq = (uint32_t)res[1];
  (void)hipEventElapsedTime(&cl.build_ms, ctx->ev[8], ctx->ev[9]);
  cl.slots = n_uniq ? res[3] : 0;
  cl.status = status;
  // worth it when clusters are shared (the template rows are extra traffic otherwise) and the templates fit a sane budget
  // (HAWK_CLUSTER_MAX_SLOTS template rows, default 2^27 = 10 GB; HAWK_CLUSTER_MIN_SHARE instances per distinct cluster, default 3:
  // read per call so that tests can send small panels down this path)
  const char* e1 = getenv("HAWK_CLUSTER_MAX_SLOTS");
  const char* e2 = getenv("HAWK_CLUSTER_MIN_SHARE");
  const uint64_t max_slots = e1 ? strtoull(e1, nullptr, 10) : (1ull << 27);
  const double min_share = e2 ? atof(e2) : 3.0;
  if (!status && (cl.slots > max_slots || (double)n_inst < min_share * (double)std::max<uint32_t>(n_real, 1))) cl.status = 4;
  cl.usable = cl.status == 0;
  return HAWK_OK;
}

int hawk_xplan_cluster_stats(const hawk_xplan* x, uint32_t* usable, uint32_t* n_instances, uint32_t* n_distinct, uint64_t* template_slots,
                             float* build_ms, uint32_t* status) {
  if (!x) return HAWK_E_INVALID;
  if (usable) *usable = x->cl.built && x->cl.usable ? 1u : 0u;
  if (n_instances) *n_instances = x->cl.n_inst;
  if (n_distinct) *n_distinct = x->cl.n_real;
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
      snprintf(hawk_hip_err_buf(), 256, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
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

}  // extern "C"
