// hawk_api_search.hip - C ABI: the fused search (hawk_search) and the guide table it leaves in HBM
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
    // (cs_res, 32 bytes per distinct cluster: {rows per strand, hits, candidates} {first template row, REF's hits before / behind it})
    if ((rc = hs->cs_res.reserve((size_t)std::max<uint32_t>(cl.n_uniq, 1) * 32)) || (rc = hs->cs_tbase.reserve((size_t)std::max<uint32_t>(cl.n_uniq, 1) * 4)) ||
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
      hawk_launch_cs_count(ctx->stream, d, va, cd, sp, hs->cs_res.p, d_tcount, tcap, d_counts_v, hs->cs_icnt.as<uint32_t>(),
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
  GuideCols ca, tc;
  int status = 0;
  uint64_t nrows = 0;
  bool emitted = false;
  auto template_overflow = [&](uint64_t tc_used) {  // the rerun reserves what this search asked for (+ 1/8), at most the plan's bound
    hs->cs_tcap = std::min<uint64_t>(vx->cl.slots, tc_used + tc_used / 8 + 64);
  };
  if (table_cap) {
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

}  // extern "C"
