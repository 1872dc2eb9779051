// hawk_api_collapse.hip - C ABI: which rows the guide report merges (hawk_table_collapse*) and the group export
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
  // HAWK_COLLAPSE_EXACT=1 forces from the start.  The verification cannot be switched off in the product library.
  const char* ex = getenv("HAWK_COLLAPSE_EXACT");
#ifdef HAWK_TEST_HOOKS  // libhawk_hip_hooks.so only (tests/hooks_collapse_check.py): a 4-bit hash so that the verify pass HAS collisions
  const char* vf = getenv("HAWK_COLLAPSE_VERIFY");  // to catch, and a way to look at the grouping without it
  const char* wk = getenv("HAWK_COLLAPSE_WEAK_HASH");
  const bool verify = !(vf && vf[0] == '0'), weak = wk && wk[0] == '1';
#else
  const bool verify = true, weak = false;
#endif
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
  // everything the rest of this call reserves in the set's collapse workspace is reserved HERE, for the full table, before the
  // mini collapse fills it: a reservation after k_cc_gidm has been queued could hand the buffers it reads back to the pool
  {
    const size_t tb_full = hawk_collapse_expand_temp_bytes(n);
    if ((rc = hs->ckeys.reserve(std::max<size_t>(2 * nm * 8, 2 * n * 4))) || (rc = hs->cvals.reserve(2 * n * 4)) ||
        (rc = hs->cgoff.reserve((nm + 1) * 8)) || (rc = hs->cgc.reserve(2 * n)) || (rc = hs->ctemp.reserve(tb_full + 16)) ||
        (rc = hs->cflags.reserve(n * 4)))
      return rc;
  }
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


}  // extern "C"
