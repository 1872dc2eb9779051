// hawk_host.h — host-side objects behind the opaque handles of include/hawk.h, shared by the C-ABI
// translation units (hawk_api.hip, hawk_comm.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <memory>
#include <vector>

#include "../../include/hawk.h"
#include "hawk_device.h"

#define HAWK_MAX_DEVICES 16
char* hawk_hip_err_buf();  // thread-local text of the last HIP failure (hawk_last_hip_error)

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      snprintf(hawk_hip_err_buf(), 256, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return HAWK_E_HIP;                                                                     \
    }                                                                                        \
  } while (0)

struct hawk_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev[10];
  void* pinned = nullptr;  // 256 B of page-locked host memory: the per-search totals + status come back in one copy
};

// Caching device allocator, one per device: a freed block goes to a free list and is handed to the next request of a
// similar size, so a per-tile loop (expand -> search -> collapse, tile after tile) allocates its planes, columns and
// workspaces once.  Everything in this library runs on one stream per context, so stream order makes the reuse safe.
int hawk_pool_alloc(void** p, size_t bytes);   // HAWK_OK / HAWK_E_HIP (after releasing the cache and retrying once)
void hawk_pool_free(void* p);
void hawk_pool_trim();                          // hipFree every cached block of the current device
#define POOLCHK(ptr, bytes)                                                      \
  do {                                                                           \
    int rc_ = hawk_pool_alloc(reinterpret_cast<void**>(ptr), (bytes));           \
    if (rc_) return rc_;                                                         \
  } while (0)

// Temporaries of one C-ABI call: whatever TEMPCHK allocated goes back to the pool when the call returns, on the error paths
// too (the frees used to sit at the end of the success path only).  Blocks return to the pool, not to hipFree: work still
// queued on the context's stream keeps them valid, and the next user is on the same stream.
struct PoolScope {
  std::vector<void*> held;
  ~PoolScope() { for (void* q : held) hawk_pool_free(q); }
  int alloc(void** out, size_t bytes) {
    int rc = hawk_pool_alloc(out, bytes);
    if (!rc) held.push_back(*out);
    return rc;
  }
};
#define TEMPCHK(scope, ptr, bytes)                                               \
  do {                                                                           \
    int rc_ = (scope).alloc(reinterpret_cast<void**>(ptr), (bytes));             \
    if (rc_) return rc_;                                                         \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return HAWK_OK;
    if (p) { hawk_pool_free(p); p = nullptr; bytes = 0; }
    size_t want = need + need / 8 + 256;
    int rc = hawk_pool_alloc(&p, want);
    if (rc) return rc;
    bytes = want;
    return HAWK_OK;
  }
  void release() { if (p) hawk_pool_free(p); p = nullptr; bytes = 0; }
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
  uint32_t n_ref_rows = 0;           // rows flagged REF (their tiles take one emit work-list entry per 512 survivors)
  bool has_meta;
  bool has_partner = false;          // hawk_hapset_set_ref_partner_range
  int32_t partner_start = 0, partner_stop = 0;
  uint32_t bph;            // workgroups (tiles of 1024 words) per haplotype row
  TileMeta* d_tile_meta;   // [n_hap * bph] per-tile record (haplotype scalars + first position-map segment)
  int64_t ref_startp;
  int64_t min_gen, max_gen;  // range of genomic positions the position maps reach (collapse sort key)
  uint64_t cols_cap = 0;      // rows the guide-table columns currently hold (0: never reserved)
  uint64_t cols_gen = 0;      // bumped by every hawk_search: a hawk_table of an older generation is stale
  uint64_t collapse_gen = 0;  // generation of the table whose collapse results sit in the c* buffers
  std::vector<double> cfd_host;  // the CFD tables resident in `cfd`
  // workspace reused across searches
  DevBuf keepF, keepR, counts, offsets, totals, misc, cfd, partial, sites, hits, guides, lists;
  DevBuf ckeys, cvals, cflags, cgidx, ctemp, cgoff, cgc, ccnt, cfull;  // hawk_table_collapse
  DevBuf ctable, cocc, cdense, cgkey, cgslot;                          // ... its hash-table path
  uint64_t last_groups = 0;   // groups of the last collapse on this set (sizes the table of the next one)
  std::shared_ptr<uint64_t> plan_groups;  // ... shared with the expansion plan the set came from: the next run of the plan starts from it
  DevBuf otoff, otcode, otid, othit;  // hawk_offtarget_scan: bucketed guides, gathered hit sites
  DevBuf big;                 // tiles whose rows exceed the hand-over list (k_search_emit's work list)
  DevBuf refbits;             // REF's candidate-window bitmaps, one per strand (k_ref_bits)
  bool refbits_valid = false;
  uint64_t refbits_key[6] = {0, 0, 0, 0, 0, 0};
  // a VIEW of an expansion plan (hawk_xplan_view): rows and metadata of the plan's haplotypes without their planes -
  // plane[] points at the plan's REF planes (row 0 only) and hawk_search runs from the plan's records (hawk_vsearch.hip)
  const struct hawk_xplan* vplan = nullptr;
  DevBuf vcnt0;               // per tile: rows of strand 0 (k_vsearch<0> -> k_vsearch<1>)
  DevBuf refhp;               // REF's PAM hits + prefix counts per strand (k_ref_hits), keyed like refbits
  uint64_t cs_tcap = 0;       // template rows a search of this view may need (raised to the plan's bound after an overflow)
  DevBuf cs_res, cs_tbase, cs_trows, cs_itb, cs_icnt;  // per distinct cluster 32 B {rows per strand, hits, candidates} {first template row, REF hits before / behind}, first template row; template rows; per instance its first template row
  DevBuf cmini[8], cm_gid;   // hawk_table_collapse of such a table: REF's rows + the template rows as a table of their own, their groups  // the cluster search of a view: per distinct cluster {rows per strand, hits, candidates}, first template row; template rows
  DevBuf colsA[8];
  DevBuf rowsA;               // packed rows of a cluster-searched table (colsA then stages REF's rows only)
  uint64_t rows_cap = 0;
  DevBuf crep[8];  // hawk_table_collapse_export: one representative row per group
};

// An expansion plan keeps everything hawk_hapset_expand needs in HBM - the variant table, the carried-variant lists, the
// per-workgroup variant ranges and (after hawk_xplan_set_meta) the metadata of the rows it produces - so that running
// it is device work only: the per-tile loop of a whole-contig search re-expands its tiles without touching the host.
struct hawk_xplan {
  hawk_ctx* ctx;
  uint32_t n_var, n_hap, ref_len;
  uint64_t ncar;
  std::vector<uint32_t> hap_len;
  uint32_t* ref_plane[4];  // the REF region's code planes, copied: the plan does not depend on the life of ref_set
  uint32_t ref_S;
  DevBuf ref5[HAWK_PLANES];  // the same planes (+ an all-zero V plane) at the ROWS' stride S: row 0 of a view (hawk_xplan_view)
  DevBuf recs, tiles, codes, off, hlen, hash;  // 32 B per carried variant, 16 B per (row, tile): hawk_expand.hip
  DevBuf heads;                               // the records' first 16 bytes {o, rs, alt_len, alt_off} once more: what the dictionary passes stream
  // metadata of the produced rows (hawk_xplan_set_meta)
  bool has_meta;
  std::vector<int32_t> scan_start, scan_stop;
  DevBuf m_is_ref, m_ss, m_se, m_seg_off, m_seg_rel, m_seg_gen, m_tile;
  uint32_t nseg, bph, S;
  int32_t ref_index;
  int64_t ref_startp, min_gen, max_gen;
  bool has_partner = false;
  int32_t partner_start = 0, partner_stop = 0;
  uint32_t n_ref_rows = 0;
  std::vector<int64_t> rev0, rev1;  // hawk_xplan_create_gt: posmap_rev of every row at the two positions asked for
  std::shared_ptr<uint64_t> groups = std::make_shared<uint64_t>(0);  // groups the last collapse of a set of this plan found
  // the cluster dictionary (hawk_csearch.hip), built by the first hawk_xplan_view after the metadata is set
  struct {
    bool built = false, usable = false;
    uint32_t n_inst = 0, n_uniq = 0;  // n_uniq: the RANGE of the distinct clusters' numbers (variants first - with holes - then the table's)
    uint32_t n_real = 0;              // how many distinct clusters there are
    uint32_t last_uniq = 0;  // distinct clusters of the previous build (sizes the first hash table of the next)
    uint64_t slots = 0;     // bound on the template rows of a search: window starts x 2 strands over all distinct clusters
    uint32_t status = 0;    // why it is not usable: 1 a chain of > 4096 records, 2 hash collision, 4 too large / too little sharing
    float build_ms = 0.f;
    DevBuf inst_uid, inst_o, inst_row, inst_pa, inst_rb, u_rec, u_n, u_row, u_o, u_seg;
  } cl;
};

// genotypes of a VCF block in HBM and, after hawk_gt_lists, the carried-variant lists of every chromosome copy
struct hawk_gt {
  hawk_ctx* ctx;
  uint64_t n_lines;
  uint32_t n_samples, n_var;
  uint8_t* d_codes;   // [n_lines][2 * n_samples]
  uint8_t* d_flags;   // [n_lines]
  uint64_t n_entries; // carried-variant entries over all columns (valid after hawk_gt_lists)
  uint64_t* d_col_off; uint32_t* d_idx; int32_t* d_o; int64_t* d_delta;
  uint64_t n_indel = 0;          // entries whose variant changes the length (var_chain != 0)
  uint32_t* d_indel = nullptr;   // their entry indices, ascending
  uint64_t* d_ioff = nullptr;    // [2 * n_samples + 1] where each column's carried indels start in d_indel
  std::vector<uint64_t> h_off, h_ioff;  // host copies of d_col_off / d_ioff (after hawk_gt_lists)
  std::vector<int64_t> h_delta;         // per column: sum of the length changes of its carried variants
};

struct hawk_table {
  hawk_hapset* hs;  // nullptr for a merged table (hawk_table_gather): it owns its columns
  hawk_ctx* ctx;
  uint64_t n_rows, n_cand, n_hits, cap;
  GuideCols cols;  // points into hs->colsA, or into own[] for a merged table
  uint32_t guidelen, pamlen, right;
  uint64_t n_groups;  // valid after hawk_table_collapse
  bool collapsed;
  uint64_t gen;     // hs->cols_gen when the table was written
  DevBuf own[8];
  // written by the cluster search of a plan view (hawk_csearch.hip): rows [0, offsets[plane_tiles]) are REF's, every other row is
  // a copy of one of t_rows template rows - which lets the collapse group the template rows instead of the table
  bool by_cluster = false;
  uint32_t plane_tiles = 0;
  uint64_t t_rows = 0;
};
// HAWK_E_INVALID when a later hawk_search on the same set has overwritten the table's columns
inline bool hawk_table_stale(const hawk_table* t) { return t->hs && t->gen != t->hs->cols_gen; }

int hawk_reserve_cols(DevBuf (&b)[8], uint64_t cap, GuideCols* c);

// shared by the C-ABI translation units (hawk_api_*.hip)
int hapset_create_impl(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, bool zero_planes, hawk_hapset** out, bool alloc_planes = true);
HapSetDev make_dev(const hawk_hapset* hs);
int make_scan_params(const hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t guidelen,
                            uint32_t right, bool need_v, ScanParams* sp);
int meta_build(uint32_t n, const std::vector<uint32_t>& hap_len, uint32_t bph, const uint8_t* is_ref, const int32_t* scan_start,
                      const int32_t* scan_stop, const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen,
                      int32_t ref_index, std::vector<TileMeta>* t0, int64_t* min_gen, int64_t* max_gen);
