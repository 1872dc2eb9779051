// hawk_device.h — device-side data model shared by the kernels (hawk_kernels.hip) and the
// C-ABI host layer (hawk_api.hip).  gfx950 only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HAWK_PLANES 5   // A, C, G, T (the IUPAC nibble's bits, encoder.py:18-34) and V (lower case)
#define HAWK_PAD 10     // GUIDESEQPAD, guide.py:21
#define HAWK_BLOCK 256  // threads per workgroup = 4 wavefronts
#define HAWK_WPT 4      // 32-base words per thread (one 16-byte load per plane)

// A haplotype set resident in HBM.  Plane p of haplotype h is plane[p] + h*S words; bit j of
// word w is base 32*w + j.  Rows are zero past hap_len and S >= words(hap_len) + 2, S % 4 == 0,
// so a 64-bit look-ahead never leaves the row and never sees another haplotype.
struct HapSetDev {
  uint32_t n_hap;
  uint32_t S;
  const uint32_t* plane[HAWK_PLANES];
  const uint32_t* hap_len;
  const uint8_t* is_ref;
  const int32_t* scan_start;
  const int32_t* scan_stop;
  const uint32_t* seg_off;
  const uint32_t* seg_rel;
  const int64_t* seg_gen;
  int32_t ref_index;
};

// Everything a search workgroup needs to know about its tile, in one 32-byte record so that a single scalar
// load (instead of a chain of dependent ones) stands between the kernel arguments and the plane loads.
struct TileMeta {
  uint32_t h, blk;       // haplotype row, tile index within the row
  uint32_t hap_len;
  int32_t scan_start, scan_stop;
  uint32_t is_ref;
  uint32_t seg0, seg_end;  // first position-map segment the tile needs; end of the haplotype's segments
};

struct ScanParams {
  uint64_t pam_fwd, pam_rev;  // PAM.bits / PAM.bitsrc: first PAM base in the most significant nibble
  int32_t pamlen, guidelen, right;
  int32_t L;                  // guidelen + pamlen
  uint32_t bph;               // workgroups per haplotype row
  uint32_t need;              // bit p set: plane p is read by this PAM (bit 4: V plane)
  int32_t poF, poR;           // raw scan: PAM offset inside the window per strand (0 = index by PAM position)
};

// Totals written by k_offsets.
struct ScanTotals {
  uint64_t n_keep;     // bits set in the keep planes (fwd + rev)
  uint64_t n_keep_fwd; // raw mode: forward total (reverse hits start here in the scan order)
  uint64_t n_cand;     // PAM hits passing is_pamhit_in_range
  uint64_t n_hits;     // PAM hits inside the scan range
};

// The guide table.  Two layouts:
//  * columns (SoA) - what the plane kernels and the per-word search of a view write: hap, pos, strand, start, stop, flags, cfdon,
//    win[5][cap]; `rows` is null;
//  * packed rows (`rows` != null; the column pointers are null) - what the cluster search of a view writes (hawk_csearch.hip):
//    ONE array of 64-byte rows, a single linear write stream.  Row i = 16 words:
//      [0] pos   [1] strand | flags << 1 | haplotype row << 9   [2] start - startp   [3] stop - start   [4,5] cfdon
//      [6 + 2 p, 7 + 2 p] window slice of plane p
//    (`startp` = genomic position of REF's base 0; haplotype rows < 2^23 and both differences within int32 are checked where the
//    rows are made).
struct GuideCols {
  uint32_t* hap;
  uint32_t* pos;
  uint8_t* strand;
  int64_t* start;
  int64_t* stop;
  uint8_t* flags;
  double* cfdon;
  uint64_t* win;  // [5][cap]
  uint64_t cap;
  uint4* rows;    // packed layout
  int64_t startp;
};
#define HAWK_ROW_HAP_SHIFT 9
#define HAWK_ROW_MAX_HAP (1u << 23)
#ifdef __HIPCC__
// field access for either layout (the branch is uniform over a launch)
__device__ __forceinline__ const uint32_t* gc_roww(const GuideCols& c, uint64_t i) { return reinterpret_cast<const uint32_t*>(c.rows) + i * 16; }
__device__ __forceinline__ uint32_t gc_hap(const GuideCols& c, uint64_t i) { return c.rows ? gc_roww(c, i)[1] >> HAWK_ROW_HAP_SHIFT : c.hap[i]; }
__device__ __forceinline__ uint32_t gc_pos(const GuideCols& c, uint64_t i) { return c.rows ? gc_roww(c, i)[0] : c.pos[i]; }
__device__ __forceinline__ uint32_t gc_strand(const GuideCols& c, uint64_t i) { return c.rows ? (gc_roww(c, i)[1] & 1u) : (uint32_t)c.strand[i]; }
__device__ __forceinline__ uint32_t gc_flags(const GuideCols& c, uint64_t i) { return c.rows ? ((gc_roww(c, i)[1] >> 1) & 0xffu) : (uint32_t)c.flags[i]; }
__device__ __forceinline__ int64_t gc_start(const GuideCols& c, uint64_t i) { return c.rows ? c.startp + (int64_t)(int32_t)gc_roww(c, i)[2] : c.start[i]; }
__device__ __forceinline__ int64_t gc_stop(const GuideCols& c, uint64_t i) {
  if (!c.rows) return c.stop[i];
  const uint32_t* w = gc_roww(c, i);
  return c.startp + (int64_t)(int32_t)w[2] + (int64_t)(int32_t)w[3];
}
__device__ __forceinline__ double gc_cfdon(const GuideCols& c, uint64_t i) {
  if (!c.rows) return c.cfdon[i];
  const uint32_t* w = gc_roww(c, i);
  return __longlong_as_double((long long)((uint64_t)w[4] | ((uint64_t)w[5] << 32)));
}
__device__ __forceinline__ uint64_t gc_win(const GuideCols& c, int pl, uint64_t i) {
  if (!c.rows) return c.win[(size_t)pl * c.cap + i];
  const uint32_t* w = gc_roww(c, i);
  return (uint64_t)w[6 + 2 * pl] | ((uint64_t)w[7 + 2 * pl] << 32);
}
#endif
// columns [0, n) -> packed rows [0, n) (n read on the device: *n_dev, or n_host when n_dev is null) and back (null destination
// columns are skipped).  Packing reports HAWK_E_UNSUPPORTED through *status for a row the layout cannot hold.
void hawk_launch_rows_pack(hipStream_t st, const GuideCols& src, const uint64_t* n_dev, uint64_t n_host, uint64_t max_rows, uint4* rows, int64_t startp,
                           int* status);
void hawk_launch_rows_unpack(hipStream_t st, const uint4* rows, uint64_t n, int64_t startp, const GuideCols& dst);

struct GuideParams {
  int32_t pamlen, guidelen, right, L;
  int32_t score_cfdon;
  const double* cfd_mm;   // device, [20][4][4]
  const double* cfd_pam;  // device, [16]
  uint32_t bph;
};

// Where the REF haplotype has candidate windows, per strand (scan range and in-range test of
// REF expressed on the window start q), and the genomic position of its base 0.
struct RefInfo {
  int32_t index;
  int32_t lo[2], hi[2];
  int64_t startp;
  const uint32_t* bits[2];  // per strand: bit q set <=> REF has a candidate window starting at q (k_ref_bits); n_bits valid bits
  uint32_t n_bits;
};
// The search straight from an expansion plan (hawk_vsearch.hip): what it reads beside the rows' metadata.
struct VcArgs {
  const uint32_t* ref[4];   // REF planes A, C, G, T (ref_S words each, zero past the last base)
  uint32_t ref_S;
  const void* recs_;        // HxVar[]: one record per carried variant of every row (hawk_hx.h)
  const uint8_t* alt_codes;
  const uint64_t* hv_off;   // [n_hap + 1] record range of every row
  const void* tiles_;       // HxTile[n_hap * tiles per row]
  const uint4* hp;          // REF's PAM hits: {hit bits of word w on strand 0, hits in the words before, the same for strand 1}, w <= ref_S
};
void hawk_launch_vsearch(hipStream_t st, int pass, const HapSetDev& hs, const VcArgs& va, const ScanParams& p, const struct GuideParams& gp,
                         const struct RefInfo& ri, const TileMeta* tmeta, uint32_t* counts, uint32_t* counts0, unsigned long long* shards,
                         const uint64_t* offsets, struct GuideCols out, int* status, uint32_t tile0, uint32_t n_tiles);
void hawk_launch_ref_hits(hipStream_t st, const HapSetDev& hs, const ScanParams& p, int32_t ref_index, void* hp);
// The cluster dictionary of an expansion plan (hawk_csearch.hip): every row's records cut into clusters (variants whose
// alleles lie within 64 positions of each other), identical clusters of different rows numbered once.
struct ClDict {
  uint32_t n_inst, n_uniq;
  const uint32_t* inst_uid;  // per instance, in (row, position) order: its distinct cluster, 0xffffffff for none
  const int32_t* inst_o;     // row position of the cluster's first allele
  const uint32_t* inst_row;
  const int32_t* inst_pa;    // start of the clean run in front of the instance and the REF shift in force there
  const int32_t* inst_rb;
  const uint32_t* u_rec;     // per distinct cluster: first record of the representative instance, its records, its row,
  const uint32_t* u_n;       // the row position of its first allele
  const uint32_t* u_row;
  const int32_t* u_o;
  const uint32_t* u_seg;     // a position-map segment of that row in force in front of every position a search of the cluster maps
};
size_t hawk_cs_row_bytes();
void hawk_launch_scan_u32(hipStream_t st, const uint32_t* cnt, uint32_t n, uint32_t* off);  // exclusive scan into n + 1 offsets, one workgroup
void hawk_launch_scan2_u32(hipStream_t st, const uint32_t* cnt_a, const uint32_t* cnt_b, uint32_t n, uint32_t* off_a, uint32_t* off_b);  // two arrays, one launch
void hawk_launch_hx_heads(hipStream_t st, const void* recs, const uint32_t* hv_idx, uint64_t n, void* heads);  // {o, rs, alt_len, variant} per record
// (the hawk_launch_cl_* passes take the HEADS as `recs`)
uint32_t hawk_cl_chunk_bound(uint64_t n_records, uint32_t n_rows);  // room for ch_row / the chunks' counts
void hawk_launch_cl_chunks(hipStream_t st, const uint64_t* hv_off, const uint8_t* is_ref, const int32_t* ss, const int32_t* se, uint32_t n_rows,
                           uint32_t* ch_off, uint32_t* ch_row);
void hawk_launch_cl_count(hipStream_t st, const void* recs, const uint64_t* hv_off, const uint32_t* hap_len, const int32_t* ss, const int32_t* se,
                          const uint32_t* ch_off, const uint32_t* ch_row, uint32_t n_rows, uint32_t n_var, uint32_t ch_bound, uint32_t* cnt /* zeroed */,
                          uint32_t* lcnt /* zeroed */);  // per chunk: the instances it opens, and how many of them go on the list
void hawk_launch_cl_fill(hipStream_t st, const void* recs, const uint64_t* hv_off, const uint32_t* hap_len, const int32_t* ss, const int32_t* se,
                         uint32_t n_rows, const uint32_t* ch_off, const uint32_t* ch_row, uint32_t ch_bound, const uint32_t* inst_base,
                         const uint32_t* list_base, int32_t* o, uint32_t* row, int32_t* pa, int32_t* rb, uint32_t* inst_uid,
                         void* var_desc /* 8 B per variant, zeroed */, uint32_t* claim_bits /* n_var bits, zeroed */, uint32_t n_var,
                         void* cx_list /* hawk_cl_listed_bytes() per listed instance */, uint32_t* status);
size_t hawk_cl_slot_bytes();    // a slot of the table of the listed instances' clusters
size_t hawk_cl_listed_bytes();  // a list entry
void hawk_launch_cl_finish(hipStream_t st, uint32_t list_bound, const uint32_t* n_inst_dev, const uint32_t* n_list_dev, uint32_t* counters /* 8, zeroed */,
                           unsigned long long* results /* 8: instances, table clusters, variant clusters, template rows, status */, uint32_t n_var, uint32_t u_cap,
                           void* tab /* zeroed */, uint32_t mask, uint32_t max_probe, uint32_t fail_bit, const void* cx_list, uint32_t* cx_state,
                           const void* var_desc, const void* recs, uint32_t* inst_uid, const uint32_t* inst_row, const uint32_t* seg_off,
                           const uint32_t* seg_rel, uint32_t* u_rec, uint32_t* u_n, uint32_t* u_row, int32_t* u_o, uint32_t* u_seg, uint32_t* u_span2,
                           uint32_t* status);
void hawk_launch_cs_templates(hipStream_t st, const HapSetDev& hs, const VcArgs& va, const ClDict& cd, const ScanParams& p, const struct GuideParams& gp,
                              const struct RefInfo& ri, void* res /* 32 B per distinct cluster */, uint32_t* tbase, void* trows,
                              unsigned long long* t_count, uint64_t t_cap, int* status);
void hawk_launch_cs_count(hipStream_t st, const HapSetDev& hs, const VcArgs& va, const ClDict& cd, const ScanParams& p, const void* res,
                          const unsigned long long* t_count, uint64_t t_cap, uint32_t* group_counts, uint32_t* counts, uint32_t* inst_tb,
                          unsigned long long* shards);
void hawk_launch_cs_emit_rows(hipStream_t st, const ClDict& cd, const uint32_t* counts, const uint32_t* inst_tb, const void* trows, const uint64_t* offsets,
                              const unsigned long long* t_count, uint64_t t_cap, void* rows, uint64_t cap, int* status);
// hawk_meta.hip: the rows' metadata of an expansion plan, built on the device
void hawk_launch_list_check(hipStream_t st, const uint64_t* row_off, uint32_t n_rows, const uint32_t* hv_idx, const int32_t* hv_o,
                            const int32_t* v_r0, const int32_t* v_span, const int32_t* v_chain, uint32_t n_var, uint32_t ref_len,
                            int check_clamp, uint32_t* status);
void hawk_launch_segments(hipStream_t st, const uint64_t* ioff, const uint32_t* indel, const uint32_t* hv_idx, const int32_t* hv_o,
                          const int32_t* v_r0, const int32_t* v_chain, const uint32_t* hap_len, uint32_t n_rows, int64_t startp,
                          uint32_t* seg_cnt, uint32_t* seg_off, uint32_t* seg_rel /* null: count + offsets only */, int64_t* seg_gen);
void hawk_launch_rev_lookup(hipStream_t st, const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, const uint32_t* hap_len,
                            uint32_t n_rows, int64_t g0, int64_t g1, int64_t* out0, int64_t* out1);
void hawk_launch_tile_meta(hipStream_t st, const uint32_t* seg_off, const uint32_t* seg_rel, const uint32_t* hap_len, const uint8_t* is_ref,
                           const int32_t* scan_start, const int32_t* scan_stop, uint32_t n_rows, uint32_t bph, TileMeta* tm);
void hawk_launch_ref_bits(hipStream_t st, const HapSetDev& hs, const ScanParams& p, const RefInfo& ri, uint32_t* bitsF, uint32_t* bitsR);

// K7 records
struct OtSite {   // one PAM-bearing genome window in guide orientation
  uint64_t code;  // base i of the window at bits 2i,2i+1 (A0 C1 G2 T3)
  uint32_t nmask; // bit i: base i is ambiguous (N / IUPAC)
  uint32_t q;     // window start inside the row | strand << 31
  uint32_t row;
  uint32_t pad;
};
struct OtHit { uint64_t site; uint32_t guide; uint32_t mm; };
// Seeds of the pigeonhole filter: the spacer is cut into max_mm + 1 blocks, so a pair within max_mm mismatches agrees
// exactly in at least one block.  Per block: its first base, the bases used as the bucket key (<= 6), where its
// bucket offsets start in `goff`, and the folded-XOR bits (even positions) of the key bases.
#define OT_MAX_BLOCKS 8
struct OtSeeds {
  int32_t nb;
  int32_t start[OT_MAX_BLOCKS], klen[OT_MAX_BLOCKS];
  uint32_t off_base[OT_MAX_BLOCKS];
  uint64_t pmask2[OT_MAX_BLOCKS];
};

// Pair seeds: the spacer cut into max_mm + 2 blocks, so a pair within max_mm mismatches agrees exactly in at least TWO of them;
// the guides are bucketed per PAIR of blocks (i < j) by the key bases of both (<= 4 each: tables of <= 4^8 buckets), which
// leaves ~4^-8 of the site x guide pairs as candidates per table instead of ~4^-4.  off_base[p]: where pair p's bucket offsets
// start in `goff`; its codes / guide ids are rows p of gcode / gid ([n_pairs][n_guides]).
#define OT_MAX_PAIRS (OT_MAX_BLOCKS * (OT_MAX_BLOCKS - 1) / 2)
struct OtPairSeeds {
  int32_t nb, n_pairs;
  int32_t start[OT_MAX_BLOCKS], klen[OT_MAX_BLOCKS];
  uint64_t pmask2[OT_MAX_BLOCKS];
  uint8_t pi[OT_MAX_PAIRS], pj[OT_MAX_PAIRS];
  uint32_t off_base[OT_MAX_PAIRS];
};
void hawk_launch_ot_match_pairs(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtPairSeeds& sd, const uint32_t* goff,
                                const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, int guidelen, int sp0, int max_mm,
                                OtHit* hits, uint64_t cap, unsigned long long* n_hits);

// launch wrappers
void hawk_launch_ot_onehot(hipStream_t st, uint32_t* const* plane, uint64_t nwords);
void hawk_launch_ot_sites(hipStream_t st, const HapSetDev& hs, const ScanParams& p, const uint32_t* keepF, const uint32_t* keepR,
                          const uint64_t* offsets, OtSite* sites);
void hawk_launch_ot_match(hipStream_t st, const OtSite* sites, uint64_t n_sites, const uint64_t* guides, uint32_t n_guides,
                          int guidelen, int sp0, int max_mm, OtHit* hits, uint64_t cap, unsigned long long* n_hits);
void hawk_launch_ot_match_seeded(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtSeeds& sd, const uint32_t* goff,
                                 const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, int guidelen, int sp0, int max_mm,
                                 OtHit* hits, uint64_t cap, unsigned long long* n_hits);
#define OT_LDS_CHUNK 1024  // guides per LDS-resident chunk of the seeded match
#define OT_LDS_KEYS 256    // buckets per block in the LDS variant (4 key bases)
void hawk_launch_ot_match_seeded_lds(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtSeeds& sd, const uint32_t* goff,
                                     const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, uint32_t n_chunks, int guidelen,
                                     int sp0, int max_mm, OtHit* hits, uint64_t cap, unsigned long long* n_hits);
void hawk_launch_ot_gather(hipStream_t st, const OtSite* sites, const OtHit* hits, uint64_t n_hits, OtSite* out);
void hawk_launch_pack(hipStream_t st, const uint8_t* ascii, const uint64_t* seq_off, uint32_t hap0, uint32_t n_hap_batch,
                      uint64_t batch_base, const uint32_t* hap_len, uint32_t S, uint32_t* const* plane,
                      unsigned long long* bad_index);
void hawk_launch_scan_raw(hipStream_t st, const HapSetDev& hs, const ScanParams& p, uint32_t* keepF, uint32_t* keepR,
                          uint32_t* counts);
void hawk_launch_emit_hits(hipStream_t st, const HapSetDev& hs, uint32_t bph, const uint32_t* keepF, const uint32_t* keepR,
                           const uint64_t* offsets, uint64_t n_fwd_total, uint32_t* hits_fwd, uint32_t* hits_rev);
void hawk_launch_search(hipStream_t st, int pass, const HapSetDev& hs, const ScanParams& p, const GuideParams& gp,
                        const RefInfo& ri, const TileMeta* tmeta, uint32_t* counts, unsigned long long* shards,
                        const uint64_t* offsets, GuideCols out, int* status, uint32_t* lists, uint32_t* big_count,
                        unsigned long long* big_list, hipEvent_t mid = nullptr, uint32_t n_tiles = 0xffffffffu);
size_t hawk_collapse_temp_bytes(uint64_t n, unsigned begin_bit, unsigned end_bit);
int hawk_launch_collapse(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int right,
                         int flank_up, int flank_down, int64_t base, unsigned begin_bit, unsigned end_bit, uint64_t seed, void* temp, size_t temp_bytes, uint64_t* keys, uint32_t* vals,
                         uint32_t* flags, uint32_t* gidx, unsigned long long* counters, uint64_t* group_off, uint8_t* gc_num,
                         uint8_t* gc_den, uint32_t* id2 /* may alias gidx */, void* full /* null: identity by hash */, int weak_hash = 0);
void hawk_launch_collapse_verify(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int flank_up,
                                 int flank_down, const uint32_t* perm, const uint32_t* grp_a, const uint32_t* grp_b, const uint64_t* group_off,
                                 unsigned long long* mismatches);
void hawk_launch_collapse_verify_rows(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, uint32_t G, int guidelen, int pamlen,
                                      int flank_up, int flank_down, const uint32_t* perm, const uint32_t* slot_of_row, const uint32_t* slot2rank,
                                      const uint64_t* group_off, void* full /* 64 B per group */, unsigned long long* mismatches);
void hawk_launch_rows_equal(hipStream_t st, const HapSetDev& hs, uint32_t n_pairs, const uint32_t* ra, const uint32_t* rb, uint8_t* equal);
size_t hawk_collapse_full_bytes(uint64_t n);
size_t hawk_collapse_expand_temp_bytes(uint64_t n);
int hawk_launch_collapse_expand(hipStream_t st, const GuideCols& c, uint64_t n, unsigned gbits, int guidelen, int pamlen, int right, void* temp,
                                size_t temp_bytes, uint32_t* gid, uint32_t* vals, uint64_t* group_off, uint8_t* gc_num, uint8_t* gc_den);
void hawk_launch_cc_ucnt(hipStream_t st, const void* res, uint32_t nu, uint32_t* cnt);
void hawk_launch_cc_mini(hipStream_t st, const GuideCols& c, uint64_t r0, const void* trows, uint64_t t_rows, const uint64_t* moff, const uint32_t* tbase,
                         uint32_t nu, int64_t startp, GuideCols m);
void hawk_launch_cc_gidm(hipStream_t st, const uint32_t* perm, const uint64_t* goff, uint64_t nm, uint64_t G, uint32_t* gidm);
void hawk_launch_cs_gid(hipStream_t st, const ClDict& cd, const void* res, const uint64_t* moff, const uint64_t* offsets, uint64_t r0, uint64_t n,
                        const uint32_t* gidm, uint32_t* gid, uint32_t* vals);
size_t hawk_collapse_hash_temp_bytes(uint64_t n, uint32_t C);
int hawk_launch_collapse_hash1(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int flank_up,
                               int flank_down, int64_t base, uint64_t seed, void* temp, size_t temp_bytes, void* table, uint32_t C,
                               uint32_t* flags, uint32_t* dense, uint64_t* gkey, uint32_t* gslot, uint32_t* slot_of_row,
                               unsigned long long* counters);
int hawk_launch_collapse_hash2(hipStream_t st, const GuideCols& c, uint64_t n, uint32_t G, int guidelen, int pamlen, int right, unsigned key_end_bit,
                               void* temp, size_t temp_bytes, uint64_t* gkey, uint32_t* gslot, uint32_t C, uint32_t* slot2rank,
                               const uint32_t* slot_of_row, uint32_t* gid, uint32_t* vals, uint64_t* group_off, uint8_t* gc_num,
                               uint8_t* gc_den);
void hawk_launch_collapse_export(hipStream_t st, const GuideCols& c, uint64_t n, uint64_t ng, const uint32_t* perm,
                                 const uint64_t* group_off, const GuideCols& rep, uint32_t* member_hap);
void hawk_launch_gt_parse(hipStream_t st, const uint8_t* text, const uint64_t* line_off, const uint64_t* gt_off, uint64_t n_lines,
                          uint32_t n_samples, uint8_t* codes, uint8_t* flags);
void hawk_launch_gt_count(hipStream_t st, const uint8_t* codes, uint32_t n_cols, const uint32_t* var_line, const uint8_t* var_allele,
                          const int32_t* var_chain, uint32_t n_var, unsigned long long* ballots, uint32_t* col_count /* [2 * n_cols] */);
void hawk_launch_gt_fill(hipStream_t st, uint32_t n_cols, const int32_t* var_r0, const int32_t* var_chain, uint32_t n_var,
                         const unsigned long long* ballots, const uint64_t* col_off, uint32_t* hv_idx, int32_t* hv_o, int64_t* col_delta,
                         const uint64_t* indel_off, uint32_t* indel_entry);
#define HAWK_LIST_CAP 512  // entries per tile in the count pass -> emit pass hand-over list (hawk_search.hip LIST_CAP)
void hawk_launch_mscan(hipStream_t st, const uint32_t* counts, uint64_t n, unsigned long long* partial,
                       const unsigned long long* shards, uint64_t* offsets, ScanTotals* totals);
void hawk_launch_cfd(hipStream_t st, const char* wt, const char* sg, uint32_t len, const char* pam2, uint64_t n,
                     const double* mm, const double* pamtab, double* out, int* status);
void hawk_launch_deepcpf1(hipStream_t st, const char* seqs, uint64_t n, const float* w, float* out, int* status);
void hawk_launch_tm_nn(hipStream_t st, const char* seqs, uint32_t len, uint64_t n, double* out, int* status);
void hawk_launch_azimuth(hipStream_t st, const char* seqs, uint64_t n, uint32_t n_trees, const int32_t* tree_off,
                         const int32_t* feature, const int32_t* left, const int32_t* right, const double* threshold,
                         const double* value, double init, double lr, double* out, double* feats_out, int* status);
void hawk_launch_gbt(hipStream_t st, const double* feats, uint64_t n, uint32_t nf, uint32_t n_trees, const int32_t* tree_off,
                     const int32_t* feature, const int32_t* left, const int32_t* right, const double* threshold, const double* value,
                     double init, double lr, int cast_f32, double* out);
uint32_t hawk_hx_tiles_per_row(uint32_t S);
size_t hawk_hx_record_bytes();
size_t hawk_hx_tile_bytes();
void hawk_launch_hx_prepare(hipStream_t st, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o, uint64_t ncar,
                            const uint32_t* v_r0, const uint32_t* v_span, const uint32_t* v_alt_off, const uint32_t* v_alt_len,
                            const void* v_am /* uint4 per variant */, const uint32_t* hap_len, uint32_t n_hap, uint32_t S, void* recs,
                            void* tiles);
void hawk_launch_hx_build(hipStream_t st, const uint32_t* const* ref, uint32_t ref_S, const void* recs, const uint8_t* alt_codes,
                          const uint64_t* hv_off, const uint32_t* hap_len, uint32_t n_hap, uint32_t S, uint32_t* const* plane,
                          const void* tiles);
void hawk_launch_hx_hash(hipStream_t st, uint32_t* const* plane, uint32_t n_hap, uint32_t S, unsigned long long* hash);
