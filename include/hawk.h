/*
 * hawk.h — C ABI of libhawk_hip.so: the MI355X (gfx950) implementation of CRISPR-HAWK's
 * variant-aware guide search hot path.
 *
 * The reference (pinellolab/CRISPR-HAWK) has no FFI layer: its hot path is a chain of plain
 * Python call sites in crisprhawk_search() (crisprhawk.py:121-138).  Each entry point below
 * names the reference call it replaces; INTEGRATION.md shows the ctypes binding a maintainer
 * would add on the reference side.  Conventions: plain pointers and sizes, no C++/torch types;
 * every function returns HAWK_OK (0) or a negative hawk_status and never throws; outputs are
 * caller-allocated with an explicit capacity, or stay resident in HBM behind an opaque handle
 * until downloaded; one hawk_ctx per process per device; not fork-safe (a HIP context does
 * not survive fork — the reference's ProcessPoolExecutor scorers, scoring.py:129, must call
 * from the parent).
 *
 * Data layout in HBM: a haplotype set is five bit-planes
 * A, C, G, T (the four bits of the reference's IUPAC nibble, encoder.py:18-34) and V
 * ("this base came from a variant" = lower case in the reference's haplotype strings),
 * one bit per base, 32 bases per little-endian 32-bit word, rows padded to a common
 * 16-byte-aligned stride.
 */
#ifndef HAWK_H
#define HAWK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HAWK_OK = 0,
  HAWK_E_INVALID = -1,    /* bad argument */
  HAWK_E_HIP = -2,        /* HIP runtime failure (hawk_last_hip_error() has the text) */
  HAWK_E_CAPACITY = -3,   /* caller buffer too small; required size is reported */
  HAWK_E_IUPAC = -4,      /* non-IUPAC character: CrisprHawkIupacTableError (encoder.py:37-45) */
  HAWK_E_CFD = -5,        /* non-ACGT base under a CFD lookup: CrisprHawkCfdScoreError (cfdscore.py:93-94) */
  HAWK_E_NODEVICE = -6,   /* no gfx950 device visible */
  HAWK_E_UNSUPPORTED = -7, /* parameter outside the kernel's range (e.g. guidelen+pamlen > 44) */
  HAWK_E_COMM = -8,        /* RCCL failure or librccl.so missing (hawk_comm_last_error() has the text) */
  HAWK_E_OVERLAP = -9,     /* a chromosome copy carries overlapping variants (haplotype.py:214-252 raises on them) */
  HAWK_E_CLAMP = -10       /* an indel reaches past the region's original length (the reference's clamp, haplotype.py:199-201) */
} hawk_status;

typedef struct hawk_ctx hawk_ctx;
typedef struct hawk_hapset hawk_hapset;
typedef struct hawk_table hawk_table;

#define HAWK_GUIDESEQPAD 10 /* guide.py:21 */
#define HAWK_MAX_CORE 44    /* guidelen + pamlen limit: the 20-nt padded window must fit 64 bits */

/* ---- context -----------------------------------------------------------------------
 * One context = one HIP stream per device and process.  hawk_init for a device that already has a context returns that
 * same context (reference-counted; the last hawk_destroy releases it): the library serves device memory from a caching
 * allocator whose reuse of freed blocks is ordered by that single stream. */
int hawk_device_count(int* n);
int hawk_init(int device, hawk_ctx** out);
void hawk_destroy(hawk_ctx* ctx);
const char* hawk_strerror(int status);
const char* hawk_last_hip_error(void);
/* the hipStream_t all of this context's work is enqueued on (for external event timing) */
void* hawk_stream(hawk_ctx* ctx);
int hawk_sync(hawk_ctx* ctx);
/* Device memory is served by a caching allocator (freed planes / columns / workspaces are reused by the next
 * haplotype set of the same shape, e.g. the next tile of a whole-contig search); this returns the cache to HIP. */
int hawk_release_cached_memory(hawk_ctx* ctx);
/* Page-locked host memory for buffers a download writes into (hawk_table_download, hawk_table_collapse_export ...): the copy
 * then runs at link speed.  Plain host memory is accepted everywhere; this is an optimisation for large results. */
int hawk_host_alloc(hawk_ctx* ctx, uint64_t bytes, void** out);
void hawk_host_free(void* p);

/* ---- haplotype set: replaces encode() per haplotype (crisprhawk.py:64-81, encoder.py:48-57)
 * plus the Haplotype fields the search reads (haplotype.py:395-491) ---------------------- */

/* Allocate planes for n_hap haplotypes of the given lengths (bases). */
int hawk_hapset_create(hawk_ctx* ctx, uint32_t n_hap, const uint32_t* hap_len, hawk_hapset** out);
void hawk_hapset_destroy(hawk_hapset* hs);

/* K1: pack cased ASCII haplotypes (host memory, concatenated, seq_off[n_hap+1] byte offsets)
 * into the planes.  Upper/lower case is data: lower case sets the V plane.  On a non-IUPAC
 * character returns HAWK_E_IUPAC and *bad_index = its offset in `seqs` (encoder.py:37-45). */
int hawk_hapset_pack_ascii(hawk_hapset* hs, const char* seqs, const uint64_t* seq_off, uint64_t* bad_index);

/* Per-haplotype metadata the search needs:
 *   is_ref[h]      1 iff haplotype.samples == "REF" (search_guides.py:468-471)
 *   scan_start/stop  compute_scan_start_stop() (search_guides.py:49-84), relative positions
 *   seg_off[n_hap+1], seg_rel[], seg_gen[]  the haplotype position map (haplotype.py:90-159)
 *     as unit-slope segments: posmap[rel] = seg_gen[k] + (rel - seg_rel[k]) for the last k in
 *     [seg_off[h], seg_off[h+1]) with seg_rel[k] <= rel.
 *   ref_index      index of the REF haplotype, or -1 */
int hawk_hapset_set_meta(hawk_hapset* hs, const uint8_t* is_ref, const int32_t* scan_start, const int32_t* scan_stop,
                         const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, int32_t ref_index);

/* Tiles of a larger region (region-tiled whole-contig search): a haplotype row is grouped with the REF guide at its
 * (start, strand) (remove_redundant_guides, search_guides.py:340-369; CFDon's wild type, scoring.py:368-381) wherever in
 * the REGION that guide's PAM lies, while this set's REF row only scans the tile it owns.  [start, stop) is the
 * region's scan range (compute_scan_start_stop) in the REF row's relative positions, clipped to the row; default =
 * REF's own scan range.  Call after hawk_hapset_set_meta. */
int hawk_hapset_set_ref_partner_range(hawk_hapset* hs, int32_t start, int32_t stop);

/* Row stride in 32-bit words, and a copy of one plane (0..4 = A,C,G,T,V) to host
 * (n_hap * stride words) — for tests and for decoding windows on the host. */
int hawk_hapset_stride(const hawk_hapset* hs, uint32_t* stride_words);
int hawk_hapset_download_plane(hawk_hapset* hs, int plane, uint32_t* out_words);
/* equal[i] = 1 iff rows rows_a[i] and rows_b[i] hold the same bases, case included (all five planes compared): what makes
 * collapsing haplotypes on their 128-bit content hash (hawk_xplan_run's hash_out; haplotypes.py:274-294 compares the
 * strings) exact - the caller checks every row it is about to alias onto another. */
int hawk_hapset_rows_equal(hawk_hapset* hs, uint32_t n_pairs, const uint32_t* rows_a, const uint32_t* rows_b, uint8_t* equal);
/* Upload planes computed elsewhere (5 * n_hap * stride words, plane-major). */
int hawk_hapset_upload_planes(hawk_hapset* hs, const uint32_t* planes);

/* ---- haplotype expansion on the device, replacing Haplotype.add_variants_phased
 * (haplotype.py:214-252) per chromosome copy.  ref_set: a one-row hapset holding the REF region.
 * Variant table (position-sorted, non-overlapping): v_r0 = position in the region, v_span = REF bases
 * replaced (SNV 1, deletion len(ref), insertion 1), alt allele = v_alt_len IUPAC codes (encoder.py
 * nibbles, one per byte) at alt_codes + v_alt_off.  Row h of the new set carries variants
 * hv_idx[hv_off[h] .. hv_off[h+1]) (ascending) whose output start positions hv_o are the exclusive
 * prefix sums r0 + sum of earlier (alt_len - span); hap_len[h] = region length + its total change
 * (all verified on the host before launch).  Alt bases are written with the V plane set, exactly as
 * the reference lower-cases them.  hash_out (optional, 2 words per row): content hash for
 * collapse_haplotypes (haplotypes.py:274-294).  The caller finishes with hawk_hapset_set_meta. */
int hawk_hapset_expand(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                       const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len,
                       uint32_t n_hap, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o,
                       const uint32_t* hap_len, hawk_hapset** out, uint64_t* hash_out, float* kernel_ms);

/* The same expansion as a reusable plan: hawk_xplan_create validates the inputs once and keeps them in HBM (it copies
 * the REF planes, so ref_set may be destroyed), hawk_xplan_run writes a fresh haplotype set from them with device work
 * only, hawk_xplan_set_meta stores the metadata (arguments as hawk_hapset_set_meta) every later run installs into the
 * set it returns.  This is what the region-tiling loop of a whole-contig search (search_guides.py:510-548 over one
 * 50 Mb region; here one tile at a time inside a fixed HBM budget) runs per tile.  hash_out / kernel_ms may be NULL
 * (then the run is asynchronous on the context's stream). */
typedef struct hawk_xplan hawk_xplan;
int hawk_xplan_create(hawk_hapset* ref_set, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                      const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, uint32_t alt_codes_len,
                      uint32_t n_hap, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o,
                      const uint32_t* hap_len, hawk_xplan** out);
int hawk_xplan_set_meta(hawk_xplan* x, const uint8_t* is_ref, const int32_t* scan_start, const int32_t* scan_stop,
                        const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, int32_t ref_index);
int hawk_xplan_set_ref_partner_range(hawk_xplan* x, int32_t start, int32_t stop);
int hawk_xplan_run(hawk_xplan* x, hawk_hapset** out, uint64_t* hash_out, float* kernel_ms);
/* A VIEW of the plan's rows: a haplotype set with the plan's metadata (hawk_xplan_set_meta must have been called, REF = row
 * 0) but WITHOUT planes.  hawk_search on a view computes the same table, totals and row order as on the set hawk_xplan_run
 * writes - encode (encoder.py:48-57 over every haplotype) and search (search_guides.py:510-548) in one step, straight from
 * REF + the rows' variant records: windows without a variant base are dropped by the reference (search_guides.py:468-471),
 * so only the words around a row's variants are assembled (in registers) and matched; the PAM hits of the verbatim REF
 * stretches in between, which only the totals count, come from prefix counts over REF's own hits.  Everything that reads
 * planes (hawk_pam_scan, hawk_hapset_download_plane, hawk_offtarget_scan ...) returns HAWK_E_INVALID on a view; tables,
 * collapse, export and gather work as on any set.  The plan must outlive the view. */
int hawk_xplan_view(hawk_xplan* x, hawk_hapset** out);
/* The cluster dictionary hawk_xplan_view builds for a plan (crispr-hawk_amd/csrc/hawk_csearch.hip): a row's carried variants
 * fall into clusters (alleles within 64 positions of each other - the reach of one padded guide window); a cluster carried by
 * many chromosome copies is searched ONCE per hawk_search and its rows copied to every carrier.  usable = 0: the plan is
 * searched per row instead (status 1: a chain of more than 4096 variants, 2: two different clusters with one hash, 4: too
 * little sharing - fewer than HAWK_CLUSTER_MIN_SHARE (3) instances per distinct cluster - or more than
 * HAWK_CLUSTER_MAX_SLOTS (2^27) template rows).  Same table either way, up to the order of a haplotype's rows.
 * No reference counterpart: the reference searches every haplotype string on its own (search_guides.py:510-548). */
int hawk_xplan_cluster_stats(const hawk_xplan* x, uint32_t* usable, uint32_t* n_instances, uint32_t* n_distinct, uint64_t* template_slots,
                             float* build_ms, uint32_t* status);
/* Builds the dictionary again from the plan's records (same result; buffers reused).  What bench.py calls inside its timed
 * step so that `value` carries the dictionary's cost: the product builds one per plan and searches it once (pipeline.search_files). */
int hawk_xplan_cluster_rebuild(hawk_xplan* x);
/* Plan creation straight from the genotype inversion: `g` is a hawk_gt after hawk_gt_lists, whose carried-variant lists
 * are still in HBM - they are used in place (rows = REF + every chromosome copy with a non-empty list, in column order),
 * nothing is downloaded, and what the host used to do over every list entry runs as kernels: the ascending /
 * non-overlapping check (HAWK_E_OVERLAP), the reference's end-of-region clamp for indels when check_clamp != 0
 * (HAWK_E_CLAMP), the position-map segments of every row (haplotype.py:90-159) and posmap_rev at the two genomic positions
 * rev_g0 / rev_g1 the scan bounds start from (search_guides.py:49-84).  v_chain[i] = alt_len - span.  hawk_xplan_rows
 * returns the rows' lengths and the two look-ups (-1: the position is deleted from that row); the caller turns them into
 * scan ranges (rows collapsed onto another get an empty one) and hawk_xplan_finish_meta builds the per-tile records.
 * hawk_xplan_segments downloads the segments (for labels / reports); hawk_xplan_install_meta gives a set hawk_xplan_run
 * wrote BEFORE the metadata existed (the run whose hashes decide which rows collapse) the finished metadata. */
typedef struct hawk_gt hawk_gt;
int hawk_xplan_create_gt(hawk_hapset* ref_set, hawk_gt* g, uint32_t n_var, const uint32_t* v_r0, const uint32_t* v_span,
                         const uint32_t* v_alt_off, const uint32_t* v_alt_len, const int32_t* v_chain, const uint8_t* alt_codes,
                         uint32_t alt_codes_len, int64_t startp, int check_clamp, int64_t rev_g0, int64_t rev_g1, uint32_t* n_hap_out,
                         hawk_xplan** out);
int hawk_xplan_rows(hawk_xplan* x, uint32_t* hap_len, int64_t* rev0, int64_t* rev1);
int hawk_xplan_finish_meta(hawk_xplan* x, const int32_t* scan_start, const int32_t* scan_stop);
int hawk_xplan_segments(hawk_xplan* x, uint32_t* seg_off, uint32_t* seg_rel, int64_t* seg_gen, uint64_t cap, uint64_t* n_seg);
int hawk_xplan_install_meta(hawk_xplan* x, hawk_hapset* hs);
void hawk_xplan_destroy(hawk_xplan* x);

/* ---- K2: pam_search() (search_guides.py:102-131) ---------------------------------------
 * Raw PAM hits of every haplotype inside its [scan_start, scan_stop): ascending relative
 * positions per haplotype, forward-PAM hits in hits_fwd and reverse-complement-PAM hits in
 * hits_rev, haplotype h occupying [off_fwd[h], off_fwd[h+1]) (off arrays have n_hap+1
 * entries).  pam_fwd/pam_rev are PAM.bits / PAM.bitsrc (pam.py:127-141). */
int hawk_pam_scan(hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t* hits_fwd,
                  uint32_t* hits_rev, uint64_t cap_fwd, uint64_t cap_rev, uint64_t* off_fwd, uint64_t* off_rev);

/* Average duration (HIP events on the context stream) of the K2 PAM-scan kernel alone over `reps`
 * back-to-back launches on this set: plane codes in, one forward and one reverse hit bit per position
 * out - the kernel SURVEY.md §8(d) prices at 0.75 B per scanned position. */
int hawk_pam_scan_time(hawk_hapset* hs, uint64_t pam_fwd, uint64_t pam_rev, uint32_t pamlen, uint32_t reps, float* avg_ms,
                       uint64_t* scanned_positions);

/* ---- fused search: search() (search_guides.py:510-548) + the CFDon slice of
 * scoring_guides() (scoring.py:352-387, after annotation.reverse_guides) ------------------- */
typedef struct {
  uint64_t pam_fwd, pam_rev; /* PAM.bits, PAM.bitsrc */
  uint32_t pamlen, guidelen;
  uint32_t right;            /* --right: guide downstream of the PAM (Cpf1-like) */
  uint32_t score_cfdon;      /* 1: compute CFDon for every kept guide (needs right == 0); a non-ACGT base under a
                                table lookup is HAWK_E_CFD, as the reference's KeyError (cfdscore.py:93-94).
                                2: the same, but such a guide scores NaN ("NA") - the opt-in for regions with N runs */
  const double* cfd_mm;      /* [20][4][4]: position, wildtype RNA base A,C,G,U, sgRNA base A,C,G,T */
  const double* cfd_pam;     /* [16]: PAM[-2:] dinucleotide, 4*b0+b1 over A,C,G,T */
} hawk_search_params;

typedef struct {          /* kernel times of the last hawk_search (HIP events on the ctx stream), ms */
  float count_ms;   /* k_search_count: scan + filters + redundancy classification -> tile counts + hand-over lists */
  float offsets_ms; /* k_mscan1-3: tile counts -> row offsets */
  float emit_ms;    /* k_emit_list + k_search_emit: finished rows (coordinates, windows, CFDon) */
  float total_ms;   /* first kernel start to last kernel end, including the host round trip for the row count */
  uint64_t scanned_positions; /* sum over haplotypes of (scan_stop - scan_start) */
  float emit_list_ms; /* k_emit_list alone (0 when the hand-over lists are switched off) */
  float v_count_ms;   /* a plan view (hawk_xplan_view): its count side alone (v_path 1: k_vsearch<0>; 2: k_cs_templates + k_cs_count) -
                         count_ms also holds the REF row's plane kernel */
  float v_emit_ms;    /* ... its emit side alone (k_vsearch<1> / k_rows_pack + k_cs_emit_rows) */
  float v_templates_ms; /* the cluster path of a view: k_cs_templates alone (v_count_ms: templates + k_cs_count) */
  uint32_t v_path;    /* 0: planes, 1: a view searched per dirty word (k_vsearch), 2: a view searched per distinct cluster (hawk_csearch.hip) */
  float v_emit_rows_ms; /* the cluster path: k_cs_emit_rows alone (v_emit_ms also holds k_rows_pack, which packs REF's staged rows) */
  float reserved;
} hawk_timing;

/* Runs the whole device pipeline; the guide table stays in HBM. `timing` may be NULL. */
int hawk_search(hawk_hapset* hs, const hawk_search_params* p, hawk_table** out, hawk_timing* timing);
void hawk_table_destroy(hawk_table* t);
/* n_rows: guides after remove_redundant_guides; n_candidates: PAM hits passing
 * is_pamhit_in_range (the bench metric's unit); n_hits: all PAM hits in scan range. */
int hawk_table_counts(const hawk_table* t, uint64_t* n_rows, uint64_t* n_candidates, uint64_t* n_hits);
/* Column download (each array n_rows long; any pointer may be NULL; destinations may be host or
 * device memory - the copy kind is inferred, so a table can be exported into buffers a
 * collective library owns without a host bounce).  Rows are haplotype-major; within a haplotype they are ordered by
 * (32 768-position tile, strand, position) - the plane kernels emit tile by tile - or, for a plan view searched per
 * distinct variant cluster (hawk_timing.v_path == 2), by (cluster along the row, strand, position);
 * sorting by (haplotype, strand, position) gives the reference's pre-dedup emission order either way.
 * Lifetime: the columns live in the haplotype set's workspace.  The next hawk_search (or hawk_hapset_set_meta) on
 * the same set overwrites them; from then on every call below on the older table returns HAWK_E_INVALID.
 *   pos: relative PAM position (what retrieve_guides iterates), start/stop: genomic
 *   (search_guides.py:260-280), strand 0/1, flags bit0 = a REF guide shares (start,strand),
 *   cfdon: NaN when no REF guide shares the key, win[5][n_rows]: bits [pos_window) of planes
 *   A,C,G,T,V for the guidelen+pamlen+20-nt window, bit 0 = leftmost base. */
int hawk_table_download(hawk_table* t, uint32_t* hap, uint32_t* pos, uint8_t* strand, int64_t* start, int64_t* stop,
                        uint8_t* flags, double* cfdon, uint64_t* win);
/* Layout of a table in HBM.  The plane kernels and the per-word search of a view write COLUMNS (the arrays above); the cluster
 * search of a plan view (hawk_timing.v_path == 2) writes PACKED ROWS: one array of 64-byte rows, a single linear write stream,
 *   word 0 pos | 1 strand + (flags << 1) + (haplotype row << 9) | 2 start - startp | 3 stop - start | 4-5 cfdon | 6 + 2p, 7 + 2p
 *   window slice of plane p (little-endian 32-bit words; startp = genomic position of the region string's base 0).
 * hawk_table_download hands out columns whichever layout the table has (a packed table is cut into the asked-for columns on the
 * device first); hawk_table_download_rows / hawk_table_device_rows give the packed rows as they lie (HAWK_E_UNSUPPORTED on a
 * columnar table), hawk_table_device_columns the column pointers (HAWK_E_UNSUPPORTED on a packed table). */
#define HAWK_LAYOUT_COLUMNS 0u
#define HAWK_LAYOUT_ROWS 1u
int hawk_table_layout(const hawk_table* t, uint32_t* layout, int64_t* startp);
int hawk_table_download_rows(hawk_table* t, void* rows64 /* n_rows x 64 bytes, host or device */);
int hawk_table_device_rows(hawk_table* t, void** rows64, int64_t* startp);
/* Device pointers of the same columns (valid until the next hawk_search on the same set);
 * plane p of the window slices starts at win + p * win_plane_stride (in uint64 elements). */
int hawk_table_device_columns(hawk_table* t, void** hap, void** pos, void** strand, void** start, void** stop,
                              void** flags, void** cfdon, void** win, uint64_t* win_plane_stride);

/* ---- f2: which rows the guide report merges.  Replaces the pandas groupby of
 * `_collapse_report_entries` (reports.py:958-1008; group columns chr, start, stop, sgRNA_sequence, pam,
 * strand, scores, gc_content, origin): two rows merge iff they agree in start, stop, strand, origin
 * (REF haplotype or not) and in the case-preserving spacer+PAM; scores and GC are functions of those.
 * hawk_table_collapse sorts and groups in HBM (results stay in the set's workspace until the next
 * search or collapse); hawk_table_collapse_download copies them out (host or device destinations,
 * any pointer may be NULL):
 *   perm[n_rows]         row indices ordered by (start, strand, group); rows of one group are contiguous
 *                        and keep table order (haplotype ascending)
 *   group_off[n_groups+1] CSR offsets into perm
 *   gc_num, gc_den[n_groups]  G/C/S and A/C/G/T/S/W base counts of the group's spacer: gc_content =
 *                        gc_num / gc_den (annotation.py:513-541 -> Biopython gc_fraction, ambiguous bases dropped)
 * Rows of one (start, strand) are told apart by 63 bits of a hash of everything compared (a false merge needs a 63-bit
 * collision inside one (start, strand): ~1e-12 over a whole-chromosome run); with HAWK_COLLAPSE_EXACT=1 in the
 * environment (read per call) the full keys are compared instead, at about twice the time. */
int hawk_table_collapse(hawk_table* t, uint64_t* n_groups, float* kernel_ms);
int hawk_table_collapse_download(hawk_table* t, uint32_t* perm, uint64_t* group_off, uint8_t* gc_num, uint8_t* gc_den);
/* The same grouping with the compared sequence widened by flank_up bases 5' and flank_down bases 3' of the guide (as it
 * reads after annotation.reverse_guides): 4 / 3 is the k-mer the model scorers see (scoring.py:50-67), so with Azimuth
 * or DeepCpf1 switched on rows that differ only in those flanks - and may score differently - stay separate, as the
 * reference's groupby over the score columns keeps them (reports.py:978-1003). */
int hawk_table_collapse_ex(hawk_table* t, uint32_t flank_up, uint32_t flank_down, uint64_t* n_groups, float* kernel_ms);
/* Group-level export after a collapse: per group its first member in table order (rep_row = that row's index, and the
 * row's columns: pos, strand, start, stop, flags, cfdon, win[5][n_groups]) and, for every row in group order (CSR by
 * group_off), the haplotype it came from (member_hap[n_rows]) - what the report needs (74 B per report row + 4 B per
 * guide row) without downloading the table. */
int hawk_table_collapse_export(hawk_table* t, uint32_t* rep_row, uint32_t* pos, uint8_t* strand, int64_t* start, int64_t* stop,
                               uint8_t* flags, double* cfdon, uint64_t* win, uint32_t* member_hap, float* kernel_ms);

/* Host-side helper of the report assembly: which of its haplotype's variants a guide SHOWS - annotation.polish_guide_variants
 * (annotation.py:185-284) - for many rows at once (the rows with an indel among their candidate variants; all-SNV rows are answered
 * in bulk by the caller).  Row k: cores[k] = the + strand spacer + PAM (L cased bytes), hap[k] its haplotype row whose position map
 * is segments [seg_start[h], seg_start[h + 1]) of (seg_rel, seg_gen), pivot[k] the guide's first position in the row, stop[k] its
 * genomic stop, cand_var[cand_off[k] .. cand_off[k + 1]) the candidate variants (ascending indices, no duplicates).  Variant v:
 * adjusted position t_pos[v], alleles in the ref / alt pools, name_rank[v] = rank of its id in string order.  out_var (capacity
 * cand_off[n]) receives per row the variants shown, in id order, out_off their CSR offsets; need_python[k] = 1 where the
 * reference's own assertion (annotation.py:185-189) would fire - the caller's Python mirror then raises as the reference does. */
int hawk_host_polish_rows(uint64_t n, uint32_t L, const uint8_t* cores, const uint32_t* hap, const int64_t* pivot, const int64_t* stop,
                          const uint64_t* cand_off, const uint32_t* cand_var, const uint64_t* seg_start, const uint32_t* seg_rel,
                          const int64_t* seg_gen, uint64_t n_haps, const int64_t* t_pos, const uint8_t* ref_pool, const uint64_t* ref_off,
                          const uint8_t* alt_pool, const uint64_t* alt_off, uint32_t n_var, const uint32_t* name_rank, uint64_t* out_off,
                          uint32_t* out_var, uint8_t* need_python);

/* The same for EVERY alt row of a report, candidates found here as well (annotation.py:246-284: the variants of the guide's haplotype
 * whose adjusted position lies in [start, max(stop, start + L)]).  Haplotype row h lists its variants - indices into the variant
 * table - at var_idx[var_off[h] .. var_off[h + 1]), ascending by position.  hawk_host_variant_window: per row k the window
 * [first[k], first[k] + count[k]) of its haplotype's list with p_lo[k] <= position <= p_hi[k]  (HAWK_E_UNSUPPORTED when a list is not in
 * position order: the caller keeps its own route).  hawk_host_polish_windows: hawk_host_polish_rows with row k's candidates read from
 * var_idx[first[k] ..), cand_off[k + 1] - cand_off[k] of them (sorted and made unique here); out_var has capacity cand_off[n]. */
int hawk_host_variant_window(uint64_t n, const uint32_t* hap, const int64_t* p_lo, const int64_t* p_hi, const uint64_t* var_off,
                             const int64_t* var_idx, const int64_t* t_pos, uint64_t n_haps, uint32_t n_var, uint64_t* first, uint32_t* count);
int hawk_host_polish_windows(uint64_t n, uint32_t L, const uint8_t* cores, const uint32_t* hap, const int64_t* pivot, const int64_t* stop,
                             const uint64_t* first, const uint64_t* cand_off, const int64_t* var_idx, uint64_t n_idx, const uint64_t* seg_start,
                             const uint32_t* seg_rel, const int64_t* seg_gen, uint64_t n_haps, const int64_t* t_pos, const uint8_t* ref_pool,
                             const uint64_t* ref_off, const uint8_t* alt_pool, const uint64_t* alt_off, uint32_t n_var, const uint32_t* name_rank,
                             uint64_t* out_off, uint32_t* out_var, uint8_t* need_python);

/* The line index of a VCF text held in (mapped) memory - replaces the streaming index pass of readers.VCF (variant.py:622-708: the
 * reference opens the file through pysam / tabix): per line its start, POS (-1 header or empty line, -2 malformed record), where
 * the sample columns begin (0: fewer than nine tabs) and the length of its CHROM field; line_start has n_lines + 1 entries.  The
 * text must end with a newline.  HAWK_E_CAPACITY (only *n_lines written) when cap < lines.  Multi-threaded, no device work. */
int hawk_host_vcf_index(const uint8_t* text, uint64_t len, uint64_t cap, uint64_t* line_start, int64_t* pos, uint64_t* gt_off,
                        uint32_t* chrom_len, uint64_t* n_lines, uint32_t* multi_contig);

/* The guide report written as TSV text straight to its file (what _store_report does with DataFrame.to_csv(sep="\t", index=False),
 * reports.py:739) from COLUMNS as the assembly keeps them, so that no row string and no text column ever exists in Python - a C3
 * report is 0.64 GB of text.  Output row i is source row order[i] (order == NULL: i), fields separated by tabs, rows ended by a
 * newline; `header` (with its newline) goes first.  The caller guarantees that no field holds a tab, quote or line break (csv's
 * minimal quoting would then apply; reports.write_report_tsv checks the vocabularies and falls back to pandas).  Column kinds:
 *   HAWK_TSV_CONST   data = the text (width bytes), the same in every row
 *   HAWK_TSV_FIXED   data = [n_rows][width] bytes
 *   HAWK_TSV_RAGGED  data = bytes, off = [n_rows + 1] offsets into them
 *   HAWK_TSV_INT64   data = int64[n_rows], written in decimal
 *   HAWK_TSV_VOCAB   data = uint32[n_rows] indices into n_vocab strings: pool = their bytes, off = [n_vocab + 1] offsets
 * Multi-threaded, two passes (row lengths, then bytes) into a shared mapping of the file (plain writes where mapping fails). */
#define HAWK_TSV_CONST 0u
#define HAWK_TSV_FIXED 1u
#define HAWK_TSV_RAGGED 2u
#define HAWK_TSV_INT64 3u
#define HAWK_TSV_VOCAB 4u
typedef struct {
  uint32_t kind, width;
  const void* data;
  const uint64_t* off;
  const void* pool;
  uint64_t n_vocab;
} hawk_tsv_col;
int hawk_host_tsv_write(const char* path, const char* header, uint64_t header_len, uint64_t n_rows, const uint64_t* order, uint32_t n_cols,
                        const hawk_tsv_col* cols, uint64_t* bytes_out);

/* Host-side helper of the report assembly (no device work): the report's `samples` / `haplotype_id` columns list every
 * carrier of every report row (reports.py:767-857) - a ragged join of label strings.  Items item_label[group_off[g] ..
 * group_off[g+1]) of group g name byte strings pool[pool_off[l] .. pool_off[l+1]); the group's text is those strings
 * joined by `sep`.  out_off[n_groups+1] always receives the groups' byte offsets; with out == NULL only sizes are
 * computed; HAWK_E_CAPACITY when out_cap is too small (required size in out_off[n_groups]). */
int hawk_host_ragged_join(const uint32_t* item_label, const uint64_t* group_off, uint64_t n_groups, const uint8_t* pool,
                          const uint64_t* pool_off, uint64_t n_labels, uint8_t sep, uint8_t* out, uint64_t out_cap,
                          uint64_t* out_off);

/* The same for a group's SET of items: haplotype h carries the items hap_item[hap_item_off[h] .. hap_item_off[h+1]) (indices
 * into a label pool held in sort order); per group the sorted unique items of its members member_hap[member_off[g] ..
 * member_off[g+1]) are joined - collapse_haplotype_ids (reports.py:845-857) and the unphased collapse_samples. */
int hawk_host_group_join(const uint64_t* member_off, const uint32_t* member_hap, uint64_t n_groups, const uint64_t* hap_item_off,
                         const uint32_t* hap_item, uint64_t n_haps, const uint8_t* pool, const uint64_t* pool_off, uint64_t n_labels,
                         uint8_t sep, uint8_t* out, uint64_t out_cap, uint64_t* out_off);
/* The phased `samples` column (reports.py:767-810): per group one `name:a|b` per sample (samples numbered in the order of
 * their first entry in the sorted entry list) with the per-copy maxima over the members' entries.  ent_ok[e] = 0 marks an
 * entry that is not `name:int|int`; group_flags[g] bit 0: the group holds a well-formed entry, bit 1: it holds one that is
 * not; a group with none well-formed gets its sorted unique entries joined as they are (strings in ent_pool), one mixing
 * both kinds is left empty for the caller to resolve. */
int hawk_host_group_samples(const uint64_t* member_off, const uint32_t* member_hap, uint64_t n_groups, const uint64_t* hap_ent_off,
                            const uint32_t* hap_ent, uint64_t n_haps, const uint32_t* ent_sample, const uint16_t* ent_a1,
                            const uint16_t* ent_a2, const uint8_t* ent_ok, uint64_t n_entries, const uint8_t* name_pool,
                            const uint64_t* name_off, uint64_t n_samples, const uint8_t* ent_pool, const uint64_t* ent_pool_off,
                            uint8_t* out, uint64_t out_cap, uint64_t* out_off, uint8_t* group_flags);
/* Position-map segments (haplotype.py:90-159 as unit-slope segments) of all rows of an expansion from its carried
 * indels (hawk_gt_lists_indels): CSR seg_start[n_rows + 1] (always written: call with cap = 0 for the size), seg_rel,
 * seg_gen.  Rows aliasing another row keep the identity segment only. */
int hawk_host_build_segments(const uint32_t* indel_entry, uint64_t n_indel, const uint32_t* hv_idx, const int32_t* hv_o,
                             const uint64_t* hv_off, uint32_t n_rows, const int64_t* var_r0, const int64_t* var_chain, int64_t startp,
                             const uint32_t* hap_len, const int64_t* alias, uint64_t* seg_start, uint32_t* seg_rel, int64_t* seg_gen,
                             uint64_t cap);
/* posmap_rev[g] (haplotype.py:159: the last relative position whose genomic position is g; -1 where g is deleted) of every
 * row of such a segment table at once: what compute_scan_start_stop (search_guides.py:49-84) looks up per haplotype. */
int hawk_host_posmap_rev(const uint64_t* seg_start, const uint32_t* seg_rel, const int64_t* seg_gen, const uint32_t* hap_len,
                         uint32_t n_rows, int64_t g, int64_t* out);

/* ---- the one exchange of a multi-GPU job.  One process per GPU, haplotypes block-partitioned with REF
 * on every rank (search_guides.py:111-131, 530-547 loop over independent haplotypes), no collective on the search
 * path; afterwards every rank's guide table goes to one rank over RCCL / xGMI.  The caller moves the 128-byte id from
 * rank 0 to the others (crisprhawk_hip/parallel.py does it over TCP).  hawk_table_gather: all-gather of the row
 * counts, then grouped ncclSend / ncclRecv of the columns straight from the tables' device memory; on `dst` *merged
 * is a table that owns its columns (download / device_columns / destroy), haplotype indices moved by each rank's
 * hap_offset (local row 0 = REF stays 0).  hawk_comm_gatherv: variable-length byte gather of host or device buffers
 * (recv_off[world + 1] byte offsets, needed on dst only). */
typedef struct hawk_comm hawk_comm;
int hawk_comm_unique_id(uint8_t* id128);
int hawk_comm_init(hawk_ctx* ctx, int world, int rank, const uint8_t* id128, hawk_comm** out);
void hawk_comm_destroy(hawk_comm* c);
const char* hawk_comm_last_error(void);
int hawk_comm_allgather_u64(hawk_comm* c, const uint64_t* mine, uint32_t k, uint64_t* all);
int hawk_comm_gatherv(hawk_comm* c, const void* send, uint64_t send_bytes, int send_on_device, void* recv,
                      const uint64_t* recv_off, int recv_on_device, int dst);
int hawk_table_gather(hawk_comm* c, hawk_table* t, uint32_t hap_offset, int dst, hawk_table** merged, float* ms);
/* The arithmetic of hawk_table_gather as a host function, free of device and RCCL calls (it runs without a GPU): from the
 * directory all ranks contributed - dir4[4 r .. 4 r + 3] = {rows, haplotype offset, candidates, hits} of rank r - the row
 * offset of every rank's slice in the merged table (row_off[world + 1]), the totals {rows, candidates, hits} and this
 * rank's transfers.  Columns are numbered hap pos strand start stop flags cfdon win[0..4] (HAWK_GATHER_COLS).  A sender gets
 * one op per column {col, dst, 0, bytes}; the destination one per column and rank {col, rank, byte offset into the merged
 * column, bytes}, its own slice included (a local copy).  ops == NULL only counts. */
#define HAWK_GATHER_COLS 12
typedef struct { uint32_t col, peer; uint64_t offset, bytes; } hawk_gather_op;
int hawk_host_gather_plan(int world, int rank, int dst, const uint64_t* dir4, uint64_t* row_off, uint64_t* totals3,
                          hawk_gather_op* ops, uint32_t cap, uint32_t* n_ops);

/* ---- f3: VCF sample columns -> allele codes -> carried-variant lists.  Replaces the per-sample Python work of
 * VariantRecord.read_vcf_line -> _genotypes_to_samples (variant.py:286-311, 558-619) and the inversion into
 * per-sample variant lists of compute_haplotypes_phased (haplotypes.py:132-159).
 * hawk_gt_parse: text = the raw bytes of n_lines VCF records ('\n'-terminated), record i =
 *   text[line_off[i], line_off[i+1]), its first sample column starts at gt_off[i] (host arrays).  The device
 *   keeps codes[n_lines][2*n_samples]: allele carried by copy 0 / copy 1 of each sample (0 REF, k = k-th ALT,
 *   255 missing or absent), and a flag byte per record: 1 = a genotype without '|' (unphased or haploid),
 *   2 = field count != n_samples, 4 = unexpected character.  hawk_gt_codes downloads both (NULL skips).
 * hawk_gt_lists: variants j = (record var_line[j], allele var_allele[j] >= 1), ascending; var_r0[j] = offset of
 *   the variant in the reference region, var_chain[j] = alt length - replaced length.  Per chromosome copy
 *   (column c = 2*sample + copy) the ascending list of carried variants is built on the device; col_off[2*n_samples+1]
 *   (host) receives the CSR offsets, col_delta (host, may be NULL) the summed length change per column.
 *   hawk_gt_lists_download copies hv_idx and hv_o = var_r0 + exclusive running sum of var_chain within the
 *   column (host or device destinations) - the inputs of hawk_hapset_expand for the rows "columns with a
 *   non-empty list, in column order". */
int hawk_gt_parse(hawk_ctx* ctx, const uint8_t* text, uint64_t text_len, const uint64_t* line_off, const uint64_t* gt_off,
                  uint64_t n_lines, uint32_t n_samples, hawk_gt** out, float* kernel_ms);
/* The same object from an allele-code matrix the caller already holds (codes[n_lines][2*n_samples], host): genotypes that
 * never were VCF text (an in-memory panel) enter the device inversion (hawk_gt_lists) without a host-side nonzero scan. */
int hawk_gt_from_codes(hawk_ctx* ctx, const uint8_t* codes, uint64_t n_lines, uint32_t n_samples, hawk_gt** out);
void hawk_gt_destroy(hawk_gt* g);
int hawk_gt_codes(hawk_gt* g, uint8_t* codes, uint8_t* line_flags);
int hawk_gt_lists(hawk_gt* g, const uint32_t* var_line, const uint8_t* var_allele, const int32_t* var_r0, const int32_t* var_chain,
                  uint32_t n_var, uint64_t* col_off, int64_t* col_delta, float* kernel_ms);
int hawk_gt_lists_download(hawk_gt* g, uint32_t* hv_idx, int32_t* hv_o);
/* The entries of those lists whose variant changes the haplotype's length (var_chain != 0), as ascending entry indices:
 * what the position-map segments are built from (haplotype.py:90-159) without a pass over all entries on the host.
 * *n_indel is always set; up to cap indices are copied (cap = 0 / NULL: ask for the number first). */
int hawk_gt_lists_indels(hawk_gt* g, uint32_t* entry_idx, uint64_t cap, uint64_t* n_indel);

/* ---- K7: off-target enumeration, replacing the external `crispritz.py search ... -mm M -bDNA 0
 * -bRNA 0` of offtargets.py:222-293 (CRISPRitz 2.6.6 is a third-party binary the reference shells
 * out to; semantics restated from the call site and the consumed fields, offtarget.py:77-101).
 * The genome is a hawk_hapset whose rows are equal-sized contig pieces (made on the host, each
 * piece overlapping the next by guidelen+pamlen-1 bases; scan_start/scan_stop of a row = the
 * window starts it owns).  hawk_genome_finalize() turns the planes one-hot once after packing.
 * guides2: one 64-bit word per guide, spacer base i (5'->3') at bits 2i,2i+1 with A0 C1 G2 T3.
 * A hit = window whose PAM positions hold unambiguous bases inside the PAM's IUPAC sets and whose
 * spacer has <= max_mm mismatches (ambiguous genome bases count as mismatches), either strand.
 * Outputs (cap entries each, any may be NULL): guide index, row, window start in the row, strand,
 * mismatches, the window in guide orientation as a 2-bit code, its ambiguity mask.  Unordered.
 * If more than cap hits exist returns HAWK_E_CAPACITY with *n_out = required. */
typedef struct { uint64_t pam_fwd, pam_rev; uint32_t pamlen, guidelen, right, max_mm; } hawk_ot_params;
typedef struct { float scan_ms, sites_ms, match_ms, total_ms; uint64_t n_sites, scanned_positions; } hawk_ot_timing;
int hawk_genome_finalize(hawk_hapset* rows);
int hawk_offtarget_scan(hawk_hapset* rows, const hawk_ot_params* p, const uint64_t* guides2, uint32_t n_guides,
                        uint32_t* out_guide, uint32_t* out_row, uint32_t* out_q, uint8_t* out_strand, uint8_t* out_mm,
                        uint64_t* out_code, uint32_t* out_nmask, uint64_t cap, uint64_t* n_out, hawk_ot_timing* timing);

/* ---- K4 stand-alone: compute_cfd() (scores/cfdscore/cfdscore.py:53-95) on n triples --------
 * wt / sg: n spacers of `len` characters each (host, contiguous, any case, T or U); pam2: n
 * two-character strings (the caller passes guide.pam[-2:], crisprhawk_scores.py:84-86).
 * out[i] = prod over i<min(len,20), wt!=sg of mm[i][wt][sg] times pam[pam2], fp64, left to
 * right.  Any non-ACGT base under a lookup -> HAWK_E_CFD (the reference's KeyError). */
int hawk_cfd(hawk_ctx* ctx, const char* wt, const char* sg, uint32_t len, const char* pam2, uint64_t n,
             const double* cfd_mm, const double* cfd_pam, double* out);

/* ---- K6: Seq-DeepCpf1 (scores/deepCpf1/seqdeepcpf1.py:22-92; wrapper scores/crisprhawk_scores.py:
 * 90-107): n 34-mers (host, contiguous, ACGT any case) -> n fp32 scores.  `weights` is one packed
 * fp32 block in the torch layout of SeqDeepCpf1: conv.weight[80][4][5], conv.bias[80],
 * fc1.weight[80][1200], fc1.bias[80], fc2.weight[40][80], fc2.bias[40], fc3.weight[40][40],
 * fc3.bias[40], output.weight[1][40], output.bias[1].  A non-ACGT base -> HAWK_E_IUPAC (the
 * reference's KeyError, seqdeepcpf1.py:91). */
#define HAWK_DEEPCPF1_NPARAMS (1600 + 80 + 96000 + 80 + 3200 + 40 + 1600 + 40 + 40 + 1)
int hawk_deepcpf1(hawk_ctx* ctx, const char* seqs34, uint64_t n, const float* weights, float* out);

/* ---- K5: Azimuth / Rule Set 2 (scoring.py:87-193 -> scores/azimuth/model_comparison.py:507-585):
 * n 30-mers (4 nt + 20-nt guide + NGG + 3 nt) -> fp64 predictions of a gradient-boosted regression
 * tree ensemble over the 627 features of features/featurization.py (order-1/2 position-dependent
 * and -independent nucleotide features, GC features, NGGX, four nearest-neighbour melting
 * temperatures).  The model is the flattened form of the sklearn GradientBoostingRegressor the
 * reference unpickles: per tree a node range [tree_off[t], tree_off[t+1]); node k is a leaf iff
 * feature[k] < 0, else x[feature[k]] (as float32) <= threshold[k] goes to left[k] else right[k]
 * (indices relative to the tree); prediction = init + learning_rate * sum of leaf values.
 * feats_out (optional, n*627 doubles) receives the feature matrix.  Non-ACGT -> HAWK_E_IUPAC. */
typedef struct {
  uint32_t n_trees, n_nodes;
  const int32_t* tree_off; /* n_trees + 1 */
  const int32_t* feature;  /* n_nodes */
  const int32_t* left;
  const int32_t* right;
  const double* threshold;
  const double* value;
  double init, learning_rate;
} hawk_gbt_model;
/* The melting temperature behind Azimuth's four Tm features on its own - Biopython's MeltingTemp.Tm_NN with its defaults
 * (featurization.py:358-397 calls it on the 30-mer and three of its slices) - for n sequences of `len` <= 32 bases each,
 * A/C/G/T only (HAWK_E_IUPAC otherwise).  The same device function k_azimuth uses; tests hold it to the value Biopython documents. */
int hawk_tm_nn(hawk_ctx* ctx, const char* seqs, uint32_t len, uint64_t n, double* out);
int hawk_azimuth(hawk_ctx* ctx, const char* seqs30, uint64_t n, const hawk_gbt_model* model, double* out, double* feats_out);

/* The same tree evaluator over a feature matrix the caller supplies (feats[n][n_features], host, row-major doubles):
 * the device half of RS3 (scoring.py:196-300 -> rs3.seq.predict_seq, scores/crisprhawk_scores.py:47-62: a LightGBM
 * model over sglearn features; the third-party featuriser stays on the caller's side, the exported LightGBM text
 * model is flattened into hawk_gbt_model by crisprhawk_hip/scoring.py).  cast_f32 = 1 compares float32(x) as sklearn
 * does, 0 the double itself as LightGBM does. */
int hawk_gbt_predict(hawk_ctx* ctx, const double* feats, uint64_t n, uint32_t n_features, const hawk_gbt_model* model, int cast_f32,
                     double* out);

#ifdef __cplusplus
}
#endif
#endif /* HAWK_H */
