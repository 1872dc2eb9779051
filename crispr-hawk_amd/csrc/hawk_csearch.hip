// hawk_csearch.hip — the guide search from an expansion plan, once per DISTINCT variant cluster instead of once per row.
//
// A non-REF haplotype row contributes guide rows only where a window touches an alt allele (search_guides.py:468-471), and
// what those rows look like - PAM hits, coordinates, the REF partner, the padded window, CFDon - depends on nothing but
// the carried variants within reach of the window: two variants interact only if fewer than CL_LINK = 64 positions
// separate their alleles (a window's padded slice spans guidelen + pamlen + 2 x 10 <= 64 positions).  So a row's records
// fall into CLUSTERS (maximal runs with gaps <= CL_LINK), and a cluster carried by many chromosome copies - a common SNV
// with no neighbour - yields the same rows in every one of them, shifted by the copies' upstream indels.  On the C3 panel
// 8.8 x 10^6 cluster instances are 7.3 x 10^4 distinct clusters.
//
//   dictionary (once per plan; hawk_xplan_view):  k_cl_chunks / k_cl_count / k_cl_fill cut every row's records into cluster instances.
//       An instance of ONE record whose windows can meet no bound of its row IS its variant: its distinct cluster is the variant's
//       index in the plan's table - nothing to look up, nothing to compare.  Every other instance with a cluster goes on a list and
//       through a hash table of the variant identities (k_cl_enter), and is compared record by record with the instance that
//       opened its cluster (k_cl_uid: exactly - same hash is not same cluster until then); those clusters are numbered behind the
//       variants.  An instance whose windows could meet a row-specific bound (scan range, row ends) is a cluster of its own; one
//       wholly outside the scan range carries no cluster at all.
//   per search:  k_cs_templates builds each distinct cluster's rows ONCE, on its representative row, with the string
//       builder, PAM match, filters and classification of hawk_vsearch.hip (64-byte template rows, strand 0 / strand 1
//       regions in position order); k_cs_count gives every instance its row count and adds up the job's totals
//       (candidates, hits: the cluster's own + REF's hits under the shift of the clean stretch in front of it) and adds a wave's
//       rows up (the offset scan runs over one entry per 64 instances); k_cs_emit_rows then copies template rows into the guide
//       table - packed 64-byte rows, one linear write stream - a wave per 256 consecutive instances, patching haplotype row
//       and position.
// The table holds the same rows as hawk_search on the materialised planes; their order within a haplotype is (cluster,
// strand, position) instead of (tile, strand, position) - GuideTable.emission_order() sorts either into the reference's.
// The REF row goes through the plane kernels as before.
#include "hawk_vc.h"

#define CL_LINK 64
#define CL_NONE 0xffffffffu
#define CL_FAR (1 << 29)     // "position" of a row's closing instance: the clean run in front of it reaches the row's end
#define CL_MAXWALK 4096      // records per cluster the dictionary accepts (longer chains: the per-word search takes the plan)
#define CS_G 8               // lanes per distinct cluster in k_cs_templates: one 32-window word each per round

// a template row, 64 bytes = one L2 sector pair: {position relative to the cluster's first allele, strand | flags << 1,
// start - REF's first position, stop - start} {cfdon, win0} {win1, win2} {win3, win4}
struct __attribute__((aligned(16))) CsRow { uint4 a, b, c, d; };
static_assert(sizeof(CsRow) == 64, "template row layout");
size_t hawk_cs_row_bytes() { return sizeof(CsRow); }

__device__ __forceinline__ uint64_t cl_mix(uint64_t h, uint64_t v) {
  h = (h ^ v) * 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}
// What the dictionary passes read of a record: {o, rs, alt_len} and WHICH variant it is (its index in the plan's variant table, from
// which hawk_expand.hip took rs, alt_len and the alt bases - so two records of one variant are the same allele at the same place in
// REF, by construction).  The plan keeps these 16 bytes as an array of their own (hawk_launch_hx_heads, at plan creation): the
// passes over every record are HBM-bound, and a 32-byte record fetched for half its bytes is twice the traffic.
struct __attribute__((aligned(16))) HxHead { int32_t o; uint32_t rs; uint32_t alt_len; uint32_t var; };
static_assert(sizeof(HxHead) == 16, "record head layout");
__global__ __launch_bounds__(256) void k_hx_heads(const HxVar* __restrict__ recs, const uint32_t* __restrict__ hv_idx, uint64_t n, uint4* __restrict__ heads) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint4 r = *reinterpret_cast<const uint4*>(recs + j);
  heads[j] = make_uint4(r.x, r.y, r.z, hv_idx[j]);
}
void hawk_launch_hx_heads(hipStream_t st, const void* recs, const uint32_t* hv_idx, uint64_t n, void* heads) {
  if (n) hipLaunchKernelGGL(k_hx_heads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const HxVar*>(recs), hv_idx, n,
                            static_cast<uint4*>(heads));
}
__device__ __forceinline__ bool cl_starts(const HxHead* __restrict__ recs, uint64_t j, uint64_t lo) {
  if (j == lo) return true;
  const int32_t prev_end = recs[j - 1].o + (int32_t)recs[j - 1].alt_len;
  return recs[j].o - prev_end > CL_LINK;
}

// ---- dictionary -----------------------------------------------------------------------------------
// The rows' records are walked in CHUNKS of CL_CHUNK consecutive records of one row (a row of C3 has two): one workgroup per chunk
// in the two passes over the records, so that the grid is ten thousand equal pieces of work and not five thousand loops.
// Instances per row: its clusters + one closing instance (the clean run behind the last cluster); REF and rows that scan nothing
// (collapsed onto another row) have none - and no chunk.
#define CL_ROW_U 4                  // records per thread in the two passes over the records: their loads are in flight together
#define CL_CHUNK (256 * CL_ROW_U)
// rows -> chunks, one workgroup: chunks per row, their offsets (ch_off[n_rows + 1]) and every chunk's row
__global__ __launch_bounds__(1024) void k_cl_chunks(const uint64_t* __restrict__ hv_off, const uint8_t* __restrict__ is_ref, const int32_t* __restrict__ ss,
                                                    const int32_t* __restrict__ se, uint32_t n_rows, uint32_t* __restrict__ ch_off,
                                                    uint32_t* __restrict__ ch_row) {
  __shared__ uint32_t s_w[1024 / WAVE];
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 1024) {
    const uint32_t row = r0 + threadIdx.x;
    uint32_t nch = 0;
    if (row < n_rows) {
      const uint64_t len = hv_off[row + 1] - hv_off[row];
      const bool dead = is_ref[row] || se[row] <= ss[row];
      nch = dead ? 0u : (len ? (uint32_t)((len + CL_CHUNK - 1) / CL_CHUNK) : 1u);
    }
    uint32_t tot;
    const uint32_t ex = carry + block_excl_scan<1024 / WAVE>(nch, s_w, &tot);
    if (row < n_rows) {
      ch_off[row] = ex;
      for (uint32_t c = 0; c < nch; ++c) ch_row[ex + c] = row;
    }
    carry += tot;
  }
  if (threadIdx.x == 0) ch_off[n_rows] = carry;
}
struct ClInst {  // per instance (k_cl_fill)
  int32_t* o;      // row position of the cluster's first allele (closing instance: CL_FAR)
  uint32_t* row;
  int32_t* pa;     // where the clean run in front of the instance starts: end of the previous record's allele, 0 at the row's start
  int32_t* rb;     // REF shift of that run
  uint32_t* uid;   // its distinct cluster (CL_NONE: none)
};
struct ClUniq {  // per distinct cluster: first record, records, row and row position of the instance that describes it; k_cl_describe adds the rest
  uint32_t* rec; uint32_t* n; uint32_t* row; int32_t* o; uint32_t* seg; uint32_t* span2;
};
#define CL_PENDING 0x80000000u
// Nine clusters in ten are ONE record of a shareable instance: such a cluster IS its variant, and its number is the variant's
// index - no table, no comparison.  Which instance describes it does not matter: every such instance whose variant the first rows
// have not described yet writes {its record, its row} - ONE aligned 64-bit store, so whichever write lands last is whole.
// Every other instance with a cluster - several records, or windows that could meet a bound of its row - is put on a LIST by the
// cutting pass (its place known without an atomic: the counting pass counts those too) and goes through a hash table of 32-byte
// slots in passes of their own over that list: {hash key (0: free), -} for the pass that enters them, {cluster number + 1 (0: not yet
// numbered), records | class << 16 of the instance that opened the slot, REF position of its first allele, its first variant} -
// ONE 16-byte piece, one request - for the pass that compares; these are numbered behind the variants, in the order they are met
// (the order carries no meaning: every table a search writes is in instance order).
// Why this shape: the passes are bound by the NUMBER of requests to L2 (a 4-byte gather costs what a 64-byte line costs: ~1.3 x 10^11
// requests/s at best) and by same-address atomics (~10 ns each, whoever waits for them) - not by bytes and not by arithmetic.
struct __attribute__((aligned(32))) ClSlot { unsigned long long key; unsigned long long pad; uint32_t up1, ncls, refp, var; };
static_assert(sizeof(ClSlot) == 32, "table slot layout");
size_t hawk_cl_slot_bytes() { return sizeof(ClSlot); }
struct __attribute__((aligned(32))) ClListed {  // a listed instance: all the table passes need of it
  uint32_t inst, key_lo, key_hi, rec, ncls, refp; int32_t o_first; uint32_t var;
};
static_assert(sizeof(ClListed) == 32, "list entry layout");
size_t hawk_cl_listed_bytes() { return sizeof(ClListed); }

// One cluster start: the walk over its records and what follows from it.  r0 / rp: the record at j and the one before it; nx1 / nx2:
// the records of lanes + 1, + 2 (a cluster's second and third record are the records of the lanes above - nine clusters in ten are
// one record, ninety-nine in a hundred at most three: no dependent load until a cluster is longer or crosses the wave's end).
struct ClCut { int32_t o_first, o_end, pa, rb; uint32_t n, cls; unsigned long long key; bool run_inside, too_long; };
__device__ __forceinline__ ClCut cl_cut(const HxHead* __restrict__ recs, uint64_t lo, uint64_t hi, uint64_t j, const uint4& r0, const uint4& rp,
                                         const uint4& nx1, const uint4& nx2, uint32_t lane, int32_t ss, int32_t se, int32_t hl, uint32_t row) {
  ClCut c;
  // A cluster's identity is WHERE in REF its first allele starts and, record by record, which variant it is (and so which allele,
  // where it lies and where REF resumes behind it) and where it lies relative to the first - hashed here, compared record by
  // record before two instances share a cluster (k_cl_uid).
  c.o_first = (int32_t)r0.x; c.pa = 0; c.rb = 0; c.too_long = false;
  if (j > lo) { c.pa = (int32_t)rp.x + (int32_t)rp.z; c.rb = (int32_t)rp.y - c.pa; }
  uint64_t e = j + 1, h = cl_mix(0x243F6A8885A308D3ull, (uint64_t)(uint32_t)(c.o_first + c.rb));
  h = cl_mix(h, (uint64_t)r0.w | ((uint64_t)r0.z << 32));
  h = cl_mix(h, (uint64_t)r0.y);
  c.n = 1;
  c.o_end = (int32_t)r0.x + (int32_t)r0.z;  // end of the cluster's last allele so far
  bool open = e < hi;                        // the record at e may still belong to the cluster
  while (open) {
    const uint32_t ahead = (uint32_t)(e - j);
    uint4 r;
    if (ahead == 1 && lane + 1 < WAVE) r = nx1;
    else if (ahead == 2 && lane + 2 < WAVE) r = nx2;
    else r = *reinterpret_cast<const uint4*>(recs + e);
    if ((int32_t)r.x - c.o_end > CL_LINK) break;      // it starts the next cluster
    if (c.n >= CL_MAXWALK) { c.too_long = true; break; }  // a chain too long for this path
    h = cl_mix(h, (uint64_t)r.w | ((uint64_t)r.z << 32));
    h = cl_mix(h, (uint64_t)r.y | ((uint64_t)(uint32_t)((int32_t)r.x - c.o_first) << 32));
    c.o_end = (int32_t)r.x + (int32_t)r.z;
    ++c.n; ++e;
    open = e < hi;
  }
  // window starts the cluster can touch: [o_first - (L - 1), o_end), L <= 44; the ranges they are tested against
  // (search_guides.py:49-84, 395-420) are [ss - po, se - po) and [PAD, len - L - PAD], po in {0, guidelen}
  const bool outside = c.o_end <= ss - 44 || c.o_first - 43 >= se;
  const bool interior = c.o_first - 43 >= (ss > HAWK_PAD ? ss : HAWK_PAD) && c.o_first >= 64 && c.o_end <= se - 44 &&
                        c.o_end <= hl - 44 - HAWK_PAD + 1 && c.o_end + 128 <= hl;
  c.key = h; c.cls = 1;
  if (outside) { c.cls = 0; c.key = 0; }
  else if (!interior) { c.cls = 2; c.key = cl_mix(h ^ 0xA4093822299F31D0ull, (uint64_t)row + 1u); }
  if (c.cls && c.key == 0) c.key = 1;
  // the clean run in front, [pa, o_first - (L - 1)), lies inside every range of both strands for every geometry: its count
  // then needs neither the row nor its bounds (bit 31 of the instance's pa)
  c.run_inside = c.pa >= (ss > HAWK_PAD ? ss : HAWK_PAD) && c.o_first <= se - 44 && c.o_first <= hl - 44 - HAWK_PAD + 1;
  return c;
}
__device__ __forceinline__ uint4 shfl_down4(const uint4& v, int d) {
  return make_uint4((uint32_t)__shfl_down((int)v.x, d), (uint32_t)__shfl_down((int)v.y, d), (uint32_t)__shfl_down((int)v.z, d),
                    (uint32_t)__shfl_down((int)v.w, d));
}

// chunk b: the instances it opens (its cluster starts + the closing instance in a row's last chunk) and how many of them go on the list
__global__ __launch_bounds__(256) void k_cl_count(const HxHead* __restrict__ recs, const uint64_t* __restrict__ hv_off, const uint32_t* __restrict__ hap_len,
                                                  const int32_t* __restrict__ ss_, const int32_t* __restrict__ se_, const uint32_t* __restrict__ ch_off,
                                                  const uint32_t* __restrict__ ch_row, uint32_t n_rows, uint32_t n_var, uint32_t* __restrict__ cnt,
                                                  uint32_t* __restrict__ lcnt) {
  __shared__ uint32_t s_w[256 / WAVE];
  const uint32_t b = blockIdx.x;
  if (b >= ch_off[n_rows]) return;  // (launched over the bound on the chunks: their number is only known on the device; workgroup-uniform)
  const uint32_t row = ch_row[b];
  const uint64_t lo = hv_off[row], hi = hv_off[row + 1];
  const uint64_t b0 = lo + (uint64_t)(b - ch_off[row]) * CL_CHUNK;
  const int32_t ss = ss_[row], se = se_[row], hl = (int32_t)hap_len[row];
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  uint32_t c = 0;  // starts | listed << 16
  if (hi > lo) {
    uint4 r0[CL_ROW_U], rp[CL_ROW_U];
#pragma unroll
    for (int u = 0; u < CL_ROW_U; ++u) {
      const uint64_t j = b0 + u * 256 + threadIdx.x;
      const uint64_t jc = j < hi ? j : hi - 1;
      r0[u] = *reinterpret_cast<const uint4*>(recs + jc);
      rp[u] = *reinterpret_cast<const uint4*>(recs + (jc > lo ? jc - 1 : jc));
    }
    auto one = [&](const int u) __attribute__((always_inline)) {
      const uint4 nx1 = shfl_down4(r0[u], 1), nx2 = shfl_down4(r0[u], 2);  // (every lane takes part in the exchange)
      const uint64_t j = b0 + u * 256 + threadIdx.x;
      if (!(j < hi && (j == lo || (int32_t)r0[u].x - ((int32_t)rp[u].x + (int32_t)rp[u].z) > CL_LINK))) return;
      const ClCut k = cl_cut(recs, lo, hi, j, r0[u], rp[u], nx1, nx2, lane, ss, se, hl, row);
      const bool simple = k.n == 1 && k.cls == 1 && r0[u].w < n_var;
      c += 1u + ((k.cls != 0 && !simple) ? 0x10000u : 0u);
    };
    static_assert(CL_ROW_U == 4, "one call per slice");
    one(0); one(1); one(2); one(3);
  }
  uint32_t tot;
  (void)block_excl_scan<256 / WAVE>(c, s_w, &tot);
  if (threadIdx.x == 0) { cnt[b] = (tot & 0xffffu) + (b + 1 == ch_off[row + 1] ? 1u : 0u); lcnt[b] = tot >> 16; }
}

// Cuts chunks [b_first, ...) of the rows into instances.  A one-record shareable instance IS its variant (uid = the variant's index);
// any other instance with a cluster goes on the list for k_cl_enter / k_cl_uid.  No atomic that anybody waits for:
// places in the instance arrays and on the list come from the counting pass; the variants the FIRST rows describe set a bit, and
// the later launch reads those bits through LDS (a chunk's records are consecutive variants-in-a-row, ascending: one window of
// the bitmap, C3: all 4 KB of it, loaded once per workgroup) and describes what they left.
#define CL_BM_WORDS 4096  // the window's LDS words (16 KB); a chunk that spans more variants asks the bitmap in HBM directly
__global__ __launch_bounds__(256) void k_cl_fill(const HxHead* __restrict__ recs, const uint64_t* __restrict__ hv_off,
                                                 const uint32_t* __restrict__ hap_len, const int32_t* __restrict__ ss_,
                                                 const int32_t* __restrict__ se_, const uint32_t* __restrict__ ch_off,
                                                 const uint32_t* __restrict__ ch_row, const uint32_t* __restrict__ inst_base,
                                                 const uint32_t* __restrict__ list_base, ClInst ci, unsigned long long* __restrict__ var_desc,
                                                 uint32_t* claim_bits, uint32_t n_var, ClListed* __restrict__ cx_list, uint32_t* __restrict__ status,
                                                 uint32_t b_first, uint32_t n_rows, uint32_t head) {
  __shared__ uint32_t s_c[256 / WAVE][CL_ROW_U];  // cluster starts per wave and record slice
  __shared__ uint32_t s_l[256 / WAVE][CL_ROW_U];  // ... and how many of them are listed
  __shared__ uint32_t s_bm[CL_BM_WORDS];
  const uint32_t b = b_first + blockIdx.x;
  if (b >= ch_off[n_rows]) return;  // (launched over the bound on the chunks; workgroup-uniform)
  const uint32_t row = ch_row[b];
  const uint64_t lo = hv_off[row], hi = hv_off[row + 1];
  const uint64_t b0 = lo + (uint64_t)(b - ch_off[row]) * CL_CHUNK;
  const int32_t ss = ss_[row], se = se_[row], hl = (int32_t)hap_len[row];
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t at = inst_base[b], lat = list_base[b];
  uint32_t round_tot = 0;
  if (hi > lo) {
    // the chunk's window of the variants' bitmap (the head launch writes the bitmap and reads nothing of it)
    const uint64_t b1 = b0 + CL_CHUNK < hi ? b0 + CL_CHUNK : hi;
    const uint32_t v_lo = recs[b0].var, v_hi = recs[b1 - 1].var;
    const uint32_t w_lo = v_lo >> 5, n_w = v_hi >= v_lo && v_hi < n_var ? (v_hi >> 5) - w_lo + 1u : 0u;
    const bool bm_lds = !head && n_w != 0u && n_w <= CL_BM_WORDS;  // workgroup-uniform
    if (bm_lds) for (uint32_t w = threadIdx.x; w < n_w; w += 256) s_bm[w] = claim_bits[w_lo + w];
    // CL_ROW_U slices of 256 consecutive records: every thread's record and the one before it, asked for at once
    uint4 r0[CL_ROW_U], rp[CL_ROW_U];
    bool st[CL_ROW_U];
    unsigned long long bal[CL_ROW_U];
#pragma unroll
    for (int u = 0; u < CL_ROW_U; ++u) {
      const uint64_t j = b0 + u * 256 + threadIdx.x;
      const uint64_t jc = j < hi ? j : hi - 1;
      r0[u] = *reinterpret_cast<const uint4*>(recs + jc);                        // {o, rs, alt_len, variant}
      rp[u] = *reinterpret_cast<const uint4*>(recs + (jc > lo ? jc - 1 : jc));
    }
#pragma unroll
    for (int u = 0; u < CL_ROW_U; ++u) {
      const uint64_t j = b0 + u * 256 + threadIdx.x;
      st[u] = j < hi && (j == lo || (int32_t)r0[u].x - ((int32_t)rp[u].x + (int32_t)rp[u].z) > CL_LINK);
      bal[u] = __ballot(st[u]);
      if (lane == 0) s_c[wv][u] = (uint32_t)__popcll(bal[u]);
    }
    __syncthreads();  // (also publishes the bitmap window)
    uint32_t pre[CL_ROW_U];  // starts of the slices / waves in front of this wave's part of slice u (slice-major = record order)
#pragma unroll
    for (int u = 0; u < CL_ROW_U; ++u) {
      pre[u] = round_tot;
#pragma unroll
      for (int w = 0; w < 256 / WAVE; ++w) {
        const uint32_t x = s_c[w][u];
        if (w < (int)wv) pre[u] += x;
        round_tot += x;
      }
    }
    uint4 ea[CL_ROW_U], eb[CL_ROW_U];  // the list entries of this thread's starts, if they are listed
    bool listed[CL_ROW_U];
    unsigned long long lbal[CL_ROW_U];
    auto cut = [&](const int u) __attribute__((always_inline)) {  // (called once per slice with a constant: the slices' registers stay registers)
      const uint4 nx1 = shfl_down4(r0[u], 1), nx2 = shfl_down4(r0[u], 2);  // (every lane takes part in the exchange)
      listed[u] = false;
      ea[u] = make_uint4(0u, 0u, 0u, 0u); eb[u] = ea[u];
      if (st[u]) {
        const uint64_t j = b0 + u * 256 + threadIdx.x;
        const uint32_t i = at + pre[u] + (uint32_t)__popcll(bal[u] & ((1ull << lane) - 1ull));
        const ClCut k = cl_cut(recs, lo, hi, j, r0[u], rp[u], nx1, nx2, lane, ss, se, hl, row);
        if (k.too_long) atomicOr(status, 1u);
        ci.o[i] = k.o_first; ci.row[i] = row; ci.pa[i] = k.pa | (k.run_inside ? (int32_t)0x80000000 : 0); ci.rb[i] = k.rb;
        const uint32_t v = r0[u].w;
        const bool simple = k.n == 1 && k.cls == 1 && v < n_var;  // the cluster is its variant
        if (!k.cls) ci.uid[i] = CL_NONE;
        if (simple) {
          ci.uid[i] = v;
          const uint32_t bit = 1u << (v & 31u);
          const unsigned long long desc = (unsigned long long)(uint32_t)j | ((unsigned long long)row << 32);  // (row >= 1: never 0)
          if (head) {  // the first rows describe what they carry and say so
            var_desc[v] = desc;
            (void)__hip_atomic_fetch_or(&claim_bits[v >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else if (!((bm_lds ? s_bm[(v >> 5) - w_lo] : claim_bits[v >> 5]) & bit)) {
            var_desc[v] = desc;  // whichever of these lands last is whole: one aligned 64-bit store
          }
        }
        listed[u] = k.cls != 0 && !simple;
        ea[u] = make_uint4(i, (uint32_t)k.key, (uint32_t)(k.key >> 32), (uint32_t)j);
        eb[u] = make_uint4(k.n | (k.cls << 16), (uint32_t)(k.o_first + k.rb), (uint32_t)k.o_first, v);
      }
      lbal[u] = __ballot(listed[u]);
      if (lane == 0) s_l[wv][u] = (uint32_t)__popcll(lbal[u]);
    };
    static_assert(CL_ROW_U == 4, "one call per slice");
    cut(0); cut(1); cut(2); cut(3);
    __syncthreads();
    uint32_t lpre = lat;  // places on the list: slice-major, like the instances
#pragma unroll
    for (int u = 0; u < CL_ROW_U; ++u) {
      uint32_t mine = lpre;
#pragma unroll
      for (int w = 0; w < 256 / WAVE; ++w) {
        const uint32_t x = s_l[w][u];
        if (w < (int)wv) mine += x;
        lpre += x;
      }
      if (listed[u]) {
        uint4* dst = reinterpret_cast<uint4*>(cx_list + mine + (uint32_t)__popcll(lbal[u] & ((1ull << lane) - 1ull)));
        dst[0] = ea[u]; dst[1] = eb[u];
      }
    }
  }
  if (threadIdx.x == 0 && b + 1 == ch_off[row + 1]) {  // the row's last chunk: the closing instance
    const uint32_t i = at + round_tot;
    int32_t pa = 0, rb = 0;
    if (hi > lo) { pa = recs[hi - 1].o + (int32_t)recs[hi - 1].alt_len; rb = (int32_t)recs[hi - 1].rs - pa; }
    ci.o[i] = CL_FAR; ci.row[i] = row; ci.pa[i] = pa; ci.rb[i] = rb; ci.uid[i] = CL_NONE;
  }
}

// The listed instances into the table, CL_UID_U per thread (their looks in flight together; an ordinary cached load first: a stale
// answer only sends the instance on to the atomic, which tells the truth).  The instance that takes a free slot opens a new distinct
// cluster: it draws the next number behind the variants (one atomic per WORKGROUP), describes the cluster and fills in the slot; one
// that finds its key there notes CL_PENDING | slot for k_cl_uid.
#define CL_UID_U 4
__global__ __launch_bounds__(256) void k_cl_enter(const ClListed* __restrict__ cx_list, const uint32_t* __restrict__ n_list_dev, uint32_t t_first,
                                                  uint32_t t_end, ClSlot* tab, uint32_t mask, uint32_t max_probe, uint32_t fail_bit,
                                                  uint32_t* __restrict__ inst_uid, const uint32_t* __restrict__ inst_row, uint32_t* __restrict__ cx_state,
                                                  ClUniq cu, uint32_t* n_table, uint32_t n_var, uint32_t u_cap, uint32_t* __restrict__ status) {
  __shared__ uint32_t s_n[256 / WAVE][CL_UID_U];
  __shared__ uint32_t s_base;
  const uint32_t n_list = *n_list_dev < t_end ? *n_list_dev : t_end;  // (launched over the bound on the list)
  const uint32_t t0 = t_first + blockIdx.x * (256 * CL_UID_U) + threadIdx.x;
  if (t_first + blockIdx.x * (256 * CL_UID_U) >= n_list) return;  // workgroup-uniform
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  uint4 ent[CL_UID_U], lk[CL_UID_U];
  uint32_t sl[CL_UID_U];
  bool on[CL_UID_U], won[CL_UID_U];
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    const uint32_t t = t0 + u * 256;
    on[u] = t < n_list;
    ent[u] = *reinterpret_cast<const uint4*>(cx_list + (on[u] ? t : t_first));  // {instance, key, first record}
  }
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    const unsigned long long k = (unsigned long long)ent[u].y | ((unsigned long long)ent[u].z << 32);
    sl[u] = (uint32_t)(k >> 17) & mask;
    lk[u] = *reinterpret_cast<const uint4*>(tab + sl[u]);
  }
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    won[u] = false;
    if (!on[u]) continue;
    const unsigned long long k = (unsigned long long)ent[u].y | ((unsigned long long)ent[u].z << 32);
    uint4 e0 = lk[u];
    uint32_t s = sl[u];
    bool placed = false;
    // (the full-size table has >= 2 slots per instance: a free slot is met long before the bound; a first attempt with a table sized
    // for the distinct clusters expected gives up after max_probe slots and the host repeats the passes with the full size)
    for (uint32_t probe = 0; probe <= mask && probe < max_probe; ++probe) {
      if (probe) e0 = *reinterpret_cast<const uint4*>(tab + s);
      unsigned long long c = (unsigned long long)e0.x | ((unsigned long long)e0.y << 32);
      if (c == 0ull) {  // (a slot never changes hands once taken: any other value seen is final)
        c = atomicCAS(&tab[s].key, 0ull, k);
        won[u] = c == 0ull;
      }
      if (c == 0ull || c == k) { placed = true; break; }
      s = (s + 1u) & mask;
    }
    sl[u] = s;
    const uint32_t t = t0 + u * 256;
    if (!placed) { won[u] = false; cx_state[t] = CL_NONE; inst_uid[ent[u].x] = CL_NONE; atomicOr(status, fail_bit); continue; }  // full-size table: cannot happen
    cx_state[t] = won[u] ? CL_NONE : (CL_PENDING | s);
  }
  // the new distinct clusters of this workgroup draw their numbers with one atomic
  unsigned long long wb[CL_UID_U];
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) { wb[u] = __ballot(won[u]); if (lane == 0) s_n[wv][u] = (uint32_t)__popcll(wb[u]); }
  __syncthreads();
  uint32_t tot = 0, mine[CL_UID_U];
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    mine[u] = tot;
#pragma unroll
    for (int w = 0; w < 256 / WAVE; ++w) { const uint32_t x = s_n[w][u]; if (w < (int)wv) mine[u] += x; tot += x; }
  }
  if (!tot) return;  // workgroup-uniform
  if (threadIdx.x == 0) s_base = atomicAdd(n_table, tot);
  __syncthreads();
  const uint32_t base = n_var + s_base;
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    if (!won[u]) continue;
    const uint32_t un = base + mine[u] + (uint32_t)__popcll(wb[u] & ((1ull << lane) - 1ull));
    const uint32_t i = ent[u].x;
    if (un >= u_cap) { inst_uid[i] = CL_NONE; atomicOr(status, fail_bit); continue; }  // (more distinct clusters than slots: cannot happen)
    const uint4 e1 = reinterpret_cast<const uint4*>(cx_list + (t0 + u * 256))[1];  // {records | class, REF position, row position, first variant}
    *reinterpret_cast<uint4*>(&tab[sl[u]].up1) = make_uint4(un + 1u, e1.x, e1.y, e1.w);
    inst_uid[i] = un;
    cu.rec[un] = ent[u].w; cu.n[un] = e1.x & 0xffffu; cu.row[un] = inst_row[i]; cu.o[un] = (int32_t)e1.z;
  }
}

// what a search needs of every distinct cluster: for a variant's cluster the description unpacked from the one word its describer
// left (a variant that is nowhere a cluster of its own stays a hole: u_n = 0); for all of them the template rows they can have and
// the position-map segment their searches start from
__global__ __launch_bounds__(256) void k_cl_describe(const uint32_t* __restrict__ n_table_dev, uint32_t n_var, uint32_t u_cap,
                                                     const unsigned long long* __restrict__ var_desc, ClUniq cu, const HxHead* __restrict__ recs,
                                                     const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_rel,
                                                     uint32_t* __restrict__ n_variant_clusters, unsigned long long* __restrict__ slots_total) {
  const uint32_t u = blockIdx.x * 256 + threadIdx.x;
  uint32_t span_u = 0;
  const uint32_t nu = n_var + *n_table_dev < u_cap ? n_var + *n_table_dev : u_cap;  // (launched over the bound on the distinct clusters)
  bool variant_cluster = false;
  if (u < nu) {
    uint32_t n, r, row;
    int32_t o_first;
    if (u < n_var) {
      const unsigned long long d = var_desc[u];
      variant_cluster = d != 0ull;
      r = (uint32_t)d; row = (uint32_t)(d >> 32); n = variant_cluster ? 1u : 0u;
      o_first = variant_cluster ? recs[r].o : 0;
      cu.rec[u] = r; cu.n[u] = n; cu.row[u] = row; cu.o[u] = o_first;
    } else {
      n = cu.n[u]; r = cu.rec[u]; row = cu.row[u]; o_first = cu.o[u];
    }
    if (n == 0) { cu.span2[u] = 0; cu.seg[u] = 0; }
    else {
      const int32_t o_end = recs[r + n - 1].o + (int32_t)recs[r + n - 1].alt_len;
      span_u = 2u * ((uint32_t)(o_end - o_first) + CL_LINK);  // rows the cluster can have: window starts [o_first - L + 1, o_end) x 2 strands
      cu.span2[u] = span_u;
      // the position-map segment in force CL_LINK + PAD positions in front of the cluster: every position a search asks for lies behind it
      const uint32_t rel = o_first > CL_LINK + HAWK_PAD ? (uint32_t)(o_first - CL_LINK - HAWK_PAD) : 0u;
      uint32_t lo = seg_off[row], hi = seg_off[row + 1];
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_rel[mid] <= rel) lo = mid; else hi = mid;
      }
      cu.seg[u] = lo;
    }
  }
  // how many variants are clusters of their own (statistics), and the template rows a search may need at most - the sum of the
  // clusters' bounds: one atomic each per WORKGROUP (same-address atomics are served one after the other, ~10 ns each)
  __shared__ uint32_t s_red[256 / WAVE][2];
  const unsigned long long vb = __ballot(variant_cluster);
  const uint32_t ssum = wave_sum(span_u);
  if ((threadIdx.x & (WAVE - 1)) == 0) { s_red[threadIdx.x / WAVE][0] = (uint32_t)__popcll(vb); s_red[threadIdx.x / WAVE][1] = ssum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int w = 0; w < 256 / WAVE; ++w) { a += s_red[w][0]; b += s_red[w][1]; }
    if (a) atomicAdd(n_variant_clusters, a);
    if (b) atomicAdd(slots_total, (unsigned long long)b);
  }
}
// what the host wants to know of a finished build, as one block (one copy instead of four)
__global__ void k_cl_results(const uint32_t* __restrict__ n_inst, const uint32_t* __restrict__ counters, const unsigned long long* __restrict__ slots_total,
                             const uint32_t* __restrict__ status, unsigned long long* __restrict__ out) {
  out[0] = *n_inst; out[1] = counters[0]; out[2] = counters[1]; out[3] = *slots_total; out[4] = *status;
}
// The listed instances that found their hash in the table: number from the slot, identity compared record by record with the instance
// that opened the cluster (exactness: same hash is not same cluster until then).  Loads level by level.
__global__ __launch_bounds__(256) void k_cl_uid(const ClListed* __restrict__ cx_list, const uint32_t* __restrict__ n_list_dev, uint32_t t_end,
                                                const uint32_t* __restrict__ cx_state, const ClSlot* __restrict__ tab, uint32_t* __restrict__ inst_uid,
                                                ClUniq cu, const HxHead* __restrict__ recs, uint32_t* __restrict__ status) {
  const uint32_t n_list = *n_list_dev < t_end ? *n_list_dev : t_end;  // (launched over the bound on the list)
  if (blockIdx.x * (256 * CL_UID_U) >= n_list) return;
  const uint32_t t0 = blockIdx.x * (256 * CL_UID_U) + threadIdx.x;
  uint32_t v[CL_UID_U];
  bool pend[CL_UID_U];
  bool any = false;
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    const uint32_t t = t0 + u * 256;
    v[u] = t < n_list ? cx_state[t] : CL_NONE;
    pend[u] = v[u] != CL_NONE && (v[u] & CL_PENDING);
    any = any || pend[u];
  }
  if (!__any(any)) return;
  uint4 e1[CL_UID_U], la[CL_UID_U], lb[CL_UID_U];
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    e1[u] = reinterpret_cast<const uint4*>(tab + (pend[u] ? (v[u] & ~CL_PENDING) : 0u))[1];  // {number + 1, records | class, REF position, first variant}
    const uint4* l = reinterpret_cast<const uint4*>(cx_list + (pend[u] ? t0 + u * 256 : 0u));
    la[u] = l[0]; lb[u] = l[1];  // {instance, key, first record} {records | class, REF position, row position, first variant}
  }
  uint32_t rrec[CL_UID_U];
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) rrec[u] = (pend[u] && e1[u].x != 0u && (lb[u].x & 0xffffu) > 1u) ? cu.rec[e1[u].x - 1u] : 0u;
  bool bad = false;
#pragma unroll
  for (int u = 0; u < CL_UID_U; ++u) {
    if (!pend[u]) continue;
    const uint32_t i = la[u].x, ncls = lb[u].x;
    if (e1[u].x == 0u) { inst_uid[i] = CL_NONE; bad = true; continue; }  // (a taken slot without a number: cannot happen)
    inst_uid[i] = e1[u].x - 1u;
    // both shareable, as many records, the same REF position of the first allele and the same first variant (its allele, place and
    // REF resume follow), then record by record
    bool b = ncls != e1[u].y || (ncls >> 16) != 1 || lb[u].y != e1[u].z || lb[u].w != e1[u].w;
    const uint32_t n = ncls & 0xffffu;
    if (!b && n > 1) {
      const HxHead* pa = recs + la[u].w;
      const HxHead* pb = recs + rrec[u];
      const int32_t oa = (int32_t)lb[u].z, ob = pb[0].o;
      for (uint32_t k = 1; k < n; ++k) {
        const uint4 xa = *reinterpret_cast<const uint4*>(pa + k), xb = *reinterpret_cast<const uint4*>(pb + k);
        b = b || xa.w != xb.w || xa.z != xb.z || xa.y != xb.y || (int32_t)xa.x - oa != (int32_t)xb.x - ob;
      }
    }
    bad = bad || b;
  }
  if (bad) atomicOr(status, 2u);
}

// rows -> chunks: ch_off[n_rows + 1] (chunks in front of every row) and the chunks' rows; ch_row has room for n_records / CL_CHUNK + n_rows entries
uint32_t hawk_cl_chunk_bound(uint64_t n_records, uint32_t n_rows) { return (uint32_t)(n_records / CL_CHUNK) + n_rows; }
void hawk_launch_cl_chunks(hipStream_t st, const uint64_t* hv_off, const uint8_t* is_ref, const int32_t* ss, const int32_t* se, uint32_t n_rows,
                           uint32_t* ch_off, uint32_t* ch_row) {
  hipLaunchKernelGGL(k_cl_chunks, dim3(1), dim3(1024), 0, st, hv_off, is_ref, ss, se, n_rows, ch_off, ch_row);
}
void hawk_launch_cl_count(hipStream_t st, const void* recs, const uint64_t* hv_off, const uint32_t* hap_len, const int32_t* ss, const int32_t* se,
                          const uint32_t* ch_off, const uint32_t* ch_row, uint32_t n_rows, uint32_t n_var, uint32_t ch_bound, uint32_t* cnt /* zeroed */,
                          uint32_t* lcnt /* zeroed */) {
  if (ch_bound)
    hipLaunchKernelGGL(k_cl_count, dim3(ch_bound), dim3(256), 0, st, static_cast<const HxHead*>(recs), hv_off, hap_len, ss, se, ch_off, ch_row, n_rows,
                       n_var, cnt, lcnt);
}
// Two launches: the first chunks (a few dozen rows), then the rest - and likewise the head of the list, then the rest.  A common
// cluster has thousands of instances, half of which are in flight at once in a single launch: in the table passes they all find
// their slot empty and all go to the atomic - thousands of device-scope atomics queued at one address (60-90 us, whatever the
// panel's size); in the cutting pass they would all describe the cluster.  After the head launch every common cluster is there.
// The launches run over BOUNDS (the numbers of chunks and of listed instances are on the device only; surplus workgroups leave at once).
#define CL_HEAD_CHUNKS 96
#define CL_HEAD_LISTED 16384
void hawk_launch_cl_fill(hipStream_t st, const void* recs, const uint64_t* hv_off, const uint32_t* hap_len, const int32_t* ss, const int32_t* se,
                         uint32_t n_rows, const uint32_t* ch_off, const uint32_t* ch_row, uint32_t ch_bound, const uint32_t* inst_base,
                         const uint32_t* list_base, int32_t* o, uint32_t* row, int32_t* pa, int32_t* rb, uint32_t* inst_uid, void* var_desc,
                         uint32_t* claim_bits, uint32_t n_var, void* cx_list, uint32_t* status) {
  ClInst ci{o, row, pa, rb, inst_uid};
  // (a small job - a stretch of a region on one of several GPUs - has no crowd to keep away from one address, and two launches
  // of pure latency to save: everything in the second launch, where every instance whose variant nobody described writes it)
  const uint32_t n_head = ch_bound < 16 * CL_HEAD_CHUNKS ? 0u : CL_HEAD_CHUNKS;
  if (n_head)
    hipLaunchKernelGGL(k_cl_fill, dim3(n_head), dim3(256), 0, st, static_cast<const HxHead*>(recs), hv_off, hap_len, ss, se, ch_off, ch_row, inst_base,
                       list_base, ci, static_cast<unsigned long long*>(var_desc), claim_bits, n_var, static_cast<ClListed*>(cx_list), status, 0u, n_rows, 1u);
  if (ch_bound > n_head)
    hipLaunchKernelGGL(k_cl_fill, dim3(ch_bound - n_head), dim3(256), 0, st, static_cast<const HxHead*>(recs), hv_off, hap_len, ss, se, ch_off, ch_row,
                       inst_base, list_base, ci, static_cast<unsigned long long*>(var_desc), claim_bits, n_var, static_cast<ClListed*>(cx_list), status,
                       n_head, n_rows, 0u);
}
// after the cutting pass: the listed instances through the table, the distinct clusters' descriptions, then the listed instances
// that share a cluster.  counters: [0] the table's distinct clusters, [1] the variants that are clusters of their own, [4..5] the
// template rows a search may need at most (zeroed); results: {instances, [0], [1], template rows, status} for the host, 64 bytes.
void hawk_launch_cl_finish(hipStream_t st, uint32_t list_bound, const uint32_t* n_inst_dev, const uint32_t* n_list_dev, uint32_t* counters,
                           unsigned long long* results, uint32_t n_var, uint32_t u_cap, void* tab,
                           uint32_t mask, uint32_t max_probe, uint32_t fail_bit, const void* cx_list, uint32_t* cx_state, const void* var_desc,
                           const void* recs, uint32_t* inst_uid, const uint32_t* inst_row, const uint32_t* seg_off, const uint32_t* seg_rel,
                           uint32_t* u_rec, uint32_t* u_n, uint32_t* u_row, int32_t* u_o, uint32_t* u_seg, uint32_t* u_span2, uint32_t* status) {
  ClUniq cu{u_rec, u_n, u_row, u_o, u_seg, u_span2};
  const uint32_t per = 256 * CL_UID_U;
  const uint32_t head = list_bound < 16 * CL_HEAD_LISTED ? 0u : CL_HEAD_LISTED;
  const ClListed* list = static_cast<const ClListed*>(cx_list);
  if (head)
    hipLaunchKernelGGL(k_cl_enter, dim3((head + per - 1) / per), dim3(256), 0, st, list, n_list_dev, 0u, head, static_cast<ClSlot*>(tab), mask, max_probe,
                       fail_bit, inst_uid, inst_row, cx_state, cu, counters, n_var, u_cap, status);
  if (list_bound > head)
    hipLaunchKernelGGL(k_cl_enter, dim3((list_bound - head + per - 1) / per), dim3(256), 0, st, list, n_list_dev, head, list_bound, static_cast<ClSlot*>(tab),
                       mask, max_probe, fail_bit, inst_uid, inst_row, cx_state, cu, counters, n_var, u_cap, status);
  hipLaunchKernelGGL(k_cl_describe, dim3((u_cap + 255) / 256), dim3(256), 0, st, counters, n_var, u_cap, static_cast<const unsigned long long*>(var_desc), cu,
                     static_cast<const HxHead*>(recs), seg_off, seg_rel, counters + 1, reinterpret_cast<unsigned long long*>(counters + 4));
  if (list_bound)
    hipLaunchKernelGGL(k_cl_uid, dim3((list_bound + per - 1) / per), dim3(256), 0, st, list, n_list_dev, list_bound, cx_state, static_cast<const ClSlot*>(tab),
                       inst_uid, cu, static_cast<const HxHead*>(recs), status);
  hipLaunchKernelGGL(k_cl_results, dim3(1), dim3(1), 0, st, n_inst_dev, counters, reinterpret_cast<const unsigned long long*>(counters + 4), status, results);
}

// ---- per search ------------------------------------------------------------------------------------
__device__ __forceinline__ VcRanges cs_ranges(const HapSetDev& hs, const ScanParams& p, uint32_t h) {
  VcRanges rg;
  const int ss = hs.scan_start[h], se = hs.scan_stop[h], haplen = (int)hs.hap_len[h];
  const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
  const int qmin = HAWK_PAD, qmax = haplen - p.L - HAWK_PAD + 1;
  rg.slo[0] = ss - poF; rg.shi[0] = se - poF; rg.slo[1] = ss - poR; rg.shi[1] = se - poR;
  rg.lo[0] = rg.slo[0] > qmin ? rg.slo[0] : qmin; rg.hi[0] = rg.shi[0] < qmax ? rg.shi[0] : qmax;
  rg.lo[1] = rg.slo[1] > qmin ? rg.slo[1] : qmin; rg.hi[1] = rg.shi[1] < qmax ? rg.shi[1] : qmax;
  return rg;
}
// inclusive sum over the CS_G lanes of a group
__device__ __forceinline__ uint32_t group_incl_scan(uint32_t v, uint32_t gl) {
#pragma unroll
  for (int d = 1; d < CS_G; d <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)v, d, CS_G);
    if (gl >= (uint32_t)d) v += t;
  }
  return v;
}

// position map (haplotype.py:90-159) from a hint: k0 = a segment of row h starting at or before every position the cluster asks for
__device__ __forceinline__ int64_t posmap_hint(const HapSetDev& hs, uint32_t k0, uint32_t kend, uint32_t rel) {
  uint32_t k = k0;
  while (k + 1 < kend && hs.seg_rel[k + 1] <= rel) ++k;
  return hs.seg_gen[k] + (int64_t)(rel - hs.seg_rel[k]);
}

// Every distinct cluster's rows, once: a group of CS_G lanes per cluster, each lane one word of 32 window starts per round.
// res[2u] = {rows of strand 0, rows of strand 1, PAM hits, candidates} of the cluster's own window starts, res[2u + 1] = {first template
// row, REF's hits before the cluster, REF's hits behind it, -} (32 bytes: one request to L2 from k_cs_count); its template rows sit
// at tbase[u] (strand 0 in position order, then strand 1), packed: a workgroup adds up its clusters' rows and takes its
// stretch of the template array with one atomic.
__global__ __launch_bounds__(256) void k_cs_templates(HapSetDev hs, VcArgs va, ClDict cd, ScanParams p, GuideParams gp, RefInfo ri,
                                                      uint4* __restrict__ res, uint32_t* __restrict__ tbase, CsRow* __restrict__ trows, unsigned long long* __restrict__ t_count, uint64_t t_cap, int* status) {
  __shared__ double s_cfd[336];
  __shared__ uint32_t s_w[256 / WAVE];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_X[15][256];  // a short cluster's strings, one column per lane: [plane * 3 + word][lane]
  __shared__ uint32_t s_k[2][256];   // ... the window starts its filters keep, per strand
  __shared__ uint32_t s_e[256];      // ... and how many the group's lanes before it keep (strand 0 | strand 1 << 16)
  const uint32_t tid = threadIdx.x;
  if (gp.score_cfdon) for (uint32_t i = tid; i < 336; i += 256) s_cfd[i] = gp.cfd_mm[i];
  __syncthreads();
  const uint32_t gl = tid & (CS_G - 1);
  const uint32_t u = (blockIdx.x * 256 + tid) / CS_G;
  // the numbers of the distinct clusters have holes (a variant that is nowhere a cluster of its own: u_n = 0); their groups and the
  // surplus groups of the last workgroup go through the motions on record 0 of row 0 with no window start, and write nothing
  const uint32_t nc_u = u < cd.n_uniq ? cd.u_n[u] : 0u;
  const bool live = nc_u != 0u;
  const bool hole = u < cd.n_uniq && !live;
  const HxVar* __restrict__ recs = static_cast<const HxVar*>(va.recs_);
  const uint32_t h = live ? cd.u_row[u] : 0u, r0 = live ? cd.u_rec[u] : 0u, nc = live ? nc_u : 1u;
  const uint32_t seg0 = live ? cd.u_seg[u] : hs.seg_off[0], seg_end = hs.seg_off[h + 1];
  const uint32_t back = (uint64_t)r0 > va.hv_off[h] ? 1u : 0u;  // the record in front of the cluster sets the REF shift it starts from
  const HxVar* __restrict__ sv = recs + r0 - back;
  const int nrec = (int)(nc + back);
  const int L = p.L;
  const int32_t haplen = (int32_t)hs.hap_len[h];
  const int32_t o_first = live ? cd.u_o[u] : 0;
  const int32_t o_end = live ? recs[r0 + nc - 1].o + (int32_t)recs[r0 + nc - 1].alt_len : 0;
  const int32_t qa = o_first - (L - 1) > 0 ? o_first - (L - 1) : 0;
  const int32_t qb = o_end < haplen ? o_end : haplen;
  const int nwords = qb > qa ? (qb - qa + 31) / 32 : 0;
  const VcRanges rg = cs_ranges(hs, p, h);
  // REF's PAM hits (both strands) in front of where the cluster's own window starts begin, and in front of where REF resumes behind
  // its last allele: the clean run between two clusters of a row then has  before(this cluster) - behind(the one in front)  hits,
  // whichever row carries the two (k_cs_count) - REF coordinates, so part of the cluster's identity
  auto hits_before = [&](int64_t x) -> uint32_t {
    const int64_t xm = (int64_t)va.ref_S * 32;
    const uint32_t xc = (uint32_t)(x < 0 ? 0 : (x > xm ? xm : x));
    const uint4 e = va.hp[xc >> 5];
    const uint32_t m = (1u << (xc & 31u)) - 1u;
    return e.y + (uint32_t)__popc(e.x & m) + e.w + (uint32_t)__popc(e.z & m);
  };
  const int32_t rb_front = back ? (int32_t)sv[0].rs - (sv[0].o + (int32_t)sv[0].alt_len) : 0;
  const uint32_t h_front = hits_before((int64_t)o_first + rb_front - (L - 1)), h_behind = hits_before((int64_t)recs[r0 + nc - 1].rs);
  const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
  const uint32_t mlo = L >= 32 ? 0xffffffffu : ((1u << L) - 1u), mhi = L <= 32 ? 0u : ((1u << (L - 32)) - 1u);
  const int W = L + 2 * HAWK_PAD;
  const uint32_t whi = W >= 64 ? 0xffffffffu : ((1u << (W - 32)) - 1u);
  const int ncfd = gp.guidelen < 20 ? gp.guidelen : 20;
  const uint32_t cfdmask = (1u << ncfd) - 1u;
  auto ref32 = [&](int pl, uint32_t r) -> uint32_t {
    const uint32_t w = (r >> 5) < va.ref_S - 2 ? (r >> 5) : va.ref_S - 2;
    return ext32_glb(va.ref[pl], (w << 5) | (r & 31u));
  };
  // one word: the five 96-bit strings around window starts [q0, q0 + 32), PAM hits, candidates, the starts kept by the filters
  auto word = [&](int32_t q0, uint32_t (&X)[5][3], uint32_t& kF, uint32_t& kR, uint32_t& hits, uint32_t& cand) {
    const int32_t p0 = q0 - HAWK_PAD;
    const int32_t p0c = p0 < 0 ? 0 : p0;
    int32_t shift_run = 0;
    if (!vc_string_fast(va, sv, nrec, p0c, haplen, q0 + 32, X, shift_run))
      hx_words_t<false, 3>(va.alt_codes, sv, sv, nrec, nrec, false, p0c, haplen, ref32, X[0], X[1], X[2], X[3], X[4]);
    if (p0 < 0) {  // the string starts at the row's first base: bit i is position q0 - PAD + i all the same
      const uint32_t sh = (uint32_t)(-p0);
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) {
        X[pl][2] = fsh(X[pl][1], X[pl][2], 32u - sh);
        X[pl][1] = fsh(X[pl][0], X[pl][1], 32u - sh);
        X[pl][0] = X[pl][0] << sh;
      }
    }
    uint32_t v0 = X[4][0], v1 = X[4][1], v2 = X[4][2];  // E: window starts whose spacer + PAM holds a variant base
    int r = 1;
    while (2 * r <= L && r < 32) {
      v0 |= fsh(v0, v1, (uint32_t)r); v1 |= fsh(v1, v2, (uint32_t)r); v2 |= v2 >> r;
      r *= 2;
    }
    const int rem = L - r;
    if (rem > 0) {
      if (rem < 32) { v0 |= fsh(v0, v1, (uint32_t)rem); v1 |= fsh(v1, v2, (uint32_t)rem); v2 |= v2 >> rem; }
      else { v0 |= v1; v1 |= v2; }
    }
    const uint32_t E = fsh(v0, v1, HAWK_PAD);
    const uint32_t own = range_mask(q0, qa, qb);
    uint32_t f = pam_match96(X, p.pam_fwd, p.pamlen, HAWK_PAD + poF) & own;
    uint32_t rv = pam_match96(X, p.pam_rev, p.pamlen, HAWK_PAD + poR) & own;
    f &= range_mask(q0, rg.slo[0], rg.shi[0]);
    rv &= range_mask(q0, rg.slo[1], rg.shi[1]);
    hits += __popc(f) + __popc(rv);
    f &= range_mask(q0, rg.lo[0], rg.hi[0]);
    rv &= range_mask(q0, rg.lo[1], rg.hi[1]);
    cand += __popc(f) + __popc(rv);
    kF = f & E;
    kR = rv & E;
  };
  // which kept starts are rows: not the REF guide at the same (start, strand) again (remove_redundant_guides)
  auto classify = [&](int32_t q0, const uint32_t (&X)[5][3], uint32_t kF, uint32_t kR, uint32_t& vF, uint32_t& vR) {
    vF = 0; vR = 0;
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      uint32_t m = s ? kR : kF;
      while (m) {
        const uint32_t bpos = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        const int64_t start = posmap_hint(hs, seg0, seg_end, (uint32_t)q0 + bpos);
        const int64_t qr64 = start - ri.startp;
        const bool inr = qr64 >= 0 && qr64 < (int64_t)ri.n_bits;
        const uint32_t qr = inr ? (uint32_t)qr64 : 0u;
        const bool has_ref = inr && (((s ? ri.bits[1] : ri.bits[0])[qr >> 5] >> (qr & 31u)) & 1u);
        bool same = has_ref;
        if (has_ref) {
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) {
            W2 wn = ext96(X[pl][0], X[pl][1], X[pl][2], bpos);
            wn.hi &= whi;
            const uint32_t clo = fsh(wn.lo, wn.hi, HAWK_PAD) & mlo, chi = (wn.hi >> HAWK_PAD) & mhi;
            const W2 rc = ext_glb(va.ref[pl], qr);
            same = same && (rc.lo & mlo) == clo && (rc.hi & mhi) == chi;
          }
        }
        if (!same) { if (s) vR |= 1u << bpos; else vF |= 1u << bpos; }
      }
    }
  };
  auto write = [&](int32_t q0, const uint32_t (&X)[5][3], uint32_t vF, uint32_t vR, uint64_t at0, uint64_t at1) {
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      uint32_t m = s ? vR : vF;
      while (m) {
        const uint32_t bpos = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        const uint32_t q = (uint32_t)q0 + bpos;
        const uint64_t k = s ? at1++ : at0++;
        if (k >= t_cap) { atomicExch(status, -3 /* HAWK_E_CAPACITY: more rows than window starts */); continue; }
        W2 win[5], core[4], rcore[4];
#pragma unroll
        for (int pl = 0; pl < 5; ++pl) {
          win[pl] = ext96(X[pl][0], X[pl][1], X[pl][2], bpos);
          win[pl].hi &= whi;
          if (pl < 4) {
            core[pl].lo = fsh(win[pl].lo, win[pl].hi, HAWK_PAD) & mlo;
            core[pl].hi = (win[pl].hi >> HAWK_PAD) & mhi;
          }
        }
        const int64_t start = posmap_hint(hs, seg0, seg_end, q), stop = posmap_hint(hs, seg0, seg_end, q + (uint32_t)L);
        const int64_t qr64 = start - ri.startp;
        const bool inr = qr64 >= 0 && qr64 < (int64_t)ri.n_bits;
        const uint32_t qr = inr ? (uint32_t)qr64 : 0u;
        const bool has_ref = inr && (((s ? ri.bits[1] : ri.bits[0])[qr >> 5] >> (qr & 31u)) & 1u);
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
          rcore[pl] = ext_glb(va.ref[pl], qr);
          rcore[pl].lo &= mlo; rcore[pl].hi &= mhi;
          if (!has_ref) rcore[pl] = core[pl];
        }
        double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
        if (gp.score_cfdon && has_ref) {
          bool err;
          score = cfdon_from_slices(core, rcore, (uint32_t)s, L, cfdmask, s_cfd, err);
          if (err && gp.score_cfdon == 1) atomicExch(status, -5 /* HAWK_E_CFD */);
        }
        const bool pamfirst = (p.right != 0) != (s != 0);
        const int64_t ds = start - ri.startp, de = stop - start;
        if (ds < INT32_MIN || ds > INT32_MAX || de < INT32_MIN || de > INT32_MAX) atomicExch(status, -7 /* HAWK_E_UNSUPPORTED */);
        const uint64_t sc = (uint64_t)__double_as_longlong(score);
        uint4* __restrict__ tp = reinterpret_cast<uint4*>(trows + k);
        tp[0] = make_uint4((uint32_t)((int32_t)(pamfirst ? q : q + (uint32_t)p.guidelen) - o_first), (uint32_t)s | (has_ref ? 2u : 0u),
                           (uint32_t)(int32_t)ds, (uint32_t)(int32_t)de);
        tp[1] = make_uint4((uint32_t)sc, (uint32_t)(sc >> 32), win[0].lo, win[0].hi);
        tp[2] = make_uint4(win[1].lo, win[1].hi, win[2].lo, win[2].hi);
        tp[3] = make_uint4(win[3].lo, win[3].hi, win[4].lo, win[4].hi);
      }
    }
  };

  // ---- pass 1.  A cluster of up to CS_G words (nearly all of them) leaves its strings and kept starts in LDS: its survivors
  // are then dealt to the group's lanes one each per round, so that the position-map and REF look-ups of a cluster run side by
  // side instead of one after the other in the lane that owns the word (the kernel is a chain of dependent loads: with one
  // lane walking its word's survivors a wave lived 40 us).  Longer clusters count, then write, word by word.
  const bool one_round = nwords <= CS_G;  // uniform within a group
  const uint32_t g0 = tid & ~(uint32_t)(CS_G - 1);
  uint32_t X[5][3];
#pragma unroll
  for (int pl = 0; pl < 5; ++pl) X[pl][0] = X[pl][1] = X[pl][2] = 0;
  uint32_t vF = 0, vR = 0, n0 = 0, n1 = 0, hits = 0, cand = 0, TF = 0, TR = 0;
  if (one_round) {
    uint32_t kF = 0, kR = 0;
    if ((int)gl < nwords) word(qa + 32 * (int32_t)gl, X, kF, kR, hits, cand);
    const uint32_t c = (uint32_t)__popc(kF) | ((uint32_t)__popc(kR) << 16);
    const uint32_t inc = group_incl_scan(c, gl);
    const uint32_t tot = (uint32_t)__shfl((int)inc, CS_G - 1, CS_G);
    TF = tot & 0xffffu; TR = tot >> 16;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) { s_X[pl * 3 + 0][tid] = X[pl][0]; s_X[pl * 3 + 1][tid] = X[pl][1]; s_X[pl * 3 + 2][tid] = X[pl][2]; }
    s_k[0][tid] = kF; s_k[1][tid] = kR; s_e[tid] = inc - c;
  } else {
#pragma unroll 1
    for (int w0 = 0; w0 < nwords; w0 += CS_G) {
      const int w = w0 + (int)gl;
      uint32_t kF = 0, kR = 0;
      vF = 0; vR = 0;
      if (w < nwords) {
        word(qa + 32 * w, X, kF, kR, hits, cand);
        classify(qa + 32 * w, X, kF, kR, vF, vR);
      }
      const uint32_t c = (uint32_t)__popc(vF) | ((uint32_t)__popc(vR) << 16);
      const uint32_t tot = (uint32_t)__shfl((int)group_incl_scan(c, gl), CS_G - 1, CS_G);
      n0 += tot & 0xffffu; n1 += tot >> 16;
    }
  }
  hits = group_incl_scan(hits, gl); cand = group_incl_scan(cand, gl);  // the group's totals in its last lane
  // ---- the workgroup's stretch of the template array: rows of the long clusters, kept starts (>= rows) of the others
  const bool leader = live && gl == CS_G - 1;
  if (hole && gl == CS_G - 1) { res[2 * u] = make_uint4(0u, 0u, 0u, 0u); res[2 * u + 1] = make_uint4(0u, 0u, 0u, 0u); tbase[u] = 0u; }
  const uint32_t want = one_round ? TF + TR : n0 + n1;
  uint32_t btot;
  const uint32_t bex = block_excl_scan<256 / WAVE>(leader ? want : 0u, s_w, &btot);  // its barriers publish the groups' LDS entries
  if (tid == 0) s_base = btot ? atomicAdd(t_count, (unsigned long long)btot) : 0ull;
  __syncthreads();
  const uint32_t gex = (uint32_t)__shfl((int)bex, CS_G - 1, CS_G);  // the leader's exclusive offset = rows of the groups before
  const uint64_t tb = s_base + gex;
  if (tb + want > 0xffffffffull) atomicExch(status, -7 /* HAWK_E_UNSUPPORTED: template rows beyond 32-bit indices */);
  if (!one_round) {
    if (leader) { res[2 * u] = make_uint4(n0, n1, hits, cand); res[2 * u + 1] = make_uint4((uint32_t)tb, h_front, h_behind, 0u); tbase[u] = (uint32_t)tb; }
    if (!live || n0 + n1 == 0) return;  // no barrier below
    uint32_t b0 = 0, b1 = 0, h2 = 0, c2 = 0;
#pragma unroll 1
    for (int w0 = 0; w0 < nwords; w0 += CS_G) {
      const int w = w0 + (int)gl;
      uint32_t kF = 0, kR = 0;
      vF = 0; vR = 0;
      if (w < nwords) {
        word(qa + 32 * w, X, kF, kR, h2, c2);
        classify(qa + 32 * w, X, kF, kR, vF, vR);
      }
      const uint32_t c = (uint32_t)__popc(vF) | ((uint32_t)__popc(vR) << 16);
      const uint32_t inc = group_incl_scan(c, gl);
      const uint32_t tot = (uint32_t)__shfl((int)inc, CS_G - 1, CS_G);
      const uint32_t ex = inc - c;
      if (w < nwords) write(qa + 32 * w, X, vF, vR, tb + b0 + (ex & 0xffffu), tb + n0 + b1 + (ex >> 16));
      b0 += tot & 0xffffu; b1 += tot >> 16;
    }
    return;
  }
  // ---- pass 2 of a short cluster: survivor j of a strand -> lane j mod CS_G
  uint32_t done = 0;  // rows written so far (strand 0 first)
#pragma unroll 1
  for (int sd = 0; sd < 2; ++sd) {
    const uint32_t T = sd ? TR : TF, sh16 = sd ? 16u : 0u;
    uint32_t ns = 0;
#pragma unroll 1
    for (uint32_t j0 = 0; j0 < T; j0 += CS_G) {  // uniform within a group
      const uint32_t j = j0 + gl;
      const bool act = j < T;
      const uint32_t jj = act ? j : 0u;
      uint32_t l = 0;
#pragma unroll
      for (uint32_t step = CS_G / 2; step; step >>= 1) l += (((s_e[g0 + l + step] >> sh16) & 0xffffu) <= jj) ? step : 0u;
      const uint32_t src = g0 + l;
      const uint32_t bpos = select_bit(s_k[sd][src], jj - ((s_e[src] >> sh16) & 0xffffu));
      const uint32_t q = (uint32_t)(qa + 32 * (int32_t)l) + bpos;
      W2 win[5], core[4], rcore[4];
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) {
        win[pl] = ext96(s_X[pl * 3 + 0][src], s_X[pl * 3 + 1][src], s_X[pl * 3 + 2][src], bpos);
        win[pl].hi &= whi;
        if (pl < 4) {
          core[pl].lo = fsh(win[pl].lo, win[pl].hi, HAWK_PAD) & mlo;
          core[pl].hi = (win[pl].hi >> HAWK_PAD) & mhi;
        }
      }
      const int64_t start = posmap_hint(hs, seg0, seg_end, q), stop = posmap_hint(hs, seg0, seg_end, q + (uint32_t)L);
      const int64_t qr64 = start - ri.startp;
      const bool inr = qr64 >= 0 && qr64 < (int64_t)ri.n_bits;
      const uint32_t qr = inr ? (uint32_t)qr64 : 0u;
      const uint32_t rw = (sd ? ri.bits[1] : ri.bits[0])[qr >> 5];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) rcore[pl] = ext_glb(va.ref[pl], qr);  // fetched whether or not REF has a guide there: one round trip
      const bool has_ref = inr && ((rw >> (qr & 31u)) & 1u);
      bool same = has_ref;
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        rcore[pl].lo &= mlo; rcore[pl].hi &= mhi;
        same = same && rcore[pl].lo == core[pl].lo && rcore[pl].hi == core[pl].hi;
        if (!has_ref) rcore[pl] = core[pl];
      }
      const bool valid = act && !same;
      const uint32_t vinc = group_incl_scan(valid ? 1u : 0u, gl);
      const uint32_t vtot = (uint32_t)__shfl((int)vinc, CS_G - 1, CS_G);
      if (valid && live) {
        const uint64_t k = tb + done + ns + (vinc - 1u);
        if (k >= t_cap) atomicExch(status, -3 /* HAWK_E_CAPACITY */);
        else {
          double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
          if (gp.score_cfdon && has_ref) {
            bool err;
            score = cfdon_from_slices(core, rcore, (uint32_t)sd, L, cfdmask, s_cfd, err);
            if (err && gp.score_cfdon == 1) atomicExch(status, -5 /* HAWK_E_CFD */);
          }
          const bool pamfirst = (p.right != 0) != (sd != 0);
          const int64_t ds = start - ri.startp, de = stop - start;
          if (ds < INT32_MIN || ds > INT32_MAX || de < INT32_MIN || de > INT32_MAX) atomicExch(status, -7 /* HAWK_E_UNSUPPORTED */);
          const uint64_t sc = (uint64_t)__double_as_longlong(score);
          uint4* __restrict__ tp = reinterpret_cast<uint4*>(trows + k);
          tp[0] = make_uint4((uint32_t)((int32_t)(pamfirst ? q : q + (uint32_t)p.guidelen) - o_first), (uint32_t)sd | (has_ref ? 2u : 0u),
                             (uint32_t)(int32_t)ds, (uint32_t)(int32_t)de);
          tp[1] = make_uint4((uint32_t)sc, (uint32_t)(sc >> 32), win[0].lo, win[0].hi);
          tp[2] = make_uint4(win[1].lo, win[1].hi, win[2].lo, win[2].hi);
          tp[3] = make_uint4(win[3].lo, win[3].hi, win[4].lo, win[4].hi);
        }
      }
      ns += vtot;
    }
    if (sd == 0) n0 = ns; else n1 = ns;
    done += ns;
  }
  if (leader) { res[2 * u] = make_uint4(n0, n1, hits, cand); res[2 * u + 1] = make_uint4((uint32_t)tb, h_front, h_behind, 0u); tbase[u] = (uint32_t)tb; }
}

// every instance: rows = those of its cluster; the job's totals get the cluster's own hits and those of the clean run in front of it -
// REF's hits before this cluster minus REF's hits behind the cluster in front (second half of the two entries, the neighbour's through the lane
// below): two 16-byte gathers per instance.  CS_CNT_U instances per thread, loads issued level by level.
#define CS_CNT_U 4
__global__ __launch_bounds__(256) void k_cs_count(HapSetDev hs, VcArgs va, ClDict cd, ScanParams p, const uint4* __restrict__ res,
                                                  const unsigned long long* __restrict__ t_count, uint64_t t_cap,
                                                  uint32_t* __restrict__ group_counts, uint32_t* __restrict__ counts, uint32_t* __restrict__ inst_tb,
                                                  unsigned long long* __restrict__ shards) {
  __shared__ uint32_t s_red[256 / WAVE][2];
  const uint32_t i0 = blockIdx.x * (256 * CS_CNT_U) + threadIdx.x;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const bool ovf = *t_count > t_cap;  // the template rows outgrew their reservation: no table (the host reruns the search)
  uint32_t uid[CS_CNT_U], below[CS_CNT_U];
  int32_t paf[CS_CNT_U];
  bool in[CS_CNT_U];
#pragma unroll
  for (int u = 0; u < CS_CNT_U; ++u) {
    const uint32_t i = i0 + u * 256;
    in[u] = i < cd.n_inst;
    const uint32_t ic = in[u] ? i : 0u;
    uid[u] = cd.inst_uid[ic]; paf[u] = cd.inst_pa[ic];
    below[u] = CL_NONE;
    if (lane == 0 && in[u] && i > 0) below[u] = cd.inst_uid[i - 1];  // (the other lanes' neighbour is the lane below)
  }
  uint4 r[CS_CNT_U], r2[CS_CNT_U];
  uint32_t behind0[CS_CNT_U];
#pragma unroll
  for (int u = 0; u < CS_CNT_U; ++u) {
    const bool has = in[u] && uid[u] != CL_NONE;
    r[u] = make_uint4(0u, 0u, 0u, 0u); r2[u] = r[u]; behind0[u] = 0u;
    if (cd.n_uniq) {
      const uint32_t us = has ? uid[u] : 0u;
      r[u] = res[2 * us]; r2[u] = res[2 * us + 1];
      if (lane == 0 && below[u] != CL_NONE) behind0[u] = res[2 * below[u] + 1].z;
    }
    if (!has) { r[u] = make_uint4(0u, 0u, 0u, 0u); r2[u] = r[u]; }
  }
  uint32_t cand = 0, hits = 0;
#pragma unroll
  for (int u = 0; u < CS_CNT_U; ++u) {
    const uint32_t i = i0 + u * 256;
    uint32_t behind = (uint32_t)__shfl_up((int)r2[u].z, 1);
    if (lane == 0) behind = behind0[u];
    uint32_t rows_i = 0;
    if (in[u]) {
      if (paf[u] < 0) {  // the run in front lies inside every range (k_cl_fill): it has a cluster at either end, and those know
        const uint32_t hc = r2[u].y - behind;
        hits += hc; cand += hc;
      } else {
        const int32_t pa = paf[u], pb = cd.inst_o[i] - (p.L - 1);
        if (pb > pa) {
          const VcRanges rg = cs_ranges(hs, p, cd.inst_row[i]);
          vc_count_run(va, rg, pa, pb, cd.inst_rb[i], cand, hits);
        }
      }
      hits += r[u].z; cand += r[u].w;
      rows_i = ovf ? 0u : r[u].x + r[u].y;
      counts[i] = rows_i;
      inst_tb[i] = r2[u].x;  // the instance's first template row: the emit pass reads it next to the count instead of going through the cluster
    }
    // what the offset scan runs over: the rows of 64 consecutive instances (one wave here, one wave's contiguous stretch of the table in
    // the emit pass) - 1.4 x 10^5 entries on C3 instead of 8.8 x 10^6
    const uint32_t gsum = wave_sum(rows_i);
    const uint32_t iw = blockIdx.x * (256 * CS_CNT_U) + u * 256 + (threadIdx.x & ~(uint32_t)(WAVE - 1));
    if (lane == 0 && iw < cd.n_inst) group_counts[iw / WAVE] = gsum;
  }
  const uint32_t a = wave_sum(cand), b = wave_sum(hits);
  if (lane == 0) { s_red[threadIdx.x / WAVE][0] = a; s_red[threadIdx.x / WAVE][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t0 = 0, t1 = 0;
#pragma unroll
    for (int wv = 0; wv < 256 / WAVE; ++wv) { t0 += s_red[wv][0]; t1 += s_red[wv][1]; }
    if (t0 | t1) {
      atomicAdd(&shards[(blockIdx.x & 255u) * 2 + 0], (unsigned long long)t0);
      atomicAdd(&shards[(blockIdx.x & 255u) * 2 + 1], (unsigned long long)t1);
    }
  }
}

// 16 bytes to a 16-byte-aligned address, past the caches' allocation (`nt`): the table is written once and read by a later kernel.
// The four dword stores to consecutive addresses are merged into one global_store_dwordx4 ... nt by the compiler.
__device__ __forceinline__ void nt_store4(uint32_t* q, const uint4& v) {
  __builtin_nontemporal_store(v.x, q); __builtin_nontemporal_store(v.y, q + 1);
  __builtin_nontemporal_store(v.z, q + 2); __builtin_nontemporal_store(v.w, q + 3);
}
// ---- packed rows (round 4): the guide table of a cluster search as ONE array of 64-byte rows -----------------------------
// {pos, strand | flags << 1 | haplotype row << 9, start - startp, stop - start} {cfdon, win0} {win1, win2} {win3, win4}: a template
// row with two words patched.  A wave's rows are contiguous, four lanes move one row (16 bytes each), so every store
// instruction writes 1 KB of consecutive addresses and the whole table is a single linear write stream - twelve column
// streams whose relative placement decided the speed before (profiles/r03_csearch_ablation.txt).
// NI x 64 consecutive instances per wave, their rows one contiguous stretch of the table.  What bounds the pass is not bytes but
// how long a wave's loads take while 1.8 GB of stores are queued at the memory controllers (the same kernel runs 0.36 ms with its
// 160 MB read set cache-resident and 0.53 with it in HBM): so a wave asks for all its instance data at once, up front, and
// keeps the template loads of the next 128 rows in flight while it stores the current 128.
template <int NI, int CE_CH>
__global__ __launch_bounds__(256) void k_cs_emit_rows(uint32_t n_inst, const uint32_t* __restrict__ inst_row, const int32_t* __restrict__ inst_o,
                                                      const uint32_t* __restrict__ counts, const uint32_t* __restrict__ inst_tb,
                                                      const CsRow* __restrict__ trows, const uint64_t* __restrict__ offsets,
                                                      const unsigned long long* __restrict__ t_count, uint64_t t_cap, uint4* __restrict__ rows,
                                                      uint64_t cap, int* status) {
  // everything below is private to a wave (its own LDS slices, no workgroup barrier): the four waves of a workgroup drift apart freely
  constexpr int NE = NI * WAVE;
  constexpr int CE_SUB = CE_CH / 16;  // 16-byte loads per lane and chunk
  __shared__ uint32_t s_ex[4][NE + 1];
  __shared__ uint32_t s_tb[4][NE], s_h[4][NE];
  __shared__ int32_t s_dq[4][NE];
  __shared__ uint32_t s_src[4][2][CE_CH], s_rh[4][2][CE_CH];
  __shared__ int32_t s_rdq[4][2][CE_CH];
  if (*t_count > t_cap) return;  // the template rows outgrew their reservation (k_cs_count left no counts): the host reruns the search
  const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const uint32_t i0 = (blockIdx.x * 4 + wv) * NE;  // the wave's first instance
  if (i0 >= n_inst) return;
  // per instance: four independent streaming loads (k_cs_count left the row count and the first template row of each instance, so
  // nothing here waits for a gather through the cluster number); lane l takes instances i0 + 64 c + l
  uint32_t cnt[NI], tb[NI], h[NI];
  int32_t dq[NI];
#pragma unroll
  for (int c = 0; c < NI; ++c) {
    const uint32_t i = i0 + c * WAVE + lane;
    const bool in = i < n_inst;
    const uint32_t ii = in ? i : n_inst - 1;
    cnt[c] = counts[ii]; tb[c] = inst_tb[ii]; h[c] = inst_row[ii]; dq[c] = inst_o[ii];
    if (!in) cnt[c] = 0;
  }
  // the wave's first row
  const uint64_t o0 = offsets[i0 / WAVE];  // one offset per 64 instances (k_cs_count's groups); wave-uniform: a scalar load
  uint32_t Wt = 0;
#pragma unroll
  for (int c = 0; c < NI; ++c) {
    const uint32_t inc = wave_incl_scan(cnt[c]);
    s_ex[wv][c * WAVE + lane] = Wt + inc - cnt[c];
    s_tb[wv][c * WAVE + lane] = tb[c]; s_h[wv][c * WAVE + lane] = h[c] << HAWK_ROW_HAP_SHIFT; s_dq[wv][c * WAVE + lane] = dq[c];
    Wt += (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
  }
  if (Wt == 0) return;  // wave-uniform
  if (o0 + Wt > cap) { if (lane == 0) atomicCAS(status, 0, -3 /* HAWK_E_CAPACITY */); return; }
  __builtin_amdgcn_wave_barrier();
  const uint32_t qd = lane & 3u, rq = lane >> 2;
  uint4* __restrict__ const dst = rows + o0 * 4 + qd;
  // which template row each of a chunk's rows copies, its haplotype row and position shift (rows past the wave's last are
  // clamped onto it: every lane then loads a valid template row and the loads need no branch)
  auto meta = [&](uint32_t c0, uint32_t buf) {
#pragma unroll
    for (uint32_t sub = 0; sub < (CE_CH + WAVE - 1) / WAVE; ++sub) {
      const uint32_t r = sub * WAVE + lane;
      if (CE_CH < WAVE && r >= CE_CH) break;
      const uint32_t t = c0 + r < Wt ? c0 + r : Wt - 1;
      uint32_t l = 0;
#pragma unroll
      for (uint32_t step = NE / 2; step; step >>= 1) l += (s_ex[wv][l + step] <= t) ? step : 0u;  // l + step <= NE - 1
      s_src[wv][buf][r] = s_tb[wv][l] + (t - s_ex[wv][l]);
      s_rh[wv][buf][r] = s_h[wv][l];
      s_rdq[wv][buf][r] = s_dq[wv][l];
    }
    __builtin_amdgcn_wave_barrier();
  };
  // four lanes per row, 16 bytes each
  auto fetch = [&](uint4 (&v)[CE_SUB], uint32_t buf) {
#pragma unroll
    for (uint32_t j = 0; j < CE_SUB; ++j) v[j] = reinterpret_cast<const uint4*>(trows + s_src[wv][buf][j * 16 + rq])[qd];
  };
  auto store = [&](const uint4 (&v)[CE_SUB], uint32_t c0, uint32_t buf) {
#pragma unroll
    for (uint32_t j = 0; j < CE_SUB; ++j) {
      const uint32_t r = j * 16 + rq;
      uint4 w = v[j];
      if (qd == 0) {
        w.x = (uint32_t)((int32_t)w.x + s_rdq[wv][buf][r]);
        w.y |= s_rh[wv][buf][r];
      }
      if (c0 + r < Wt) nt_store4(reinterpret_cast<uint32_t*>(dst + (size_t)(c0 + r) * 4), w);
    }
  };
  uint4 va[CE_SUB], vb[CE_SUB];
  meta(0, 0);
  fetch(va, 0);
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < Wt; c0 += 2 * CE_CH) {
    const bool more1 = c0 + CE_CH < Wt, more2 = c0 + 2 * CE_CH < Wt;  // wave-uniform
    if (more1) { meta(c0 + CE_CH, 1); fetch(vb, 1); }
    store(va, c0, 0);
    __builtin_amdgcn_wave_barrier();
    if (more1) {
      if (more2) { meta(c0 + 2 * CE_CH, 0); fetch(va, 0); }
      store(vb, c0 + CE_CH, 1);
      __builtin_amdgcn_wave_barrier();
    }
  }
}
void hawk_launch_cs_emit_rows(hipStream_t st, const ClDict& cd, const uint32_t* counts, const uint32_t* inst_tb, const void* trows, const uint64_t* offsets,
                              const unsigned long long* t_count, uint64_t t_cap, void* rows, uint64_t cap, int* status) {
  if (!cd.n_inst) return;
  // 256 instances per wave, 256 rows in flight per wave (184 VGPRs: two waves per SIMD).  Measured on C3 (profiles/r04_emit_rows.md):
  // 64 / 128 / 256 rows in flight at 8 / 4 / 2 waves per SIMD: 0.53 / 0.47 / 0.45 ms before the offset scan went to one entry per
  // 64 instances, 0.39 (128 rows, 128 instances) / 0.38 after - few waves with long contiguous bursts write best
  constexpr int NI = 4, CH = 256;
  hipLaunchKernelGGL((k_cs_emit_rows<NI, CH>), dim3((cd.n_inst + 256 * NI - 1) / (256 * NI)), dim3(256), 0, st, cd.n_inst, cd.inst_row, cd.inst_o, counts,
                     inst_tb, static_cast<const CsRow*>(trows), offsets, t_count, t_cap, static_cast<uint4*>(rows), cap, status);
}

void hawk_launch_cs_templates(hipStream_t st, const HapSetDev& hs, const VcArgs& va, const ClDict& cd, const ScanParams& p, const GuideParams& gp,
                              const RefInfo& ri, void* res, uint32_t* tbase, void* trows, unsigned long long* t_count, uint64_t t_cap, int* status) {
  if (!cd.n_uniq) return;
  const uint32_t nb = (uint32_t)(((uint64_t)cd.n_uniq * CS_G + 255) / 256);
  hipLaunchKernelGGL(k_cs_templates, dim3(nb), dim3(256), 0, st, hs, va, cd, p, gp, ri, static_cast<uint4*>(res), tbase, static_cast<CsRow*>(trows),
                     t_count, t_cap, status);
}
void hawk_launch_cs_count(hipStream_t st, const HapSetDev& hs, const VcArgs& va, const ClDict& cd, const ScanParams& p, const void* res,
                          const unsigned long long* t_count, uint64_t t_cap, uint32_t* group_counts, uint32_t* counts, uint32_t* inst_tb,
                          unsigned long long* shards) {
  if (!cd.n_inst) return;
  hipLaunchKernelGGL(k_cs_count, dim3((cd.n_inst + 256 * CS_CNT_U - 1) / (256 * CS_CNT_U)), dim3(256), 0, st, hs, va, cd, p, static_cast<const uint4*>(res), t_count, t_cap,
                     group_counts, counts, inst_tb, shards);
}

// columns -> packed rows (REF's rows, which the plane kernels write as columns; a columnar table before an exchange) and back
__global__ __launch_bounds__(256) void k_rows_pack(GuideCols c, const uint64_t* __restrict__ n_dev, uint64_t n_host, uint4* __restrict__ rows, int64_t startp,
                                                   int* status) {
  const uint64_t n = n_dev ? *n_dev : n_host;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t start = c.start[i], ds = start - startp, de = c.stop[i] - start;
  const uint32_t h = c.hap[i];
  if (ds < INT32_MIN || ds > INT32_MAX || de < INT32_MIN || de > INT32_MAX || h >= HAWK_ROW_MAX_HAP) atomicCAS(status, 0, -7 /* HAWK_E_UNSUPPORTED */);
  const uint64_t sc = (uint64_t)__double_as_longlong(c.cfdon[i]);
  uint64_t w[HAWK_PLANES];
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) w[pl] = c.win[(size_t)pl * c.cap + i];
  uint4* __restrict__ r = rows + i * 4;
  r[0] = make_uint4(c.pos[i], (uint32_t)(c.strand[i] & 1u) | ((uint32_t)c.flags[i] << 1) | (h << HAWK_ROW_HAP_SHIFT), (uint32_t)(int32_t)ds, (uint32_t)(int32_t)de);
  r[1] = make_uint4((uint32_t)sc, (uint32_t)(sc >> 32), (uint32_t)w[0], (uint32_t)(w[0] >> 32));
  r[2] = make_uint4((uint32_t)w[1], (uint32_t)(w[1] >> 32), (uint32_t)w[2], (uint32_t)(w[2] >> 32));
  r[3] = make_uint4((uint32_t)w[3], (uint32_t)(w[3] >> 32), (uint32_t)w[4], (uint32_t)(w[4] >> 32));
}
__global__ __launch_bounds__(256) void k_rows_unpack(const uint4* __restrict__ rows, uint64_t n, int64_t startp, GuideCols d) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4* __restrict__ r = rows + i * 4;
  const uint4 a = r[0], b = r[1], cc = r[2], e = r[3];
  const int64_t start = startp + (int64_t)(int32_t)a.z;
  if (d.hap) d.hap[i] = a.y >> HAWK_ROW_HAP_SHIFT;
  if (d.pos) d.pos[i] = a.x;
  if (d.strand) d.strand[i] = (uint8_t)(a.y & 1u);
  if (d.flags) d.flags[i] = (uint8_t)((a.y >> 1) & 0xffu);
  if (d.start) d.start[i] = start;
  if (d.stop) d.stop[i] = start + (int64_t)(int32_t)a.w;
  if (d.cfdon) d.cfdon[i] = __longlong_as_double((long long)((uint64_t)b.x | ((uint64_t)b.y << 32)));
  if (d.win) {
    d.win[i] = (uint64_t)b.z | ((uint64_t)b.w << 32);
    d.win[d.cap + i] = (uint64_t)cc.x | ((uint64_t)cc.y << 32);
    d.win[2 * d.cap + i] = (uint64_t)cc.z | ((uint64_t)cc.w << 32);
    d.win[3 * d.cap + i] = (uint64_t)e.x | ((uint64_t)e.y << 32);
    d.win[4 * d.cap + i] = (uint64_t)e.z | ((uint64_t)e.w << 32);
  }
}
void hawk_launch_rows_pack(hipStream_t st, const GuideCols& src, const uint64_t* n_dev, uint64_t n_host, uint64_t max_rows, uint4* rows, int64_t startp,
                           int* status) {
  if (!max_rows) return;
  hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)((max_rows + 255) / 256)), dim3(256), 0, st, src, n_dev, n_host, rows, startp, status);
}
void hawk_launch_rows_unpack(hipStream_t st, const uint4* rows, uint64_t n, int64_t startp, const GuideCols& dst) {
  if (!n) return;
  hipLaunchKernelGGL(k_rows_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rows, n, startp, dst);
}

// ---- the collapse of a cluster-searched table (hawk_api.hip: collapse_by_templates) ---------------------------------------
// REF's rows + the template rows as a table of their own: the grouping compares nothing that differs between a template row and
// its copies (start, stop, strand, REF-or-not, the windows)
__global__ __launch_bounds__(256) void k_cc_ucnt(const uint4* __restrict__ res, uint32_t nu, uint32_t* __restrict__ cnt) {
  const uint32_t u = blockIdx.x * 256 + threadIdx.x;
  if (u < nu) { const uint4 r = res[2 * u]; cnt[u] = r.x + r.y; }
}
// mini row r0 + moff[u] + k is row k of distinct cluster u: template row tbase[u] + k
__global__ __launch_bounds__(256) void k_cc_mini(GuideCols c, uint64_t r0, const CsRow* __restrict__ trows, uint64_t t_rows,
                                                 const uint64_t* __restrict__ moff, const uint32_t* __restrict__ tbase, uint32_t nu, int64_t startp,
                                                 GuideCols m) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= r0 + t_rows) return;
  if (i < r0) {
    m.hap[i] = gc_hap(c, i); m.pos[i] = gc_pos(c, i); m.strand[i] = (uint8_t)gc_strand(c, i); m.start[i] = gc_start(c, i); m.stop[i] = gc_stop(c, i);
    m.flags[i] = (uint8_t)gc_flags(c, i); m.cfdon[i] = gc_cfdon(c, i);
#pragma unroll
    for (int pl = 0; pl < HAWK_PLANES; ++pl) m.win[(size_t)pl * m.cap + i] = gc_win(c, pl, i);
    return;
  }
  uint32_t lo = 0, hi = nu;  // last u with moff[u] <= i - r0
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (moff[mid] <= i - r0) lo = mid; else hi = mid;
  }
  const uint4* __restrict__ tp = reinterpret_cast<const uint4*>(trows + (tbase[lo] + (i - r0 - moff[lo])));
  const uint4 a = tp[0], b = tp[1], cc = tp[2], d = tp[3];
  const int64_t start = startp + (int64_t)(int32_t)a.z;
  m.hap[i] = 1u;  // any row but REF's: the grouping asks for the origin only
  m.pos[i] = a.x; m.strand[i] = (uint8_t)(a.y & 1u); m.flags[i] = (uint8_t)((a.y >> 1) & 0xffu);
  m.start[i] = start; m.stop[i] = start + (int64_t)(int32_t)a.w;
  m.cfdon[i] = __longlong_as_double((long long)((uint64_t)b.x | ((uint64_t)b.y << 32)));
  m.win[0 * m.cap + i] = (uint64_t)b.z | ((uint64_t)b.w << 32);
  m.win[1 * m.cap + i] = (uint64_t)cc.x | ((uint64_t)cc.y << 32);
  m.win[2 * m.cap + i] = (uint64_t)cc.z | ((uint64_t)cc.w << 32);
  m.win[3 * m.cap + i] = (uint64_t)d.x | ((uint64_t)d.y << 32);
  m.win[4 * m.cap + i] = (uint64_t)d.z | ((uint64_t)d.w << 32);
}
// group number of every mini row from the mini collapse's (permutation, CSR offsets)
__global__ __launch_bounds__(256) void k_cc_gidm(const uint32_t* __restrict__ perm, const uint64_t* __restrict__ goff, uint64_t nm, uint64_t G,
                                                 uint32_t* __restrict__ gidm) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nm) return;
  uint64_t lo = 0, hi = G;  // last g with goff[g] <= j
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (goff[mid] <= j) lo = mid; else hi = mid;
  }
  gidm[perm[j]] = (uint32_t)lo;
}
// every table row's group number = its template row's (REF's rows: their own), next to its index: the pairs the sort orders
__global__ __launch_bounds__(256) void k_cs_gid(ClDict cd, const uint4* __restrict__ res, const uint64_t* __restrict__ moff,
                                                const uint64_t* __restrict__ offsets, uint64_t r0, uint64_t n, const uint32_t* __restrict__ gidm,
                                                uint32_t* __restrict__ gid, uint32_t* __restrict__ vals) {
  __shared__ uint32_t s_ex[4][WAVE + 1];
  __shared__ uint32_t s_tb[4][WAVE];
  const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t cnt = 0, tb = 0;
  if (i < cd.n_inst) {
    const uint32_t u = cd.inst_uid[i];
    if (u != CL_NONE) { const uint4 r = res[2 * u]; cnt = r.x + r.y; tb = (uint32_t)moff[u]; }
  }
  const uint32_t inc = wave_incl_scan(cnt);
  const uint32_t Wt = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
  const uint32_t i0 = blockIdx.x * 256 + wv * WAVE;
  const uint64_t o0 = i0 < cd.n_inst ? offsets[i0 / WAVE] : 0;  // one offset per 64 instances (k_cs_count's groups)
  s_ex[wv][lane] = inc - cnt;
  s_tb[wv][lane] = tb;
  __syncthreads();
  if (o0 + Wt > n) return;  // cannot happen for the table these offsets were scanned for
  for (uint32_t t = lane; t < Wt; t += WAVE) {
    uint32_t l = 0;
#pragma unroll
    for (uint32_t step = WAVE / 2; step; step >>= 1) l += (s_ex[wv][l + step] <= t) ? step : 0u;
    const uint64_t o = o0 + t;
    gid[o] = gidm[r0 + s_tb[wv][l] + (t - s_ex[wv][l])];
    vals[o] = (uint32_t)o;
  }
}
__global__ __launch_bounds__(256) void k_cc_refgid(uint64_t r0, const uint32_t* __restrict__ gidm, uint32_t* __restrict__ gid, uint32_t* __restrict__ vals) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < r0) { gid[i] = gidm[i]; vals[i] = (uint32_t)i; }
}
void hawk_launch_cc_ucnt(hipStream_t st, const void* res, uint32_t nu, uint32_t* cnt) {
  hipLaunchKernelGGL(k_cc_ucnt, dim3((nu + 255) / 256), dim3(256), 0, st, static_cast<const uint4*>(res), nu, cnt);
}
void hawk_launch_cc_mini(hipStream_t st, const GuideCols& c, uint64_t r0, const void* trows, uint64_t t_rows, const uint64_t* moff, const uint32_t* tbase,
                         uint32_t nu, int64_t startp, GuideCols m) {
  const uint64_t nm = r0 + t_rows;
  hipLaunchKernelGGL(k_cc_mini, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, c, r0, static_cast<const CsRow*>(trows), t_rows, moff, tbase, nu,
                     startp, m);
}
void hawk_launch_cc_gidm(hipStream_t st, const uint32_t* perm, const uint64_t* goff, uint64_t nm, uint64_t G, uint32_t* gidm) {
  hipLaunchKernelGGL(k_cc_gidm, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, perm, goff, nm, G, gidm);
}
void hawk_launch_cs_gid(hipStream_t st, const ClDict& cd, const void* res, const uint64_t* moff, const uint64_t* offsets, uint64_t r0, uint64_t n,
                        const uint32_t* gidm, uint32_t* gid, uint32_t* vals) {
  if (r0) hipLaunchKernelGGL(k_cc_refgid, dim3((unsigned)((r0 + 255) / 256)), dim3(256), 0, st, r0, gidm, gid, vals);
  if (cd.n_inst) hipLaunchKernelGGL(k_cs_gid, dim3((cd.n_inst + 255) / 256), dim3(256), 0, st, cd, static_cast<const uint4*>(res), moff, offsets, r0, n,
                                    gidm, gid, vals);
}
