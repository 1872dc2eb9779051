// hawk_collapse.hip — f2: which guide rows the report merges (reports.py:958-1008
// `_collapse_report_entries`: a pandas groupby over chr, start, stop, sgRNA_sequence, pam, strand,
// scores, gc_content, origin).  Scores and GC are functions of the sequence and of the REF guide at
// the same (start, strand), so two rows merge iff they agree in start, stop, strand, origin (REF
// haplotype or not) and in the case-preserving spacer+PAM — on the device: the five L-bit core
// slices of the window planes (the V plane is the case).  Equality of the stored (+ strand) cores is
// equality of the reverse-complemented guides the report prints.
//
//   k_collapse_keys   row -> 64-bit sort key  (start - base) << 32 | strand << 31 | hash31(core, stop, origin), and 32
//                     further bits of the same 64-bit hash of the row's identity (id2, table order)
//   rocprim::radix_sort_pairs (key, row id) over the bits in use - the low hash bits are dropped (zero in the key) so
//                     that start + strand + hash fill a whole number of 8-bit passes with >= 24 hash bits (a 4 Mb tile:
//                     48 bits, 6 passes instead of 7); stable, so equal keys keep table order (haplotype ascending)
//   k_collapse_heads  neighbours in sorted order open a group when their keys differ or, with equal keys, their id2
//                     (two 4-byte gathers); counts heads by key and heads by identity - if they differ, two different
//                     rows of one (start, strand) collided in the 31 key bits and may interleave: the host layer
//                     re-runs with another seed.  Two different rows are merged only if they agree in start, strand
//                     and 63 hash bits (~10^-12 for a whole C4 contig); HAWK_COLLAPSE_EXACT=1 compares the FULL row
//                     keys instead (64-byte records written by k_collapse_keys, two 64-byte gathers per row: 8.5 of
//                     the 18 ms a 1.1 x 10^8-row tile took) - the tests run both
//   rocprim::exclusive_scan of the flags, k_collapse_groups: CSR offsets + GC counts of each group's guide
//                     (annotation.py:513-541: gc_fraction(guide.guide); Biopython's default drops ambiguous bases)
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "hawk_device.h"

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

struct RowKey {
  int64_t start, stop;
  uint64_t core[HAWK_PLANES];
  uint32_t sr;  // strand | origin << 1
};
// The compared slice is the spacer+PAM core widened by `up` bases on the guide's 5' side and `down` on its 3' side
// (0/0: the report without model scores; 4/3: the k-mer the model scorers read, scoring.py:50-67, so that rows whose
// flanks differ - and whose Azimuth / DeepCpf1 scores may therefore differ - stay separate, reports.py:978-1003).
// Rows are stored on the + strand: a strand-1 guide's 5' side is the window's right side.
__device__ __forceinline__ RowKey row_key(const GuideCols& c, const uint8_t* __restrict__ is_ref, uint64_t i, int L, int up, int down) {
  RowKey k;
  k.start = gc_start(c, i);
  k.stop = gc_stop(c, i);
  const uint32_t strand = gc_strand(c, i);
  k.sr = strand | ((uint32_t)(is_ref[gc_hap(c, i)] != 0) << 1);
  const int fl = strand ? down : up, fr = strand ? up : down;
  const int width = L + fl + fr;
  const uint64_t mask = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) k.core[pl] = (gc_win(c, pl, i) >> (HAWK_PAD - fl)) & mask;
  return k;
}
__device__ __forceinline__ bool same_row(const RowKey& a, const RowKey& b) {
  bool s = a.start == b.start && a.stop == b.stop && a.sr == b.sr;
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) s = s && a.core[pl] == b.core[pl];
  return s;
}

__global__ __launch_bounds__(256) void k_collapse_keys(GuideCols c, const uint8_t* __restrict__ is_ref, uint64_t n, int L, int up, int down,
                                                       int64_t base, uint64_t seed, uint32_t hash_mask, uint32_t id_mask,
                                                       uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ id2,
                                                       ulonglong4* __restrict__ full /* exact mode, else null */) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const RowKey k = row_key(c, is_ref, i, L, up, down);
  // the full key as one 64-byte record: the head pass compares neighbours of the SORTED order, i.e. rows scattered
  // over the table - one line per row here instead of nine (one per column) there
  if (full) {
    full[2 * i] = make_ulonglong4((unsigned long long)k.start, (unsigned long long)k.stop, (unsigned long long)k.sr, k.core[0]);
    full[2 * i + 1] = make_ulonglong4(k.core[1], k.core[2], k.core[3], k.core[4]);
  }
  uint64_t h = seed;
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) h = mix64(h ^ k.core[pl]);
  h = mix64(h ^ (uint64_t)k.stop ^ ((uint64_t)(k.sr >> 1) << 63));
  keys[i] = ((uint64_t)(k.start - base) << 32) | ((uint64_t)(k.sr & 1u) << 31) | ((h >> 33) & hash_mask);
  vals[i] = (uint32_t)i;
  id2[i] = (uint32_t)h & id_mask;  // id_mask / hash_mask = 0: the test that forces hash collisions (HAWK_COLLAPSE_WEAK_HASH)
}

__device__ __forceinline__ bool same4(const ulonglong4& a, const ulonglong4& b) {
  return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
}
__global__ __launch_bounds__(256) void k_collapse_heads(const ulonglong4* __restrict__ full, const uint32_t* __restrict__ id2, uint64_t n,
                                                        const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                        uint32_t* __restrict__ flags, unsigned long long* __restrict__ counters) {
  __shared__ uint32_t s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t hk = 0, hf = 0;
  if (j < n) {
    if (j == 0) {
      hk = hf = 1;
    } else {
      hk = keys[j] != keys[j - 1];
      if (!hk) {
        const uint64_t a = vals[j], b = vals[j - 1];
        hf = full ? !(same4(full[2 * a], full[2 * b]) && same4(full[2 * a + 1], full[2 * b + 1])) : id2[a] != id2[b];
      } else {
        hf = 1;
      }
    }
    flags[j] = hf;
  }
  const unsigned long long bk = __ballot(hk), bf = __ballot(hf);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt[0], (uint32_t)__popcll(bk)); atomicAdd(&s_cnt[1], (uint32_t)__popcll(bf)); }
  __syncthreads();
  if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// flags (0/1) and their exclusive scan gidx: a head at sorted position j opens group gidx[j]
__global__ __launch_bounds__(256) void k_collapse_groups(GuideCols c, uint64_t n, const uint32_t* __restrict__ vals,
                                                         const uint32_t* __restrict__ flags, const uint32_t* __restrict__ gidx,
                                                         int guidelen, int pamlen, int right, uint64_t* __restrict__ group_off,
                                                         uint8_t* __restrict__ gc_num, uint8_t* __restrict__ gc_den) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n || !flags[j]) return;
  const uint32_t g = gidx[j];
  group_off[g] = j;
  const uint64_t r = vals[j];
  // the spacer inside the stored (+ strand) core: behind the PAM when the PAM comes first (right ^ strand)
  const bool pamfirst = (right != 0) != (gc_strand(c, r) != 0);
  const int sh = HAWK_PAD + (pamfirst ? pamlen : 0);
  const uint64_t m = guidelen >= 64 ? ~0ull : ((1ull << guidelen) - 1ull);
  const uint64_t A = (gc_win(c, 0, r) >> sh) & m, C = (gc_win(c, 1, r) >> sh) & m, G = (gc_win(c, 2, r) >> sh) & m,
                 T = (gc_win(c, 3, r) >> sh) & m;
  const uint64_t gc = (C | G) & ~(A | T);  // C, G, S
  const uint64_t at = (A | T) & ~(C | G);  // A, T, W
  gc_num[g] = (uint8_t)__popcll(gc);
  gc_den[g] = (uint8_t)__popcll(gc | at);
}

// Every member of a group against the group's first member on the FULL key (start, stop, strand, origin, the five core
// slices): grouping by hash - 63 bits in either path above - is thereby checked, not trusted; one mismatch sends the call
// to the exact path.  `grp_a[j] + (grp_b ? grp_b[j] - 1 : 0)` = the group of sorted position j (sort path: exclusive scan of
// the head flags + the flag - 1; hash path: the sorted group numbers).  The head's key is shared by the group's members
// (L2), the member's own key is the one scattered read per row.
__global__ __launch_bounds__(256) void k_collapse_verify(GuideCols c, const uint8_t* __restrict__ is_ref, uint64_t n, int L, int up, int down,
                                                         const uint32_t* __restrict__ perm, const uint32_t* __restrict__ grp_a,
                                                         const uint32_t* __restrict__ grp_b, const uint64_t* __restrict__ group_off,
                                                         unsigned long long* __restrict__ mismatches) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (j < n) {
    const uint32_t g = grp_b ? grp_a[j] + grp_b[j] - 1u : grp_a[j];
    const uint64_t hj = group_off[g];
    if (hj != j) bad = !same_row(row_key(c, is_ref, perm[j], L, up, down), row_key(c, is_ref, perm[hj], L, up, down));
  }
  const unsigned long long b = __ballot(bad);
  if (b && (threadIdx.x & 63) == 0) atomicAdd(mismatches, (unsigned long long)__popcll(b));
}
// The same check in TABLE order, for the hash-table path, which knows every row's group (slot -> group number): first the
// full key of every group's first member as one 64-byte record per group (k_collapse_head_keys: a few 10^5 records that
// stay in L2), then every row reads its own key where it lies (coalesced) and its group's record - three gathers per row
// instead of two scattered keys of nine columns each.
__global__ __launch_bounds__(256) void k_collapse_head_keys(GuideCols c, const uint8_t* __restrict__ is_ref, uint32_t G, int L, int up, int down,
                                                            const uint32_t* __restrict__ perm, const uint64_t* __restrict__ group_off,
                                                            ulonglong4* __restrict__ full) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const RowKey k = row_key(c, is_ref, perm[group_off[g]], L, up, down);
  full[2 * (size_t)g] = make_ulonglong4((unsigned long long)k.start, (unsigned long long)k.stop, (unsigned long long)k.sr, k.core[0]);
  full[2 * (size_t)g + 1] = make_ulonglong4(k.core[1], k.core[2], k.core[3], k.core[4]);
}
__global__ __launch_bounds__(256) void k_collapse_verify_rows(GuideCols c, const uint8_t* __restrict__ is_ref, uint64_t n, int L, int up, int down,
                                                              const uint32_t* __restrict__ slot_of_row, const uint32_t* __restrict__ slot2rank,
                                                              const ulonglong4* __restrict__ full, unsigned long long* __restrict__ mismatches) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i < n) {
    const RowKey k = row_key(c, is_ref, i, L, up, down);
    const size_t g = slot2rank[slot_of_row[i]];
    const ulonglong4 a = full[2 * g], b = full[2 * g + 1];
    bad = !(a.x == (unsigned long long)k.start && a.y == (unsigned long long)k.stop && a.z == (unsigned long long)k.sr && a.w == k.core[0] &&
            b.x == k.core[1] && b.y == k.core[2] && b.z == k.core[3] && b.w == k.core[4]);
  }
  const unsigned long long bl = __ballot(bad);
  if (bl && (threadIdx.x & 63) == 0) atomicAdd(mismatches, (unsigned long long)__popcll(bl));
}
void hawk_launch_collapse_verify_rows(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, uint32_t G, int guidelen, int pamlen,
                                      int flank_up, int flank_down, const uint32_t* perm, const uint32_t* slot_of_row, const uint32_t* slot2rank,
                                      const uint64_t* group_off, void* full /* 64 B per group */, unsigned long long* mismatches) {
  if (!n || !G) return;
  hipLaunchKernelGGL(k_collapse_head_keys, dim3((G + 255) / 256), dim3(256), 0, st, c, is_ref, G, guidelen + pamlen, flank_up, flank_down, perm,
                     group_off, (ulonglong4*)full);
  hipLaunchKernelGGL(k_collapse_verify_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, is_ref, n, guidelen + pamlen, flank_up,
                     flank_down, slot_of_row, slot2rank, (const ulonglong4*)full, mismatches);
}
void hawk_launch_collapse_verify(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int flank_up,
                                 int flank_down, const uint32_t* perm, const uint32_t* grp_a, const uint32_t* grp_b, const uint64_t* group_off,
                                 unsigned long long* mismatches) {
  if (!n) return;
  hipLaunchKernelGGL(k_collapse_verify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, is_ref, n, guidelen + pamlen, flank_up,
                     flank_down, perm, grp_a, grp_b, group_off, mismatches);
}

// whole rows of a set against each other, all five planes: the check behind collapsing haplotypes on their content hash
__global__ __launch_bounds__(256) void k_rows_equal(HapSetDev hs, const uint32_t* __restrict__ ra, const uint32_t* __restrict__ rb,
                                                    uint8_t* __restrict__ equal) {
  __shared__ uint32_t s_diff;
  if (threadIdx.x == 0) s_diff = 0;
  __syncthreads();
  const size_t a = (size_t)ra[blockIdx.x] * hs.S, b = (size_t)rb[blockIdx.x] * hs.S;
  uint32_t d = 0;
  for (uint32_t w = threadIdx.x; w < hs.S; w += 256)
#pragma unroll
    for (int pl = 0; pl < HAWK_PLANES; ++pl) d |= hs.plane[pl][a + w] ^ hs.plane[pl][b + w];
  if (d) atomicOr(&s_diff, 1u);
  __syncthreads();
  if (threadIdx.x == 0) equal[blockIdx.x] = s_diff ? 0 : 1;
}
void hawk_launch_rows_equal(hipStream_t st, const HapSetDev& hs, uint32_t n_pairs, const uint32_t* ra, const uint32_t* rb, uint8_t* equal) {
  if (n_pairs) hipLaunchKernelGGL(k_rows_equal, dim3(n_pairs), dim3(256), 0, st, hs, ra, rb, equal);
}

// bytes of the full-key records (64 per row)
size_t hawk_collapse_full_bytes(uint64_t n) { return (size_t)n * 64; }

size_t hawk_collapse_temp_bytes(uint64_t n, unsigned begin_bit, unsigned end_bit) {
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, begin_bit,
                                  end_bit, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
  return a > b ? a : b;
}

// keys/vals: [2][n] ping-pong; flags, gidx: [n]; counters: [2], zeroed by the caller; group_off must hold
// n + 1 entries (the number of groups is only known afterwards)
int hawk_launch_collapse(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int right,
                         int flank_up, int flank_down, int64_t base, unsigned begin_bit, unsigned end_bit, uint64_t seed, void* temp, size_t temp_bytes, uint64_t* keys, uint32_t* vals,
                         uint32_t* flags, uint32_t* gidx, unsigned long long* counters, uint64_t* group_off, uint8_t* gc_num,
                         uint8_t* gc_den, uint32_t* id2, void* full, int weak_hash) {
  const int L = guidelen + pamlen;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_collapse_keys, grid, block, 0, st, c, is_ref, n, L, flank_up, flank_down, base, seed,
                     weak_hash ? 0u : (~((1u << begin_bit) - 1u) & 0x7fffffffu), weak_hash ? 0u : 0xffffffffu, keys, vals, id2, (ulonglong4*)full);
  size_t tb = temp_bytes;
  if (rocprim::radix_sort_pairs(temp, tb, keys, keys + n, vals, vals + n, n, begin_bit, end_bit, st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_collapse_heads, grid, block, 0, st, (const ulonglong4*)full, id2, n, keys + n, vals + n, flags, counters);
  tb = temp_bytes;
  if (rocprim::exclusive_scan(temp, tb, flags, gidx, 0u, n, rocprim::plus<uint32_t>(), st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_collapse_groups, grid, block, 0, st, c, n, vals + n, flags, gidx, guidelen, pamlen, right, group_off, gc_num,
                     gc_den);
  return 0;
}

// ---- group export: one representative row per group + every row's haplotype in group order, so that the host
// downloads group-level columns (74 B per report row) and 4 B per guide row instead of the whole table
__global__ __launch_bounds__(256) void k_collapse_export(GuideCols c, uint64_t n, uint64_t ng, const uint32_t* __restrict__ perm,
                                                         const uint64_t* __restrict__ group_off, GuideCols rep,
                                                         uint32_t* __restrict__ member_hap) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) member_hap[j] = gc_hap(c, perm[j]);
  if (j < ng) {
    const uint64_t r = perm[group_off[j]];  // the group's first member in table order
    rep.hap[j] = (uint32_t)r;               // the representative's row index in the full table
    rep.pos[j] = gc_pos(c, r);
    rep.strand[j] = (uint8_t)gc_strand(c, r);
    rep.start[j] = gc_start(c, r);
    rep.stop[j] = gc_stop(c, r);
    rep.flags[j] = (uint8_t)gc_flags(c, r);
    rep.cfdon[j] = gc_cfdon(c, r);
#pragma unroll
    for (int pl = 0; pl < HAWK_PLANES; ++pl) rep.win[(size_t)pl * rep.cap + j] = gc_win(c, pl, r);
  }
}
void hawk_launch_collapse_export(hipStream_t st, const GuideCols& c, uint64_t n, uint64_t ng, const uint32_t* perm,
                                 const uint64_t* group_off, const GuideCols& rep, uint32_t* member_hap) {
  const uint64_t m = n > ng ? n : ng;
  hipLaunchKernelGGL(k_collapse_export, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, c, n, ng, perm, group_off, rep, member_hap);
}

// ---------------------------------------------------------------------------------------------------------------
// Grouping through a hash table instead of a sort of all rows: when groups are far fewer than rows (a C4 tile: 1.1 x 10^8
// rows in 8.9 x 10^5 groups) the table (16 B per slot, a few MB) lives in the memory-side cache, every row finds its
// group with one or two probes, and what is sorted afterwards is (group number, row) on ~20 bits instead of the 48-bit
// row keys.  Same key, same identity test (63 hash bits; a 31-bit collision inside one (start, strand) is counted and
// answered with another seed) and the SAME output as the sort path: groups in key order, members in table order.
//   k_cg_insert  row -> key, identity; claim / find the key's slot (CAS), keep the smallest (identity, row) per slot
//   k_cg_occ + rocprim::exclusive_scan + k_cg_pack   occupied slots -> dense (key, slot) pairs, their number
//   rocprim::radix_sort_pairs (key, slot); k_cg_rank: slot -> position of its key = group number
//   k_cg_rowgid  row -> group number;  rocprim::radix_sort_pairs (group number, row);  k_cg_groups: CSR offsets + GC
struct CgSlot { unsigned long long key, idrow; };
#define CG_EMPTY 0xffffffffffffffffull
#define CG_MAXPROBE 64

__global__ __launch_bounds__(256) void k_cg_insert(GuideCols c, const uint8_t* __restrict__ is_ref, uint64_t n, int L, int up, int down,
                                                   int64_t base, uint64_t seed, CgSlot* __restrict__ T, uint32_t mask,
                                                   uint32_t* __restrict__ slot_of_row, unsigned long long* __restrict__ counters) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // the table has turned out too small (far more groups than expected): the call is going to the sort path anyway
  if (__hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4096ull) return;
  const RowKey k = row_key(c, is_ref, i, L, up, down);
  uint64_t h = seed;
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) h = mix64(h ^ k.core[pl]);
  h = mix64(h ^ (uint64_t)k.stop ^ ((uint64_t)(k.sr >> 1) << 63));
  const unsigned long long key = ((uint64_t)(k.start - base) << 32) | ((uint64_t)(k.sr & 1u) << 31) | (h >> 33);
  const unsigned long long idr = ((unsigned long long)(uint32_t)h << 32) | (uint32_t)i;
  uint32_t s = (uint32_t)mix64(key) & mask;
  for (int probe = 0; probe < CG_MAXPROBE; ++probe, s = (s + 1) & mask) {
    // One 16-byte read of the slot: the usual row finds its key and a smaller row of its identity there and is done.  A
    // plain load - what it returns may be stale, and every decision taken on it is either confirmed by an atomic (an
    // empty slot is claimed by CAS, a larger minimum replaced by atomicMin, both of which answer with the truth) or
    // stays right when the slot has moved on (keys never change; the minimum only falls; any earlier minimum is a row of
    // the same key to compare identities with).
    const ulonglong2 sl = *reinterpret_cast<const ulonglong2*>(&T[s]);
    unsigned long long cur = sl.x, old = sl.y;
    if (cur == CG_EMPTY) {
      const unsigned long long prev = atomicCAS(&T[s].key, CG_EMPTY, key);
      cur = prev == CG_EMPTY ? key : prev;
    }
    if (cur != key) continue;
    // the slot's (identity, row) minimum: rows arrive roughly in table order, so most leave without an atomic
    if (old == CG_EMPTY || old > idr) old = atomicMin(&T[s].idrow, idr);
    if (old != CG_EMPTY && (uint32_t)(old >> 32) != (uint32_t)(idr >> 32)) atomicAdd(&counters[0], 1ull);  // two identities, one key
    slot_of_row[i] = s;
    return;
  }
  atomicAdd(&counters[1], 1ull);  // table too full: the caller falls back to the sort
  slot_of_row[i] = 0;
}

__global__ __launch_bounds__(256) void k_cg_occ(const CgSlot* __restrict__ T, uint32_t C, uint32_t* __restrict__ flags) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s < C) flags[s] = T[s].key != CG_EMPTY ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_cg_pack(const CgSlot* __restrict__ T, uint32_t C, const uint32_t* __restrict__ flags,
                                                 const uint32_t* __restrict__ dense, uint64_t* __restrict__ gkey,
                                                 uint32_t* __restrict__ gslot, unsigned long long* __restrict__ counters) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= C) return;
  if (flags[s]) { gkey[dense[s]] = T[s].key; gslot[dense[s]] = s; }
  if (s == C - 1) counters[2] = (unsigned long long)dense[s] + flags[s];
}
__global__ __launch_bounds__(256) void k_cg_rank(const uint32_t* __restrict__ gslot_sorted, uint32_t G, uint32_t* __restrict__ slot2rank) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < G) slot2rank[gslot_sorted[p]] = p;
}
__global__ __launch_bounds__(256) void k_cg_rowgid(const uint32_t* __restrict__ slot_of_row, const uint32_t* __restrict__ slot2rank, uint64_t n,
                                                   uint32_t* __restrict__ gid, uint32_t* __restrict__ vals) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { gid[i] = slot2rank[slot_of_row[i]]; vals[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(256) void k_cg_groups(GuideCols c, uint64_t n, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ gid,
                                                   int guidelen, int pamlen, int right, uint64_t* __restrict__ group_off,
                                                   uint8_t* __restrict__ gc_num, uint8_t* __restrict__ gc_den) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t g = gid[j];
  if (j && gid[j - 1] == g) return;
  group_off[g] = j;
  const uint64_t r = vals[j];  // the group's first member in table order
  const bool pamfirst = (right != 0) != (gc_strand(c, r) != 0);
  const int sh = HAWK_PAD + (pamfirst ? pamlen : 0);
  const uint64_t m = guidelen >= 64 ? ~0ull : ((1ull << guidelen) - 1ull);
  const uint64_t A = (gc_win(c, 0, r) >> sh) & m, C = (gc_win(c, 1, r) >> sh) & m, G = (gc_win(c, 2, r) >> sh) & m,
                 T = (gc_win(c, 3, r) >> sh) & m;
  const uint64_t gc = (C | G) & ~(A | T), at = (A | T) & ~(C | G);
  gc_num[g] = (uint8_t)__popcll(gc);
  gc_den[g] = (uint8_t)__popcll(gc | at);
}

size_t hawk_collapse_hash_temp_bytes(uint64_t n, uint32_t C) {
  size_t a = 0, b = 0, d = 0;
  (void)rocprim::exclusive_scan(nullptr, a, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)C, rocprim::plus<uint32_t>(), (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)C, 0, 64,
                                  (hipStream_t)0);
  (void)rocprim::radix_sort_pairs(nullptr, d, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32,
                                  (hipStream_t)0);
  return std::max(a, std::max(b, d));
}

// stage 1: rows into the table, occupied slots packed; counters[0] = identity collisions, [1] = rows that found no
// slot, [2] = groups - read by the host before stage 2.  table: C slots (a power of two), flags / dense: C words each,
// gkey: 2 * C, gslot: 2 * C (sort ping-pong)
int hawk_launch_collapse_hash1(hipStream_t st, const GuideCols& c, const uint8_t* is_ref, uint64_t n, int guidelen, int pamlen, int flank_up,
                               int flank_down, int64_t base, uint64_t seed, void* temp, size_t temp_bytes, void* table, uint32_t C,
                               uint32_t* flags, uint32_t* dense, uint64_t* gkey, uint32_t* gslot, uint32_t* slot_of_row,
                               unsigned long long* counters) {
  const int L = guidelen + pamlen;
  if (hipMemsetAsync(table, 0xff, (size_t)C * sizeof(CgSlot), st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_cg_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, is_ref, n, L, flank_up, flank_down, base, seed,
                     (CgSlot*)table, C - 1, slot_of_row, counters);
  const dim3 cg((C + 255) / 256);
  hipLaunchKernelGGL(k_cg_occ, cg, dim3(256), 0, st, (const CgSlot*)table, C, flags);
  size_t tb = temp_bytes;
  if (rocprim::exclusive_scan(temp, tb, flags, dense, 0u, (size_t)C, rocprim::plus<uint32_t>(), st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_cg_pack, cg, dim3(256), 0, st, (const CgSlot*)table, C, flags, dense, gkey, gslot, counters);
  return 0;
}

// stage 2: groups in key order, rows by (group, row).  slot2rank: C words (may be `flags` of stage 1); gid: 2 * n, vals: 2 * n
int hawk_launch_collapse_hash2(hipStream_t st, const GuideCols& c, uint64_t n, uint32_t G, int guidelen, int pamlen, int right, unsigned key_end_bit,
                               void* temp, size_t temp_bytes, uint64_t* gkey, uint32_t* gslot, uint32_t C, uint32_t* slot2rank,
                               const uint32_t* slot_of_row, uint32_t* gid, uint32_t* vals, uint64_t* group_off, uint8_t* gc_num,
                               uint8_t* gc_den) {
  size_t tb = temp_bytes;
  if (rocprim::radix_sort_pairs(temp, tb, gkey, gkey + C, gslot, gslot + C, (size_t)G, 0, key_end_bit, st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_cg_rank, dim3((G + 255) / 256), dim3(256), 0, st, gslot + C, G, slot2rank);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(k_cg_rowgid, grid, dim3(256), 0, st, slot_of_row, slot2rank, n, gid, vals);
  unsigned gbits = 1;
  while (gbits < 32 && (G >> gbits) != 0) ++gbits;
  tb = temp_bytes;
  if (rocprim::radix_sort_pairs(temp, tb, gid, gid + n, vals, vals + n, (size_t)n, 0, gbits, st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_cg_groups, grid, dim3(256), 0, st, c, n, vals + n, gid + n, guidelen, pamlen, right, group_off, gc_num, gc_den);
  return 0;
}

// the tail of the grouping for a table whose rows already know their group number (collapse_by_templates): rows sorted by
// (group, row), CSR offsets and G/C counts from each group's first member.  gid / vals: 2 * n words each (sort ping-pong)
size_t hawk_collapse_expand_temp_bytes(uint64_t n) {
  size_t d = 0;
  (void)rocprim::radix_sort_pairs(nullptr, d, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32,
                                  (hipStream_t)0);
  return d;
}
int hawk_launch_collapse_expand(hipStream_t st, const GuideCols& c, uint64_t n, unsigned gbits, int guidelen, int pamlen, int right, void* temp,
                                size_t temp_bytes, uint32_t* gid, uint32_t* vals, uint64_t* group_off, uint8_t* gc_num, uint8_t* gc_den) {
  size_t tb = temp_bytes;
  if (rocprim::radix_sort_pairs(temp, tb, gid, gid + n, vals, vals + n, (size_t)n, 0, gbits, st) != hipSuccess) return -2;
  hipLaunchKernelGGL(k_cg_groups, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, n, vals + n, gid + n, guidelen, pamlen, right, group_off, gc_num,
                     gc_den);
  return 0;
}
