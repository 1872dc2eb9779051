// hawk_expand.hip — SURVEY.md §8 row f1: haplotype expansion on the device.
//
// Restates Haplotype.add_variants_phased (haplotype.py:214-252) for one chromosome copy per
// output row, without ever forming strings: from the REF region's planes, a position-sorted
// table of non-overlapping variants (SNV 1->1, deletion span->1, insertion 1->k) and, per row,
// the list of variants it carries with their output start positions (an exclusive prefix sum of
// the length changes, computed by the caller), every 32-base output word of every plane is
// assembled independently: stretches between variants are unaligned 32-bit copies of the REF
// planes (funnel shift), alt alleles are written from their IUPAC codes with the V bit set
// (the reference lower-cases every alt base, haplotype.py:120: a deletion's anchor is marked
// even though it equals REF).
//   once per plan (hawk_xplan_create):
//   k_hx_records one 32-byte record per carried variant of every row: where it starts in the row, where REF resumes
//               behind it, its first 32 alt bases as plane bits
//   k_hx_index  per (row, tile of 32768 output positions): the row's records the tile needs, the REF words it reads
//   every run (hawk_xplan_run):
//   k_hx_build  one thread per four consecutive output words; the tile's records and its image in the REF planes
//               are staged in LDS first (two independent coalesced loads), then no thread waits on global memory
//   k_hx_hash   128-bit position-tagged content hash per row (for collapse_haplotypes,
//               haplotypes.py:274-294, and the homozygous test 326-333, done by the caller)
#include "hawk_hx.h"

__global__ __launch_bounds__(256) void k_hx_records(const uint32_t* __restrict__ hv_idx, const int32_t* __restrict__ hv_o, uint64_t ncar,
                                                    const uint32_t* __restrict__ v_r0, const uint32_t* __restrict__ v_span,
                                                    const uint32_t* __restrict__ v_alt_off, const uint32_t* __restrict__ v_alt_len,
                                                    const uint4* __restrict__ v_am, HxVar* __restrict__ recs) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncar) return;
  const uint32_t vi = hv_idx[i];
  const uint4 m = v_am[vi];
  HxVar v;
  v.o = hv_o[i]; v.rs = v_r0[vi] + v_span[vi]; v.alt_len = v_alt_len[vi]; v.alt_off = v_alt_off[vi];
  v.m[0] = m.x; v.m[1] = m.y; v.m[2] = m.z; v.m[3] = m.w;
  recs[i] = v;
}

// Per (row, tile), before any plane is built: which of the row's carried variants the tile needs - the last one
// starting at or before the tile's first output position (its alt allele / the copy behind it may reach in) and every
// one starting inside its 32768 positions - and which REF words its copies read.  One thread per (row, tile): the
// binary searches are dependent global loads, cheap when 10^5 of them run side by side, but microseconds of serial
// latency when lane 0 of every k_hx_build workgroup does them with 255 lanes waiting (the first version: 2.4 ms
// instead of 0.9 ms).
__global__ __launch_bounds__(256) void k_hx_index(const uint64_t* __restrict__ hv_off, const HxVar* __restrict__ recs,
                                                  const uint32_t* __restrict__ hap_len, uint32_t n_hap, uint32_t wpr,
                                                  HxTile* __restrict__ tiles) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)n_hap * wpr) return;
  const uint32_t h = (uint32_t)(i / wpr), wb = (uint32_t)(i % wpr);
  const uint64_t lo = hv_off[h];
  const int K = (int)(hv_off[h + 1] - lo);
  const int32_t p_lo = (int32_t)(wb * HX_TW * 32u), p_hi = p_lo + HX_TW * 32;
  int a = 0, b = K;  // first index with o > p_lo
  while (a < b) { const int m = (a + b) >> 1; if (recs[lo + m].o <= p_lo) a = m + 1; else b = m; }
  int c = a, d = K;  // first index with o >= p_hi
  while (c < d) { const int m = (c + d) >> 1; if (recs[lo + m].o < p_hi) c = m + 1; else d = m; }
  const int first = a - 1 < 0 ? 0 : a - 1;
  const int want = c - first;
  const int n = want < 0 ? 0 : (want > HX_MAXV ? HX_MAXV : want);
  // REF position an output position maps to under variant k's stretch (k < 0: unmodified REF); inside the alt allele:
  // where the copy behind it starts.  Copies read REF in ascending order over the tile, so the first and the last
  // output position bound what is read.
  auto refpos = [&](int k, int32_t p) -> uint32_t {
    if (k < 0) return (uint32_t)p;
    const int32_t e = recs[lo + k].o + (int32_t)recs[lo + k].alt_len;
    return recs[lo + k].rs + (uint32_t)(p > e ? p - e : 0);
  };
  const int32_t len = (int32_t)hap_len[h];
  const int32_t plast = (p_hi < len ? p_hi : len) - 1;
  uint32_t ws = 0, flags = HX_FITS;
  if (plast >= p_lo) {
    ws = (refpos(a - 1, p_lo) >> 5) & ~3u;                 // 16-byte loads stage the window
    const uint32_t we = (refpos(c - 1, plast) >> 5) + 2;  // a 32-bit read at bit r touches words r >> 5 and (r >> 5) + 1
    if (we - ws > HX_RW) flags = 0;
  }
  if (n == want || want < 0) flags |= HX_ALL;
  if (a - 1 >= 0) flags |= HX_HEAD;
  const uint64_t f = lo + (uint64_t)first;
  tiles[i] = HxTile{(uint32_t)f, (uint32_t)(f >> 32), (uint32_t)n | flags, ws};
}

// Staged REF word i of a plane lives at (i & 3) * HX_RW / 4 + (i >> 2): a thread owns four consecutive output words, so
// the lanes of a wave read words 4 apart - a 4-way bank conflict in the natural layout, consecutive banks in this one.
__device__ __forceinline__ uint32_t hx_slot(uint32_t i) { return (i & 3u) * (HX_RW / 4) + (i >> 2); }

struct HxArgs {
  const uint32_t* ref[4];
  uint32_t ref_S;
  const HxVar* recs;
  const uint8_t* alt_codes;
};

// The four words of one thread (hx_words_t, hawk_hx.h).  FAST: every record the tile needs and its whole REF image are in
// LDS (the usual tile); otherwise records beyond the staged ones and / or the REF words come from global memory.
template <bool FAST>
__device__ __forceinline__ void hx_words(const HxArgs& g, const HxVar* __restrict__ s_v, const uint32_t (*__restrict__ s_ref)[HX_RW],
                                         const HxVar* __restrict__ first, int n, int j_end /* records of the row from `first` on */,
                                         bool head, bool staged, uint32_t ws, int32_t p0, int32_t len,
                                         uint32_t (&oA)[4], uint32_t (&oC)[4], uint32_t (&oG)[4], uint32_t (&oT)[4], uint32_t (&oV)[4]) {
  // 32 bits of REF plane pl from bit r.  A read under a mapping that a later variant of the word replaces, or of bits
  // past the end of the row, may point outside what exists: the word index is clamped, the bits are overwritten / masked.
  auto ref32 = [&](int pl, uint32_t r) -> uint32_t {
    if (FAST || staged) {
      uint32_t w = (r >> 5) - ws;
      w = w < HX_RW - 2 ? w : HX_RW - 2;
      return fsh(s_ref[pl][hx_slot(w)], s_ref[pl][hx_slot(w + 1)], r & 31u);
    }
    const uint32_t w = (r >> 5) < g.ref_S - 3 ? (r >> 5) : g.ref_S - 3;
    return ext_glb(g.ref[pl], (w << 5) | (r & 31u)).lo;
  };
  hx_words_t<FAST, 4>(g.alt_codes, s_v, first, n, j_end, head, p0, len, ref32, oA, oC, oG, oT, oV);
}

__global__ __launch_bounds__(HAWK_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_hx_build(HxArgs g, const uint64_t* __restrict__ hv_off, const uint32_t* __restrict__ hap_len,
                                                          uint32_t S, uint32_t wpr /*tiles per row*/, const HxTile* __restrict__ tiles,
                                                          uint32_t* pA, uint32_t* pC, uint32_t* pG, uint32_t* pT, uint32_t* pV) {
  __shared__ HxVar s_v[HX_MAXV];
  __shared__ uint32_t s_ref[4][HX_RW];
  const uint32_t h = blockIdx.x / wpr, wb = blockIdx.x % wpr;
  const uint32_t w0 = wb * HX_TW + threadIdx.x * 4u;
  const HxTile t = tiles[blockIdx.x];  // k_hx_index
  const int32_t len = (int32_t)hap_len[h];
  const int n = (int)(t.n_flags & 0xffffu);
  const bool staged = (t.n_flags & HX_FITS) != 0, all = (t.n_flags & HX_ALL) != 0, head = (t.n_flags & HX_HEAD) != 0;
  const uint32_t ws = t.ws;
  const HxVar* first = g.recs + (((uint64_t)t.first_hi << 32) | t.first_lo);
  // both stagings are independent coalesced loads, in flight together
  if (staged) {  // the REF words the tile's copies read: 16-byte loads, each word to its swizzled slot
    for (uint32_t i = threadIdx.x; i < HX_RW / 4; i += HAWK_BLOCK) {
      const uint32_t w = ws + 4 * i + 4 <= g.ref_S ? ws + 4 * i : g.ref_S - 4;  // rows are whole 16-byte quads
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        const uint4 q = *reinterpret_cast<const uint4*>(g.ref[pl] + w);
        s_ref[pl][i] = q.x; s_ref[pl][HX_RW / 4 + i] = q.y; s_ref[pl][2 * (HX_RW / 4) + i] = q.z; s_ref[pl][3 * (HX_RW / 4) + i] = q.w;
      }
    }
  }
  if ((int)threadIdx.x < n) s_v[threadIdx.x] = first[threadIdx.x];  // n <= HX_MAXV < workgroup size
  __syncthreads();
  if (w0 >= S) return;
  uint32_t oA[4] = {0, 0, 0, 0}, oC[4] = {0, 0, 0, 0}, oG[4] = {0, 0, 0, 0}, oT[4] = {0, 0, 0, 0}, oV[4] = {0, 0, 0, 0};
  const int32_t p0 = (int32_t)(w0 * 32u);
  if (p0 < len) {
    if (staged && all) {
      hx_words<true>(g, s_v, s_ref, first, n, n, head, true, ws, p0, len, oA, oC, oG, oT, oV);
    } else {  // more records than LDS holds start inside the tile (one every 200 nt over all of it), or a long REF image
      const int j_end = (int)(g.recs + hv_off[h + 1] - first);
      hx_words<false>(g, s_v, s_ref, first, n, j_end, head, staged, ws, p0, len, oA, oC, oG, oT, oV);
    }
  }
  const size_t o = (size_t)h * S + w0;  // S and w0 are multiples of 4: one 16-byte store per plane
  *reinterpret_cast<uint4*>(pA + o) = make_uint4(oA[0], oA[1], oA[2], oA[3]);
  *reinterpret_cast<uint4*>(pC + o) = make_uint4(oC[0], oC[1], oC[2], oC[3]);
  *reinterpret_cast<uint4*>(pG + o) = make_uint4(oG[0], oG[1], oG[2], oG[3]);
  *reinterpret_cast<uint4*>(pT + o) = make_uint4(oT[0], oT[1], oT[2], oT[3]);
  *reinterpret_cast<uint4*>(pV + o) = make_uint4(oV[0], oV[1], oV[2], oV[3]);
}

uint32_t hawk_hx_tiles_per_row(uint32_t S) { return (S + HX_TW - 1) / HX_TW; }
size_t hawk_hx_record_bytes() { return sizeof(HxVar); }
size_t hawk_hx_tile_bytes() { return sizeof(HxTile); }

// plan creation: records of all carried variants, then the per-tile index over them
void hawk_launch_hx_prepare(hipStream_t st, const uint64_t* hv_off, const uint32_t* hv_idx, const int32_t* hv_o, uint64_t ncar,
                            const uint32_t* v_r0, const uint32_t* v_span, const uint32_t* v_alt_off, const uint32_t* v_alt_len,
                            const void* v_am, const uint32_t* hap_len, uint32_t n_hap, uint32_t S, void* recs, void* tiles) {
  if (ncar)
    hipLaunchKernelGGL(k_hx_records, dim3((unsigned)((ncar + 255) / 256)), dim3(256), 0, st, hv_idx, hv_o, ncar, v_r0, v_span, v_alt_off,
                       v_alt_len, (const uint4*)v_am, (HxVar*)recs);
  const uint32_t wpr = hawk_hx_tiles_per_row(S);
  const uint64_t nwg = (uint64_t)n_hap * wpr;
  hipLaunchKernelGGL(k_hx_index, dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, st, hv_off, (const HxVar*)recs, hap_len, n_hap, wpr,
                     (HxTile*)tiles);
}

void hawk_launch_hx_build(hipStream_t st, const uint32_t* const* ref, uint32_t ref_S, const void* recs, const uint8_t* alt_codes,
                          const uint64_t* hv_off, const uint32_t* hap_len, uint32_t n_hap, uint32_t S, uint32_t* const* plane,
                          const void* tiles) {
  const uint32_t wpr = hawk_hx_tiles_per_row(S);
  HxArgs g;
  for (int p = 0; p < 4; ++p) g.ref[p] = ref[p];
  g.ref_S = ref_S; g.recs = (const HxVar*)recs; g.alt_codes = alt_codes;
  hipLaunchKernelGGL(k_hx_build, dim3(n_hap * wpr), dim3(HAWK_BLOCK), 0, st, g, hv_off, hap_len, S, wpr, (const HxTile*)tiles, plane[0],
                     plane[1], plane[2], plane[3], plane[4]);
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
__device__ __forceinline__ uint64_t rotl64(uint64_t k, int r) { return r ? (k << r) | (k >> (64 - r)) : k; }
// hash[2h], hash[2h+1]: two multilinear sums over the row's non-zero words, sum_(plane, word index) word x key(plane, index)
// mod 2^64, with keys mixed from the word index once and rotated per plane (zero words - row padding included - contribute
// nothing, so rows of different stride hash alike).  Only a FILTER: rows whose hashes agree are compared word for word
// (hawk_hapset_rows_equal) before they are treated as one haplotype.  Four words per plane and thread, 16-byte loads; the first
// version mixed every word twice through fmix64 and ran at 2.3 TB/s (1.36 ms on C3), VALU-bound.
__global__ __launch_bounds__(HAWK_BLOCK) void k_hx_hash(const uint32_t* pA, const uint32_t* pC, const uint32_t* pG, const uint32_t* pT,
                                                         const uint32_t* pV, uint32_t S, uint32_t wpr, unsigned long long* hash) {
  __shared__ unsigned long long s_h[2];
  const uint32_t h = blockIdx.x / wpr, wb = blockIdx.x % wpr;
  const uint32_t w = (wb * HAWK_BLOCK + threadIdx.x) * 4;
  if (threadIdx.x < 2) s_h[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long a = 0, b = 0;
  if (w < S) {  // S is a multiple of 4: the four words exist
    const size_t o = (size_t)h * S + w;
    const uint32_t* const pl[5] = {pA, pC, pG, pT, pV};
    uint4 x[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) x[p] = *reinterpret_cast<const uint4*>(pl[p] + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t k = fmix64((uint64_t)(w + j) + 0x9e3779b97f4a7c15ull) | 1ull, k2 = fmix64(k ^ 0xd6e8feb86659fd93ull) | 1ull;
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        const uint32_t xv = j == 0 ? x[p].x : j == 1 ? x[p].y : j == 2 ? x[p].z : x[p].w;
        a += (uint64_t)xv * rotl64(k, 11 * p);
        b += (uint64_t)xv * rotl64(k2, 7 * p + 3);
      }
    }
  }
  // wave sums first (64-bit as two 32-bit halves with carry-free 64-bit shuffles), then one LDS atomic per wave
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
  if ((threadIdx.x & (WAVE - 1)) == 0) { atomicAdd(&s_h[0], a); atomicAdd(&s_h[1], b); }
  __syncthreads();
  if (threadIdx.x == 0 && (s_h[0] | s_h[1])) { atomicAdd(&hash[2 * h], s_h[0]); atomicAdd(&hash[2 * h + 1], s_h[1]); }
}
void hawk_launch_hx_hash(hipStream_t st, uint32_t* const* plane, uint32_t n_hap, uint32_t S, unsigned long long* hash) {
  const uint32_t wpr = (S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;  // four words per thread
  hipLaunchKernelGGL(k_hx_hash, dim3(n_hap * wpr), dim3(HAWK_BLOCK), 0, st, plane[0], plane[1], plane[2], plane[3], plane[4], S, wpr, hash);
}
