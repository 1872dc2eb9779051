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
//   k_hx_build  one thread per output word; the row's variants overlapping the workgroup's 8192
//               positions are staged in LDS first
//   k_hx_hash   128-bit position-tagged content hash per row (for collapse_haplotypes,
//               haplotypes.py:274-294, and the homozygous test 326-333, done by the caller)
#include "hawk_bits.h"

#define HX_MAXV 192  // carried variants staged per workgroup (8192 output positions)

struct HxVar { int32_t o; uint32_t r0; uint32_t span; uint32_t alt_len; uint32_t alt_off; };

// Which of a row's carried variants a workgroup of k_hx_build needs: the last one starting at or before the
// workgroup's first output position (its alt allele / the copy behind it may reach in) and every one starting
// inside its 8192 positions.  One thread per (row, workgroup): the two binary searches are dependent global
// loads, cheap when 6 x 10^5 of them run side by side, but 5 us of serial latency when lane 0 of every
// k_hx_build workgroup does them with 255 lanes waiting (the first version: 2.4 ms instead of 0.9 ms).
__global__ __launch_bounds__(256) void k_hx_index(const uint64_t* __restrict__ hv_off, const int32_t* __restrict__ hv_o, uint32_t n_hap,
                                                  uint32_t wpr, int32_t* __restrict__ wg_k0, uint32_t* __restrict__ wg_n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)n_hap * wpr) return;
  const uint32_t h = (uint32_t)(i / wpr), wb = (uint32_t)(i % wpr);
  const uint64_t lo = hv_off[h];
  const int K = (int)(hv_off[h + 1] - lo);
  const int32_t p_lo = (int32_t)(wb * HAWK_BLOCK * 32u), p_hi = p_lo + HAWK_BLOCK * 32;
  int a = 0, b = K;  // first index with o > p_lo
  while (a < b) { const int m = (a + b) >> 1; if (hv_o[lo + m] <= p_lo) a = m + 1; else b = m; }
  int c = a, d = K;  // first index with o >= p_hi
  while (c < d) { const int m = (c + d) >> 1; if (hv_o[lo + m] < p_hi) c = m + 1; else d = m; }
  const int first = a - 1 < 0 ? 0 : a - 1;
  int n = c - first;
  wg_k0[i] = a - 1;
  wg_n[i] = (uint32_t)(n < 0 ? 0 : (n > HX_MAXV ? HX_MAXV : n));
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_hx_build(const uint32_t* __restrict__ refA, const uint32_t* __restrict__ refC,
                                                          const uint32_t* __restrict__ refG, const uint32_t* __restrict__ refT,
                                                          const uint32_t* __restrict__ v_r0, const uint32_t* __restrict__ v_span,
                                                          const uint32_t* __restrict__ v_alt_off, const uint32_t* __restrict__ v_alt_len,
                                                          const uint8_t* __restrict__ alt_codes, const uint64_t* __restrict__ hv_off,
                                                          const uint32_t* __restrict__ hv_idx, const int32_t* __restrict__ hv_o,
                                                          const uint32_t* __restrict__ hap_len, uint32_t S, uint32_t wpr /*workgroups per row*/,
                                                          const int32_t* __restrict__ wg_k0, const uint32_t* __restrict__ wg_n,
                                                          uint32_t* pA, uint32_t* pC, uint32_t* pG, uint32_t* pT, uint32_t* pV) {
  __shared__ HxVar s_v[HX_MAXV];
  const uint32_t h = blockIdx.x / wpr, wb = blockIdx.x % wpr;
  const uint32_t w = wb * HAWK_BLOCK + threadIdx.x;
  const uint64_t lo = hv_off[h], hi = hv_off[h + 1];
  const int K = (int)(hi - lo);
  const int32_t p_lo = (int32_t)(wb * HAWK_BLOCK * 32u), p_hi = p_lo + HAWK_BLOCK * 32;
  const int32_t len = (int32_t)hap_len[h];
  const int s_k0 = wg_k0[blockIdx.x];  // k_hx_index
  const int n = (int)wg_n[blockIdx.x];
  const int k0 = s_k0 < 0 ? 0 : s_k0;  // row-local index of the first staged variant
  for (int i = threadIdx.x; i < n; i += HAWK_BLOCK) {
    const uint32_t vi = hv_idx[lo + k0 + i];
    s_v[i] = HxVar{hv_o[lo + k0 + i], v_r0[vi], v_span[vi], v_alt_len[vi], v_alt_off[vi]};
  }
  __syncthreads();
  if (w >= S) return;
  // more carried variants start inside this workgroup's 8192 positions than LDS holds (> HX_MAXV, i.e. one every
  // ~40 nt): the ones beyond the staged range are read from global memory (workgroup-uniform flag, rare)
  const bool overflow = k0 + n < K && hv_o[lo + k0 + n] < p_hi;
  // variant kk of the row (0 <= kk < K): staged copy when there is one
  auto getv = [&](int kk) -> HxVar {
    if (kk >= k0 && kk < k0 + n) return s_v[kk - k0];
    const uint32_t vi = hv_idx[lo + kk];
    return HxVar{hv_o[lo + kk], v_r0[vi], v_span[vi], v_alt_len[vi], v_alt_off[vi]};
  };
  auto geto = [&](int kk) -> int32_t { return kk >= k0 && kk < k0 + n ? s_v[kk - k0].o : hv_o[lo + kk]; };
  uint32_t oA = 0, oC = 0, oG = 0, oT = 0, oV = 0;
  const int32_t p0 = (int32_t)(w * 32u);
  if (p0 < len) {
    const int32_t pend = p0 + 32 < len ? p0 + 32 : len;
    // k: row-local index of the last carried variant with o <= p0 (-1: none, the word starts in unmodified REF).
    // Variants before k0 start before the workgroup's first position and before variant k0 (or there is none <= p_lo),
    // so the search range is [k0, k0 + n), extended to the rest of the row in overflow mode.
    int a = k0, b = overflow ? K : k0 + n;
    while (a < b) { const int m = (a + b) >> 1; if (geto(m) <= p0) a = m + 1; else b = m; }
    int k = a - 1;
    int32_t cur = p0;
    while (cur < pend) {  // every iteration advances cur or k; k only while variants start inside the word
      const bool havev = k >= 0;
      HxVar v = HxVar{0, 0, 0, 0, 0};
      if (havev) v = getv(k);
      const int32_t next_o = k + 1 < K ? geto(k + 1) : 0x7fffffff;  // output start of the following variant
      if (havev && cur < v.o + (int32_t)v.alt_len) {
        // alt allele bases [cur - o, ...)
        const int32_t e = v.o + (int32_t)v.alt_len < pend ? v.o + (int32_t)v.alt_len : pend;
        for (int32_t q = cur; q < e; ++q) {
          const uint32_t c = alt_codes[v.alt_off + (uint32_t)(q - v.o)], bit = 1u << (q - p0);
          if (c & 1u) oA |= bit; if (c & 2u) oC |= bit; if (c & 4u) oG |= bit; if (c & 8u) oT |= bit;
          oV |= bit;
        }
        cur = e;
      } else {
        // copy REF: r = position in REF of output position cur
        const uint32_t r = havev ? v.r0 + v.span + (uint32_t)(cur - (v.o + (int32_t)v.alt_len)) : (uint32_t)cur;
        const int32_t e = pend < next_o ? pend : next_o;
        const int nb = e - cur;
        if (nb > 0) {
          const uint32_t m = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);
          const int sh = cur - p0;
          oA |= (ext_glb(refA, r).lo & m) << sh; oC |= (ext_glb(refC, r).lo & m) << sh;
          oG |= (ext_glb(refG, r).lo & m) << sh; oT |= (ext_glb(refT, r).lo & m) << sh;
          cur = e;
        }
      }
      if (cur >= next_o) ++k;  // the next variant starts here
    }
  }
  const size_t o = (size_t)h * S + w;
  pA[o] = oA; pC[o] = oC; pG[o] = oG; pT[o] = oT; pV[o] = oV;
}

void hawk_launch_hx_index(hipStream_t st, const uint64_t* hv_off, const int32_t* hv_o, uint32_t n_hap, uint32_t S, int32_t* wg_k0,
                          uint32_t* wg_n) {
  const uint32_t wpr = (S + HAWK_BLOCK - 1) / HAWK_BLOCK;
  const uint64_t nwg = (uint64_t)n_hap * wpr;
  hipLaunchKernelGGL(k_hx_index, dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, st, hv_off, hv_o, n_hap, wpr, wg_k0, wg_n);
}

void hawk_launch_hx_build(hipStream_t st, const uint32_t* const* ref, const uint32_t* v_r0, const uint32_t* v_span,
                          const uint32_t* v_alt_off, const uint32_t* v_alt_len, const uint8_t* alt_codes, const uint64_t* hv_off,
                          const uint32_t* hv_idx, const int32_t* hv_o, const uint32_t* hap_len, uint32_t n_hap, uint32_t S,
                          uint32_t* const* plane, int32_t* wg_k0, uint32_t* wg_n) {
  const uint32_t wpr = (S + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hipLaunchKernelGGL(k_hx_build, dim3(n_hap * wpr), dim3(HAWK_BLOCK), 0, st, ref[0], ref[1], ref[2], ref[3], v_r0, v_span, v_alt_off,
                     v_alt_len, alt_codes, hv_off, hv_idx, hv_o, hap_len, S, wpr, wg_k0, wg_n, plane[0], plane[1], plane[2], plane[3],
                     plane[4]);
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
// hash[2h], hash[2h+1]: sums of two differently seeded mixes of (plane, word index, word) over the row
__global__ __launch_bounds__(HAWK_BLOCK) void k_hx_hash(const uint32_t* pA, const uint32_t* pC, const uint32_t* pG, const uint32_t* pT,
                                                         const uint32_t* pV, uint32_t S, uint32_t wpr, unsigned long long* hash) {
  __shared__ unsigned long long s_h[2];
  const uint32_t h = blockIdx.x / wpr, wb = blockIdx.x % wpr;
  const uint32_t w = wb * HAWK_BLOCK + threadIdx.x;
  if (threadIdx.x < 2) s_h[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long a = 0, b = 0;
  if (w < S) {
    const size_t o = (size_t)h * S + w;
    const uint32_t x[5] = {pA[o], pC[o], pG[o], pT[o], pV[o]};
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      if (!x[p]) continue;  // zero words (incl. row padding) contribute nothing: rows of different stride hash alike
      const uint64_t key = ((uint64_t)(p + 1) << 56) | ((uint64_t)w << 32) | x[p];
      a += fmix64(key ^ 0x9e3779b97f4a7c15ull);
      b += fmix64(key * 0xd6e8feb86659fd93ull + 0x2545f4914f6cdd1dull);
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
  const uint32_t wpr = (S + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hipLaunchKernelGGL(k_hx_hash, dim3(n_hap * wpr), dim3(HAWK_BLOCK), 0, st, plane[0], plane[1], plane[2], plane[3], plane[4], S, wpr, hash);
}
