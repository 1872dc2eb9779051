// hawk_kernels.hip — hand-written gfx950 (CDNA4) kernels for CRISPR-HAWK's guide search.
//
// Everything here is integer / bitwise work on bit-sliced sequence planes; there is no GEMM
// shape anywhere on this path, so no MFMA: the kernels are HBM-streaming (K1, K2/K3) or
// latency-tolerant per-row gathers (K3b, K4).  Wavefront = 64 lanes is assumed.
//
//   k_pack      K1  cased ASCII  -> planes A,C,G,T,V  (wave ballots: one v_cmp per plane per
//                   64 bases, encoder.py:18-57 + the lower-case-as-data rule of haplotype.py:120)
//   k_scan      K2  PAM bitmask scan, both strands (search_guides.py:32-46, 87-99), fused with
//               K3a the in-range test (395-420) and the REF-identical window filter (468-471)
//   k_offsets       exclusive scan of the per-workgroup survivor counts
//   k_emit_*    K3b deterministic compaction of surviving bits into records / hit lists
//   k_guides    K3c coordinates (260-303), alt==REF redundancy (340-369), window gather
//               (134-160) and K4 CFDon (cfdscore.py:53-95 after annotation.py:27-51)
//   k_compact_* drop rows marked redundant
#include "hawk_device.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
// bits [s, s+32) of the 64-bit value {hi,lo}, 0 <= s < 32 : one v_alignbit_b32
__device__ __forceinline__ uint32_t fsh(uint32_t lo, uint32_t hi, uint32_t s) {
  return __builtin_amdgcn_alignbit(hi, lo, s);
}
// bits [s, s+32) of the 6-word little-endian bit string a[], starting at word k, 0 <= s < 64
__device__ __forceinline__ uint32_t shifted(const uint32_t (&a)[6], int k, int s) {
  return s < 32 ? fsh(a[k], a[k + 1], (uint32_t)s) : fsh(a[k + 1], a[k + 2], (uint32_t)(s - 32));
}
// mask of bit positions j (0..31) with lo <= base + j < hi
__device__ __forceinline__ uint32_t range_mask(int base, int lo, int hi) {
  int a = lo - base, b = hi - base;
  a = a < 0 ? 0 : a;
  b = b > 32 ? 32 : b;
  if (b <= a) return 0u;
  uint32_t m = b == 32 ? 0xffffffffu : ((1u << b) - 1u);
  return m & ~((1u << a) - 1u);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// exclusive scan over a workgroup of NW wavefronts; *total gets the workgroup sum
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  uint32_t inc = wave_incl_scan(v);
  if (lane == WAVE - 1) s_w[wv] = inc;
  __syncthreads();
  uint32_t pre = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    uint32_t x = s_w[i];
    if (i < wv) pre += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return pre + inc - v;
}

// ---------------------------------------------------------------------------------------
// K1: pack
// ---------------------------------------------------------------------------------------
// One wavefront packs 2048 consecutive bases of one haplotype: 32 rounds, in each round lane l
// holds base r*64+l, and five wave ballots give the 64 bits of the five planes at once.
// LUT value: bits 0-3 IUPAC mask, bit 4 lower case, bit 5 invalid.
__global__ __launch_bounds__(HAWK_BLOCK) void k_pack(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ seq_off,
                                                      uint32_t hap0, uint32_t n_hap_batch, uint64_t batch_base,
                                                      const uint32_t* __restrict__ hap_len, uint32_t S, uint32_t* pA,
                                                      uint32_t* pC, uint32_t* pG, uint32_t* pT, uint32_t* pV,
                                                      unsigned long long* bad_index) {
  __shared__ uint8_t lut[256];
  {
    const int c = threadIdx.x;  // HAWK_BLOCK == 256
    int u = c & 0xDF;           // ASCII upper-case fold for letters
    uint8_t m = 0x20;
    switch (u) {
      case 'A': m = 1; break;  case 'C': m = 2; break;  case 'G': m = 4; break;  case 'T': m = 8; break;
      case 'N': m = 15; break; case 'R': m = 5; break;  case 'Y': m = 10; break; case 'S': m = 6; break;
      case 'W': m = 9; break;  case 'K': m = 12; break; case 'M': m = 3; break;  case 'B': m = 14; break;
      case 'D': m = 13; break; case 'H': m = 11; break; case 'V': m = 7; break;
      default: break;
    }
    const bool letter = (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
    if (!letter) m = 0x20;
    if (m != 0x20 && c >= 'a') m |= 0x10;
    lut[c] = m;
  }
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t cph = (S + 63) / 64;  // 2048-base chunks per haplotype row
  const uint64_t gw = (uint64_t)blockIdx.x * (HAWK_BLOCK / WAVE) + threadIdx.x / WAVE;
  if (gw >= (uint64_t)cph * n_hap_batch) return;  // whole wave exits together
  const uint32_t hb = (uint32_t)(gw / cph), c = (uint32_t)(gw % cph);
  const uint32_t h = hap0 + hb;
  const uint32_t len = hap_len[h];
  const uint64_t src0 = seq_off[h] - batch_base;  // offset inside the staged batch
  uint32_t a = 0, cc = 0, g = 0, t = 0, v = 0;
  for (int r = 0; r < 32; ++r) {
    const uint32_t idx = c * 2048u + (uint32_t)r * 64u + (uint32_t)lane;
    uint32_t m = 0;
    if (idx < len) m = lut[ascii[src0 + idx]];
    const unsigned long long bad = __ballot(m & 0x20);
    if (bad) {
      if (lane == 0) atomicMin(bad_index, (unsigned long long)(seq_off[h] + c * 2048u + r * 64u + __builtin_ctzll(bad)));
    }
    const unsigned long long bA = __ballot(m & 1), bC = __ballot(m & 2), bG = __ballot(m & 4), bT = __ballot(m & 8),
                             bV = __ballot(m & 16);
    if (lane == 2 * r) {
      a = (uint32_t)bA; cc = (uint32_t)bC; g = (uint32_t)bG; t = (uint32_t)bT; v = (uint32_t)bV;
    } else if (lane == 2 * r + 1) {
      a = (uint32_t)(bA >> 32); cc = (uint32_t)(bC >> 32); g = (uint32_t)(bG >> 32); t = (uint32_t)(bT >> 32);
      v = (uint32_t)(bV >> 32);
    }
  }
  const uint32_t w = c * 64u + (uint32_t)lane;
  if (w < S) {
    const size_t o = (size_t)h * S + w;
    pA[o] = a; pC[o] = cc; pG[o] = g; pT[o] = t; pV[o] = v;
  }
}

void hawk_launch_pack(hipStream_t st, const uint8_t* ascii, const uint64_t* seq_off, uint32_t hap0, uint32_t n_hap_batch,
                      uint64_t batch_base, const uint32_t* hap_len, uint32_t S, uint32_t* const* plane,
                      unsigned long long* bad_index) {
  const uint64_t waves = (uint64_t)((S + 63) / 64) * n_hap_batch;
  const uint32_t grid = (uint32_t)((waves + 3) / 4);
  if (!grid) return;
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(HAWK_BLOCK), 0, st, ascii, seq_off, hap0, n_hap_batch, batch_base, hap_len, S,
                     plane[0], plane[1], plane[2], plane[3], plane[4], bad_index);
}

// ---------------------------------------------------------------------------------------
// K2 + K3a: PAM scan fused with the candidate filters
// ---------------------------------------------------------------------------------------
// Load this thread's 4 words of one plane plus 2 look-ahead words (from the next lane's
// registers; the last lane of a wave reads them from memory).
__device__ __forceinline__ void load6(const uint32_t* __restrict__ row, uint32_t u, uint32_t S, bool active,
                                      uint32_t (&a)[6]) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (active) v = *reinterpret_cast<const uint4*>(row + 4 * (size_t)u);
  a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
  uint32_t nx = __shfl_down(v.x, 1), ny = __shfl_down(v.y, 1);
  if ((threadIdx.x & (WAVE - 1)) == WAVE - 1) {
    nx = 0; ny = 0;
    if (active && 4 * u + 4 < S) { nx = row[4 * (size_t)u + 4]; ny = row[4 * (size_t)u + 5]; }
  }
  a[4] = nx; a[5] = ny;
}

// m[k] bit j  <=>  for every PAM position i: (pam[i] & base[32*(4u+k) + j + po + i]) != 0
// (search_guides.py:32-46: set intersection per nibble; an N nibble matches any real base).
__device__ __forceinline__ void pam_match(const uint32_t (&A)[6], const uint32_t (&C)[6], const uint32_t (&G)[6],
                                          const uint32_t (&T)[6], uint64_t pam, int pamlen, int po, uint32_t (&m)[4]) {
  m[0] = m[1] = m[2] = m[3] = 0xffffffffu;
  for (int i = 0; i < pamlen; ++i) {
    const uint32_t nib = (uint32_t)(pam >> (4 * (pamlen - 1 - i))) & 15u;
    if (nib == 15u) continue;  // wave-uniform: pam is a kernel argument
    uint32_t sel[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      uint32_t s = 0;
      if (nib & 1u) s |= A[k];
      if (nib & 2u) s |= C[k];
      if (nib & 4u) s |= G[k];
      if (nib & 8u) s |= T[k];
      sel[k] = s;
    }
    const int s = po + i;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] &= shifted(sel, k, s);
  }
}

// v[k] bit j := OR of the original bits [32k+j, 32k+j+L), valid for k < 4 when L <= 64
__device__ __forceinline__ void window_or(uint32_t (&v)[6], int L) {
  int r = 1;
  while (2 * r <= L && r < 32) {
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] |= fsh(v[k], v[k + 1], (uint32_t)r);
    v[5] |= v[5] >> r;
    r *= 2;
  }
  const int rem = L - r;
  if (rem > 0) {
    if (rem < 32) {
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] |= fsh(v[k], v[k + 1], (uint32_t)rem);
    } else {
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] |= v[k + 1];
    }
  }
}

// MODE 0: raw pam_search(): bit q of keepF/keepR = PAM (fwd / reverse-complement) matches at
//         relative position q in [scan_start, scan_stop).
// MODE 1: candidates indexed by WINDOW START q (leftmost base of spacer+PAM on the + strand):
//         for strand s the stored orientation is `right ^ s` (search_guides.py:538); a PAM-first
//         window has its PAM at q, a spacer-first window at q+guidelen.  A bit survives if the
//         PAM matches, the PAM position is inside the scan range, the 10-nt padded window fits
//         the haplotype (is_pamhit_in_range) and - for non-REF haplotypes - at least one variant
//         bit lies inside [q, q+L) (the `isupper()` skip of retrieve_guides).
template <int MODE>
__global__ __launch_bounds__(HAWK_BLOCK) void k_scan(HapSetDev hs, ScanParams p, uint32_t* __restrict__ keepF,
                                                      uint32_t* __restrict__ keepR, uint32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ aux) {
  __shared__ uint32_t s_acc[4];
  const uint32_t h = blockIdx.x / p.bph, blk = blockIdx.x % p.bph;
  const uint32_t u = blk * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
  const int haplen = (int)hs.hap_len[h];
  const int ss = hs.scan_start[h], se = hs.scan_stop[h];
  const bool isref = hs.is_ref[h] != 0;
  if (threadIdx.x < 4) s_acc[threadIdx.x] = 0;

  uint32_t A[6] = {0, 0, 0, 0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0}, G[6] = {0, 0, 0, 0, 0, 0}, T[6] = {0, 0, 0, 0, 0, 0};
  if (p.need & 1u) load6(hs.plane[0] + rowbase, u, hs.S, active, A);
  if (p.need & 2u) load6(hs.plane[1] + rowbase, u, hs.S, active, C);
  if (p.need & 4u) load6(hs.plane[2] + rowbase, u, hs.S, active, G);
  if (p.need & 8u) load6(hs.plane[3] + rowbase, u, hs.S, active, T);

  // orientation of the stored window per strand: PAM first?
  const bool pamfirstF = MODE == 0 ? true : (p.right != 0);
  const bool pamfirstR = MODE == 0 ? true : (p.right == 0);
  const int poF = pamfirstF ? 0 : p.guidelen, poR = pamfirstR ? 0 : p.guidelen;
  uint32_t mF[4], mR[4];
  pam_match(A, C, G, T, p.pam_fwd, p.pamlen, poF, mF);
  pam_match(A, C, G, T, p.pam_rev, p.pamlen, poR, mR);

  // scan range / in-range limits expressed on q
  int sloF = ss - poF, shiF = se - poF, sloR = ss - poR, shiR = se - poR;
  int loF = sloF, hiF = shiF, loR = sloR, hiR = shiR;
  if (MODE == 1) {
    const int qmin = HAWK_PAD, qmax = haplen - p.L - HAWK_PAD + 1;  // exclusive
    loF = loF > qmin ? loF : qmin; hiF = hiF < qmax ? hiF : qmax;
    loR = loR > qmin ? loR : qmin; hiR = hiR < qmax ? hiR : qmax;
  }
  uint32_t E[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0};
  if (MODE == 1 && !isref) {  // workgroup-uniform
    load6(hs.plane[4] + rowbase, u, hs.S, active, E);
    window_or(E, p.L);
  }
  const int base0 = (int)(u * 128u);
  const bool interior = base0 >= loF && base0 >= loR && base0 >= sloF && base0 >= sloR && base0 + 128 <= hiF &&
                        base0 + 128 <= hiR && base0 + 128 <= shiF && base0 + 128 <= shiR;
  uint32_t kF[4], kR[4];
  uint32_t cF = 0, cR = 0, cand = 0, hits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t f = mF[k], r = mR[k];
    if (!interior) {
      const int b = base0 + 32 * k;
      const uint32_t hf = f & range_mask(b, sloF, shiF), hr = r & range_mask(b, sloR, shiR);
      hits += __popc(hf) + __popc(hr);
      f &= range_mask(b, loF, hiF);
      r &= range_mask(b, loR, hiR);
      if (MODE == 1) { f &= hf; r &= hr; }
    } else {
      hits += __popc(f) + __popc(r);
    }
    cand += __popc(f) + __popc(r);
    if (MODE == 1) { f &= E[k]; r &= E[k]; }
    kF[k] = f; kR[k] = r;
    cF += __popc(f); cR += __popc(r);
  }
  if (active) {
    *reinterpret_cast<uint4*>(keepF + rowbase + 4 * (size_t)u) = make_uint4(kF[0], kF[1], kF[2], kF[3]);
    *reinterpret_cast<uint4*>(keepR + rowbase + 4 * (size_t)u) = make_uint4(kR[0], kR[1], kR[2], kR[3]);
  } else {
    cF = cR = cand = hits = 0;
  }
  // per-thread counts are <= 256, per-wave sums <= 16384: two 16-bit fields per word are safe
  const uint32_t w0 = wave_sum(cF | (cR << 16)), w1 = wave_sum(cand | (hits << 16));
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    atomicAdd(&s_acc[0], w0 & 0xffffu); atomicAdd(&s_acc[1], w0 >> 16);
    atomicAdd(&s_acc[2], w1 & 0xffffu); atomicAdd(&s_acc[3], w1 >> 16);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // MODE 1 order: (haplotype, strand, workgroup) = the reference's emission order;
    // MODE 0 order: (strand, haplotype, workgroup) so forward and reverse lists are separate.
    const size_t iF = MODE == 1 ? ((size_t)h * 2 + 0) * p.bph + blk : ((size_t)h) * p.bph + blk;
    const size_t iR = MODE == 1 ? ((size_t)h * 2 + 1) * p.bph + blk : ((size_t)hs.n_hap + h) * p.bph + blk;
    counts[iF] = s_acc[0]; counts[iR] = s_acc[1];
    aux[(size_t)h * p.bph + blk] = s_acc[2];
    aux[(size_t)(hs.n_hap + h) * p.bph + blk] = s_acc[3];
  }
}

void hawk_launch_scan(hipStream_t st, int mode, const HapSetDev& hs, const ScanParams& p, uint32_t* keepF, uint32_t* keepR,
                      uint32_t* counts, uint32_t* aux) {
  const dim3 grid(hs.n_hap * p.bph), block(HAWK_BLOCK);
  if (mode == 0) hipLaunchKernelGGL(k_scan<0>, grid, block, 0, st, hs, p, keepF, keepR, counts, aux);
  else hipLaunchKernelGGL(k_scan<1>, grid, block, 0, st, hs, p, keepF, keepR, counts, aux);
}

// ---------------------------------------------------------------------------------------
// exclusive scan of the workgroup counts (single workgroup; n is a few 10^5)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_offsets(const uint32_t* __restrict__ counts, uint64_t* __restrict__ offsets,
                                                   uint64_t n, uint64_t n_first_half, const uint32_t* __restrict__ aux,
                                                   uint64_t n_aux, ScanTotals* totals) {
  __shared__ unsigned long long s_part[1024];
  __shared__ unsigned long long s_aux[2];
  const uint32_t t = threadIdx.x;
  const uint64_t chunk = (n + 1023) / 1024;
  const uint64_t lo = (uint64_t)t * chunk < n ? (uint64_t)t * chunk : n;
  const uint64_t hi = lo + chunk < n ? lo + chunk : n;
  unsigned long long s = 0;
  for (uint64_t i = lo; i < hi; ++i) s += counts[i];
  s_part[t] = s;
  if (t < 2) s_aux[t] = 0;
  __syncthreads();
  // Hillis-Steele over 1024 partials
  for (int d = 1; d < 1024; d <<= 1) {
    unsigned long long x = t >= (uint32_t)d ? s_part[t - d] : 0;
    __syncthreads();
    s_part[t] += x;
    __syncthreads();
  }
  unsigned long long run = s_part[t] - s;
  for (uint64_t i = lo; i < hi; ++i) {
    offsets[i] = run;
    run += counts[i];
  }
  unsigned long long c = 0, hh = 0;
  for (uint64_t i = t; i < n_aux; i += 1024) { c += aux[i]; hh += aux[n_aux + i]; }
  atomicAdd(&s_aux[0], c);
  atomicAdd(&s_aux[1], hh);
  __syncthreads();
  if (t == 1023) {
    offsets[n] = s_part[1023];
    totals->n_keep = s_part[1023];
    totals->n_cand = s_aux[0];
    totals->n_hits = s_aux[1];
  }
  // prefix at the half boundary (raw mode: number of forward hits)
  if (lo <= n_first_half && n_first_half < hi) {
    unsigned long long r2 = s_part[t] - s;
    for (uint64_t i = lo; i < n_first_half; ++i) r2 += counts[i];
    totals->n_keep_fwd = r2;
  }
  if (n_first_half >= n && t == 0) totals->n_keep_fwd = s_part[1023];
}

void hawk_launch_offsets(hipStream_t st, const uint32_t* counts, uint64_t* offsets, uint64_t n, uint64_t n_first_half,
                         const uint32_t* aux, uint64_t n_aux, ScanTotals* totals) {
  hipLaunchKernelGGL(k_offsets, dim3(1), dim3(1024), 0, st, counts, offsets, n, n_first_half, aux, n_aux, totals);
}

// ---------------------------------------------------------------------------------------
// K3b: deterministic emission of surviving bits
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(HAWK_BLOCK) void k_emit(HapSetDev hs, uint32_t bph, const uint32_t* __restrict__ keepF,
                                                      const uint32_t* __restrict__ keepR, const uint64_t* __restrict__ offsets,
                                                      uint64_t n_fwd_total, uint64_t* __restrict__ rec,
                                                      uint32_t* __restrict__ hits_fwd, uint32_t* __restrict__ hits_rev) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  const uint32_t h = blockIdx.x / bph, blk = blockIdx.x % bph;
  const uint32_t u = blk * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint32_t* keep = s ? keepR : keepF;
    uint4 kw = make_uint4(0, 0, 0, 0);
    if (active) kw = *reinterpret_cast<const uint4*>(keep + rowbase + 4 * (size_t)u);
    const uint32_t c = __popc(kw.x) + __popc(kw.y) + __popc(kw.z) + __popc(kw.w);
    uint32_t tot;
    const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(c, s_w, &tot);
    if (tot == 0) continue;  // workgroup-uniform
    const size_t ci = MODE == 1 ? ((size_t)h * 2 + s) * bph + blk : ((size_t)s * hs.n_hap + h) * bph + blk;
    uint64_t o = offsets[ci] + ex;
    const uint32_t w[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k];
      while (x) {
        const uint32_t j = (uint32_t)__builtin_ctz(x);
        x &= x - 1;
        const uint32_t q = (4 * u + k) * 32 + j;
        if (MODE == 1) rec[o] = ((uint64_t)h << 32) | ((uint64_t)s << 31) | q;
        else if (s == 0) hits_fwd[o] = q;
        else hits_rev[o - n_fwd_total] = q;
        ++o;
      }
    }
  }
}

void hawk_launch_emit_records(hipStream_t st, const HapSetDev& hs, uint32_t bph, const uint32_t* keepF, const uint32_t* keepR,
                              const uint64_t* offsets, uint64_t* rec) {
  hipLaunchKernelGGL(k_emit<1>, dim3(hs.n_hap * bph), dim3(HAWK_BLOCK), 0, st, hs, bph, keepF, keepR, offsets, 0ull, rec,
                     (uint32_t*)nullptr, (uint32_t*)nullptr);
}
void hawk_launch_emit_hits(hipStream_t st, const HapSetDev& hs, uint32_t bph, const uint32_t* keepF, const uint32_t* keepR,
                           const uint64_t* offsets, uint64_t n_fwd_total, uint32_t* hits_fwd, uint32_t* hits_rev) {
  hipLaunchKernelGGL(k_emit<0>, dim3(hs.n_hap * bph), dim3(HAWK_BLOCK), 0, st, hs, bph, keepF, keepR, offsets, n_fwd_total,
                     (uint64_t*)nullptr, hits_fwd, hits_rev);
}

// ---------------------------------------------------------------------------------------
// K3c + K4: one thread per surviving candidate
// ---------------------------------------------------------------------------------------
// nbits (<= 64) bits of a plane row starting at bit position bp
__device__ __forceinline__ uint64_t extract_bits(const uint32_t* __restrict__ row, uint32_t bp, int nbits) {
  const uint32_t w = bp >> 5, sh = bp & 31u;
  const uint64_t lo = (uint64_t)row[w] | ((uint64_t)row[w + 1] << 32);
  uint64_t v = lo >> sh;
  if (sh) v |= (uint64_t)row[w + 2] << (64 - sh);
  if (nbits < 64) v &= (1ull << nbits) - 1ull;
  return v;
}
// haplotype position map (haplotype.py:90-159) as unit-slope segments
__device__ __forceinline__ int64_t posmap(const HapSetDev& hs, uint32_t h, uint32_t rel) {
  uint32_t lo = hs.seg_off[h], hi = hs.seg_off[h + 1];  // last k in [lo,hi) with seg_rel[k] <= rel
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (hs.seg_rel[mid] <= rel) lo = mid; else hi = mid;
  }
  return hs.seg_gen[lo] + (int64_t)(rel - hs.seg_rel[lo]);
}
__device__ __forceinline__ int base_index(uint32_t code) {  // A,C,G,T -> 0..3, anything else -1
  return (code & (code - 1u)) ? -1 : (code ? __builtin_ctz(code) : -1);
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_guides(HapSetDev hs, GuideParams gp, const uint32_t* __restrict__ keepF,
                                                        const uint32_t* __restrict__ keepR, const uint64_t* __restrict__ rec,
                                                        uint64_t n_rec, GuideCols out, uint8_t* __restrict__ valid,
                                                        unsigned long long* n_invalid, int* status) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= n_rec) return;
  const uint64_t r = rec[i];
  const uint32_t h = (uint32_t)(r >> 32), s = (uint32_t)(r >> 31) & 1u, q = (uint32_t)r & 0x7fffffffu;
  const int L = gp.L;
  const bool pamfirst = (gp.right != 0) != (s != 0);  // `right` as stored by search() for this strand
  const uint32_t pos = pamfirst ? q : q + (uint32_t)gp.guidelen;
  // search_guides.py:260-280: both orientations reduce to posmap[q], posmap[q+L]
  const int64_t start = posmap(hs, h, q), stop = posmap(hs, h, q + (uint32_t)L);
  const size_t rowbase = (size_t)h * hs.S;
  uint64_t core[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) core[p] = extract_bits(hs.plane[p] + rowbase, q, L);

  // remove_redundant_guides (search_guides.py:340-369): is there a REF guide at (start, strand)?
  bool has_ref = false, keep = true;
  uint64_t rcore[4] = {core[0], core[1], core[2], core[3]};
  if (hs.ref_index >= 0) {
    const uint32_t hr = (uint32_t)hs.ref_index;
    if (h == hr) {
      has_ref = true;
    } else {
      const int64_t qr = start - hs.seg_gen[hs.seg_off[hr]];  // REF position map is the identity + startp
      if (qr >= 0 && qr < (int64_t)hs.S * 32) {
        const uint32_t* kr = (s ? keepR : keepF) + (size_t)hr * hs.S;
        if ((kr[qr >> 5] >> (qr & 31)) & 1u) {
          has_ref = true;
          bool same = true;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            rcore[p] = extract_bits(hs.plane[p] + (size_t)hr * hs.S, (uint32_t)qr, L);
            same = same && rcore[p] == core[p];
          }
          keep = !same;  // an alt guide whose upper-cased spacer+PAM equals REF's is redundant
        }
      }
    }
  }
  valid[i] = keep ? 1 : 0;
  if (!keep) atomicAdd(n_invalid, 1ull);

  out.hap[i] = h;
  out.pos[i] = pos;
  out.strand[i] = (uint8_t)s;
  out.start[i] = start;
  out.stop[i] = stop;
  out.flags[i] = has_ref ? 1 : 0;
  const int W = L + 2 * HAWK_PAD;
#pragma unroll
  for (int p = 0; p < HAWK_PLANES; ++p) out.win[(size_t)p * out.cap + i] = extract_bits(hs.plane[p] + rowbase, q - HAWK_PAD, W);

  // K4: CFDon = compute_cfd(ref.guide, sg.guide, sg.pam[-2:]) on the 5'->3' guide, i.e. after
  // annotation.reverse_guides for strand 1 (scoring.py:352-387, crisprhawk_scores.py:65-87,
  // cfdscore.py:53-95).  fp64, multiplied left to right exactly like the reference.
  double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
  if (gp.score_cfdon && has_ref && keep) {
    score = 1.0;
    bool err = false;
    const int n = gp.guidelen < 20 ? gp.guidelen : 20;
    for (int k = 0; k < n; ++k) {
      // spacer base k (5'->3'): strand 0 -> + strand offset k; strand 1 -> complement of offset L-1-k
      const int off = s ? L - 1 - k : k;
      const uint32_t cw = (uint32_t)((rcore[0] >> off) & 1) | (uint32_t)((rcore[1] >> off) & 1) << 1 |
                          (uint32_t)((rcore[2] >> off) & 1) << 2 | (uint32_t)((rcore[3] >> off) & 1) << 3;
      const uint32_t cs = (uint32_t)((core[0] >> off) & 1) | (uint32_t)((core[1] >> off) & 1) << 1 |
                          (uint32_t)((core[2] >> off) & 1) << 2 | (uint32_t)((core[3] >> off) & 1) << 3;
      if (cw == cs) continue;
      int a = base_index(cw), b = base_index(cs);
      if (a < 0 || b < 0) { err = true; break; }  // KeyError in the reference
      if (s) { a = 3 - a; b = 3 - b; }
      score *= gp.cfd_mm[(k * 4 + a) * 4 + b];
    }
    if (!err) {
      // PAM[-2:] of the 5'->3' guide: strand 0 -> offsets L-2, L-1; strand 1 -> comp(off 1), comp(off 0)
      const int o0 = s ? 1 : L - 2, o1 = s ? 0 : L - 1;
      const uint32_t c0 = (uint32_t)((core[0] >> o0) & 1) | (uint32_t)((core[1] >> o0) & 1) << 1 |
                          (uint32_t)((core[2] >> o0) & 1) << 2 | (uint32_t)((core[3] >> o0) & 1) << 3;
      const uint32_t c1 = (uint32_t)((core[0] >> o1) & 1) | (uint32_t)((core[1] >> o1) & 1) << 1 |
                          (uint32_t)((core[2] >> o1) & 1) << 2 | (uint32_t)((core[3] >> o1) & 1) << 3;
      int p0 = base_index(c0), p1 = base_index(c1);
      if (p0 < 0 || p1 < 0 || gp.pamlen < 2) err = true;
      else {
        if (s) { p0 = 3 - p0; p1 = 3 - p1; }
        score *= gp.cfd_pam[4 * p0 + p1];
      }
    }
    if (err) { atomicExch(status, -5 /* HAWK_E_CFD */); score = __longlong_as_double(0x7ff8000000000000ll); }
  }
  out.cfdon[i] = score;
}

void hawk_launch_guides(hipStream_t st, const HapSetDev& hs, const GuideParams& gp, const uint32_t* keepF,
                        const uint32_t* keepR, const uint64_t* rec, uint64_t n_rec, GuideCols cols, uint8_t* valid,
                        unsigned long long* n_invalid, int* status) {
  if (!n_rec) return;
  const uint32_t grid = (uint32_t)((n_rec + HAWK_BLOCK - 1) / HAWK_BLOCK);
  hipLaunchKernelGGL(k_guides, dim3(grid), dim3(HAWK_BLOCK), 0, st, hs, gp, keepF, keepR, rec, n_rec, cols, valid, n_invalid,
                     status);
}

// ---------------------------------------------------------------------------------------
// compaction of rows marked redundant
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HAWK_BLOCK) void k_compact_count(const uint8_t* __restrict__ valid, uint64_t n,
                                                               uint32_t* __restrict__ blocksum) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  uint32_t tot;
  block_excl_scan<HAWK_BLOCK / WAVE>(i < n ? valid[i] : 0u, s_w, &tot);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(HAWK_BLOCK) void k_compact_scatter(const uint8_t* __restrict__ valid, uint64_t n,
                                                                 const uint64_t* __restrict__ blockoff, GuideCols src,
                                                                 GuideCols dst) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  const uint32_t v = i < n ? valid[i] : 0u;
  uint32_t tot;
  const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(v, s_w, &tot);
  if (!v) return;
  const uint64_t o = blockoff[blockIdx.x] + ex;
  dst.hap[o] = src.hap[i]; dst.pos[o] = src.pos[i]; dst.strand[o] = src.strand[i];
  dst.start[o] = src.start[i]; dst.stop[o] = src.stop[i]; dst.flags[o] = src.flags[i]; dst.cfdon[o] = src.cfdon[i];
#pragma unroll
  for (int p = 0; p < HAWK_PLANES; ++p) dst.win[(size_t)p * dst.cap + o] = src.win[(size_t)p * src.cap + i];
}

void hawk_launch_compact(hipStream_t st, const uint8_t* valid, uint64_t n, uint32_t* blocksum, uint64_t* blockoff,
                         GuideCols src, GuideCols dst) {
  if (!n) return;
  const uint32_t grid = (uint32_t)((n + HAWK_BLOCK - 1) / HAWK_BLOCK);
  hipLaunchKernelGGL(k_compact_count, dim3(grid), dim3(HAWK_BLOCK), 0, st, valid, n, blocksum);
  hipLaunchKernelGGL(k_offsets, dim3(1), dim3(1024), 0, st, blocksum, blockoff, (uint64_t)grid, (uint64_t)grid,
                     (const uint32_t*)nullptr, (uint64_t)0, (ScanTotals*)(blockoff + grid + 1));
  hipLaunchKernelGGL(k_compact_scatter, dim3(grid), dim3(HAWK_BLOCK), 0, st, valid, n, blockoff, src, dst);
}

// ---------------------------------------------------------------------------------------
// K4 stand-alone: compute_cfd on string triples (cfdscore.py:53-95)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int rna_index(char c) {
  switch (c & 0xDF) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': case 'U': return 3; default: return -1; }
}
__global__ __launch_bounds__(HAWK_BLOCK) void k_cfd(const char* __restrict__ wt, const char* __restrict__ sg, uint32_t len,
                                                     const char* __restrict__ pam2, uint64_t n, const double* __restrict__ mm,
                                                     const double* __restrict__ pamtab, double* __restrict__ out, int* status) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const char* w = wt + i * len;
  const char* s = sg + i * len;
  double score = 1.0;
  bool err = false;
  const uint32_t m = len < 20 ? len : 20;
  for (uint32_t k = 0; k < m; ++k) {
    char cw = w[k] & 0xDF, cs = s[k] & 0xDF;  // upper(); '-' (0x2D) folds to 0x0D on both sides
    if (cw == 'T') cw = 'U';
    if (cs == 'T') cs = 'U';
    if (cw == cs) continue;
    if (w[k] == '-' || s[k] == '-') continue;  // bulge placeholders are skipped
    const int a = rna_index(cw), b = rna_index(cs);
    if (a < 0 || b < 0) { err = true; break; }
    score *= mm[(k * 4 + a) * 4 + b];
  }
  const char p0c = pam2[2 * i] & 0xDF, p1c = pam2[2 * i + 1] & 0xDF;
  const int p0 = rna_index(p0c), p1 = rna_index(p1c);
  if (p0 < 0 || p1 < 0 || p0c == 'U' || p1c == 'U') err = true;
  if (err) { atomicExch(status, -5); out[i] = __longlong_as_double(0x7ff8000000000000ll); return; }
  out[i] = score * pamtab[4 * p0 + p1];
}
void hawk_launch_cfd(hipStream_t st, const char* wt, const char* sg, uint32_t len, const char* pam2, uint64_t n,
                     const double* mm, const double* pamtab, double* out, int* status) {
  if (!n) return;
  hipLaunchKernelGGL(k_cfd, dim3((uint32_t)((n + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, wt, sg, len, pam2, n,
                     mm, pamtab, out, status);
}
