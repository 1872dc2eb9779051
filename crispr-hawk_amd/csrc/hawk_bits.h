// hawk_bits.h — device helpers shared by the kernels: funnel shifts over bit-sliced planes,
// wave64 / workgroup scans, the PAM match and the sliding-window OR.  gfx950, wave = 64.
#pragma once
#include "hawk_device.h"

#define WAVE 64

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
// DPP lane moves (v_mov_b32_dpp): lanes without a source keep 0
#define DPP0(v, ctrl, rmask, bmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), (bmask), false))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
// inclusive prefix sum over the 64 lanes: 16-lane rows by row_shr, then row broadcasts (8 DPP adds)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  uint32_t v = x + DPP0(x, DPP_ROW_SHR(1), 0xf, 0xf);
  v += DPP0(x, DPP_ROW_SHR(2), 0xf, 0xf);
  v += DPP0(x, DPP_ROW_SHR(3), 0xf, 0xf);
  v += DPP0(v, DPP_ROW_SHR(4), 0xf, 0xe);
  v += DPP0(v, DPP_ROW_SHR(8), 0xf, 0xc);
  v += DPP0(v, DPP_ROW_BCAST15, 0xa, 0xf);
  v += DPP0(v, DPP_ROW_BCAST31, 0xc, 0xf);
  return v;
}
// sum over the 64 lanes, returned in every lane (lane 63 of the inclusive scan, via v_readlane)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), WAVE - 1);
}
// exclusive scan over a workgroup of NW wavefronts; *total gets the workgroup sum.
// s_w: NW words of LDS.  Contains two barriers: every thread of the workgroup must call it.
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

// rank of this thread among the threads of the workgroup with `flag` set, and their number; two barriers
template <int NW>
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* s_w, uint32_t* total) {
  const unsigned long long b = __ballot(flag);
  const int wv = threadIdx.x / WAVE;
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[wv] = (uint32_t)__popcll(b);
  __syncthreads();
  uint32_t pre = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t x = s_w[i];
    if (i < wv) pre += x;
    tot += x;
  }
  __syncthreads();
  *total = tot;
  return pre + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
}

// This thread's 4 words of one plane plus 2 look-ahead words (taken from the next lane's
// registers; the last lane of a wave reads them from memory).  Split in two so that a kernel can
// put every plane's loads in flight (load6_issue) before the first use waits for any of them
// (load6_finish): with the DPP right behind the load the compiler serialises the planes, one full
// memory round trip each.  All 64 lanes must call both halves.
struct Ld6 {
  uint4 v;
  uint2 t;
};
__device__ __forceinline__ void load6_issue(const uint32_t* __restrict__ row, uint32_t u, uint32_t S, bool active, Ld6& r) {
  r.v = make_uint4(0, 0, 0, 0);
  r.t = make_uint2(0, 0);
  if (__all(active)) r.v = *reinterpret_cast<const uint4*>(row + 4 * (size_t)u);  // every wave but a row's last: no exec-mask detour
  else if (active) r.v = *reinterpret_cast<const uint4*>(row + 4 * (size_t)u);
  if ((threadIdx.x & (WAVE - 1)) == WAVE - 1 && active && 4 * u + 4 < S)
    r.t = *reinterpret_cast<const uint2*>(row + 4 * (size_t)u + 4);  // 16-byte aligned
}
__device__ __forceinline__ void load6_finish(const Ld6& r, uint32_t (&a)[6]) {
  a[0] = r.v.x; a[1] = r.v.y; a[2] = r.v.z; a[3] = r.v.w;
  const uint32_t nx = DPP0(r.v.x, DPP_WAVE_SHL1, 0xf, 0xf), ny = DPP0(r.v.y, DPP_WAVE_SHL1, 0xf, 0xf);  // lane i <- lane i+1
  const bool last = (threadIdx.x & (WAVE - 1)) == WAVE - 1;
  a[4] = last ? r.t.x : nx;
  a[5] = last ? r.t.y : ny;
}
__device__ __forceinline__ void load6(const uint32_t* __restrict__ row, uint32_t u, uint32_t S, bool active,
                                      uint32_t (&a)[6]) {
  Ld6 r;
  load6_issue(row, u, S, active, r);
  load6_finish(r, a);
}

// m[k] bit j  <=>  for every PAM position i: (pam[i] & base[32*(4u+k) + j + po + i]) != 0
// (search_guides.py:32-46: set intersection per nibble; an N nibble matches any real base).
__device__ __forceinline__ void pam_match(const uint32_t (&A)[6], const uint32_t (&C)[6], const uint32_t (&G)[6],
                                          const uint32_t (&T)[6], uint64_t pam, int pamlen, int po, uint32_t (&m)[4]) {
  m[0] = m[1] = m[2] = m[3] = 0xffffffffu;
#pragma unroll 1
  for (int i = 0; i < pamlen; ++i) {
    const uint32_t nib = (uint32_t)(pam >> (4 * (pamlen - 1 - i))) & 15u;
    if (nib == 15u) continue;  // wave-uniform: pam is a kernel argument
    const int s = po + i;
    // single-base PAM positions (the common case) shift the plane's registers directly; the word the shift starts
    // in is a wave-uniform branch, not a per-word select
#define PAM_AND_PLANE(X)                                                                    \
    if (s < 32) { _Pragma("unroll") for (int k = 0; k < 4; ++k) m[k] &= fsh(X[k], X[k + 1], (uint32_t)s); } \
    else { _Pragma("unroll") for (int k = 0; k < 4; ++k) m[k] &= fsh(X[k + 1], X[k + 2], (uint32_t)(s - 32)); }
    if (nib == 1u) {
      PAM_AND_PLANE(A)
    } else if (nib == 2u) {
      PAM_AND_PLANE(C)
    } else if (nib == 4u) {
      PAM_AND_PLANE(G)
    } else if (nib == 8u) {
      PAM_AND_PLANE(T)
    } else {
      const uint32_t mA = (nib & 1u) ? 0xffffffffu : 0u, mC = (nib & 2u) ? 0xffffffffu : 0u;
      const uint32_t mG = (nib & 4u) ? 0xffffffffu : 0u, mT = (nib & 8u) ? 0xffffffffu : 0u;
      uint32_t sel[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) sel[k] = (A[k] & mA) | (C[k] & mC) | (G[k] & mG) | (T[k] & mT);
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] &= shifted(sel, k, s);
    }
  }
#undef PAM_AND_PLANE
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

// A <= 64-bit slice of a plane as two 32-bit words: every operation on it is a full-rate 32-bit
// VALU op (per-lane variable 64-bit shifts are quarter rate on CDNA and dominated phase C).
struct W2 { uint32_t lo, hi; };
// 64 bits of a plane row starting at bit bp, one 12-byte request (rows are 4-byte aligned, >= 2 pad words)
struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };
__device__ __forceinline__ W2 ext_glb(const uint32_t* __restrict__ row, uint32_t bp) {
  const uint32_t w = bp >> 5, sh = bp & 31u;
  const U3 t = *reinterpret_cast<const U3*>(row + w);
  return W2{fsh(t.a, t.b, sh), fsh(t.b, t.c, sh)};
}
__device__ __forceinline__ int base_index(uint32_t code) {  // A,C,G,T -> 0..3, anything else -1
  return (code & (code - 1u)) ? -1 : (code ? __builtin_ctz(code) : -1);
}
// haplotype position map (haplotype.py:90-159) as unit-slope segments, global-memory search
__device__ __forceinline__ int64_t posmap_global(const HapSetDev& hs, uint32_t h, uint32_t rel) {
  uint32_t lo = hs.seg_off[h], hi = hs.seg_off[h + 1];  // last k in [lo,hi) with seg_rel[k] <= rel
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (hs.seg_rel[mid] <= rel) lo = mid; else hi = mid;
  }
  return hs.seg_gen[lo] + (int64_t)(rel - hs.seg_rel[lo]);
}
