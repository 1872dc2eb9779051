// hawk_rows.h - row-level helpers shared by the plane-based search (hawk_search.hip) and the search from an expansion
// plan (hawk_vsearch.hip): position-map slice search, bit reversal of a slice, CFDon from plane slices.
#pragma once
#include "hawk_bits.h"

#define NSEG 64    // position-map segments staged per tile
__device__ __forceinline__ int seg_find(const uint32_t* s_rel, int n, uint32_t rel) {
  int lo = 0, hi = n;  // last j in [0,n) with s_rel[j] <= rel (s_rel[0] <= every rel of the tile)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (s_rel[mid] <= rel) lo = mid; else hi = mid;
  }
  return lo;
}
// reverse the low L bits of a slice (bit i <-> bit L-1-i), 32 < L <= 64 or L <= 32
__device__ __forceinline__ W2 rev_bits(W2 v, int L) {
  const uint32_t rl = __brev(v.hi), rh = __brev(v.lo);  // 64-bit reversal
  const uint32_t sh = (uint32_t)(64 - L);               // then shift right by 64 - L (0 <= sh < 64)
  if (sh == 0) return W2{rl, rh};
  if (sh < 32) return W2{fsh(rl, rh, sh), rh >> sh};
  return W2{sh == 32 ? rh : rh >> (sh - 32), 0u};
}

// K4: CFDon on the 5'->3' guide.  Strand-1 slices are first turned into the 5'->3' guide (reverse the L
// bits, swap A<->T and C<->G planes = reverse complement), after which both strands read spacer base t at
// bit t and PAM[-2:] at bits L-2, L-1.  Only positions where REF and this guide differ contribute, visited
// in ascending t so the fp64 product is formed exactly as cfdscore.py:78-95 forms it.
__device__ __forceinline__ double cfdon_from_slices(const W2 (&core)[4], const W2 (&rcore)[4], uint32_t s, int L,
                                                    uint32_t cfdmask, const double* s_cfd, bool& err) {
  W2 g[4], r[4];
  if (s) {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) { g[pl] = rev_bits(core[3 - pl], L); r[pl] = rev_bits(rcore[3 - pl], L); }
  } else {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) { g[pl] = core[pl]; r[pl] = rcore[pl]; }
  }
  // spacer positions 0..min(guidelen,20)-1 all sit in the low word
  uint32_t diff = ((g[0].lo ^ r[0].lo) | (g[1].lo ^ r[1].lo) | (g[2].lo ^ r[2].lo) | (g[3].lo ^ r[3].lo)) & cfdmask;
  // a lookup needs both bases to be exactly one of A,C,G,T (else KeyError in the reference)
  const uint32_t g2 = (g[0].lo & g[1].lo) | ((g[0].lo | g[1].lo) & (g[2].lo | g[3].lo)) | (g[2].lo & g[3].lo);
  const uint32_t r2 = (r[0].lo & r[1].lo) | ((r[0].lo | r[1].lo) & (r[2].lo | r[3].lo)) | (r[2].lo & r[3].lo);
  err = (diff & (g2 | r2)) != 0;
  const uint32_t gb0 = g[1].lo | g[3].lo, gb1 = g[2].lo | g[3].lo;  // base index bits: A0 C1 G2 T3
  const uint32_t rb0 = r[1].lo | r[3].lo, rb1 = r[2].lo | r[3].lo;
  double score = 1.0;
  while (diff && !err) {
    const uint32_t t = (uint32_t)__builtin_ctz(diff);
    diff &= diff - 1;
    const uint32_t a = ((rb0 >> t) & 1u) | (((rb1 >> t) & 1u) << 1);
    const uint32_t b = ((gb0 >> t) & 1u) | (((gb1 >> t) & 1u) << 1);
    score *= s_cfd[(t * 4 + a) * 4 + b];
  }
  if (!err) {
    const int o0 = L - 2, o1 = L - 1;  // PAM[-2:] (wave-uniform positions)
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      c0 |= (((o0 < 32 ? g[pl].lo : g[pl].hi) >> (o0 & 31)) & 1u) << pl;
      c1 |= (((o1 < 32 ? g[pl].lo : g[pl].hi) >> (o1 & 31)) & 1u) << pl;
    }
    const int p0 = base_index(c0), p1 = base_index(c1);
    if (p0 < 0 || p1 < 0) err = true;
    else score *= s_cfd[320 + 4 * p0 + p1];
  }
  return err ? __longlong_as_double(0x7ff8000000000000ll) : score;
}

