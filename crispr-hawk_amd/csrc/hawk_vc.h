// hawk_vc.h - device helpers of the searches that run straight from an expansion plan (hawk_vsearch.hip: per dirty word of
// every row; hawk_csearch.hip: per distinct variant cluster): 96-bit strings of the five planes built in registers from REF
// and the carried-variant records, the PAM match on them, REF's hit-prefix tables for the clean stretches.
#pragma once
#include "hawk_hx.h"
#include "hawk_rows.h"

#ifndef VS_ABL
#define VS_ABL 0  // ablation builds only (tools/ab_vs.sh)
#endif

// 32 bits of a REF plane from bit r: one 8-byte request (rows are 4-byte aligned, >= 2 pad words)
struct __attribute__((packed, aligned(4))) U2 { uint32_t a, b; };
__device__ __forceinline__ uint32_t ext32_glb(const uint32_t* __restrict__ row, uint32_t bp) {
  const U2 t = *reinterpret_cast<const U2*>(row + (bp >> 5));
  return fsh(t.a, t.b, bp & 31u);
}
// 64 bits starting at bit `off` (0 <= off < 32) of a 96-bit string held as three words
__device__ __forceinline__ W2 ext96(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t off) { return W2{fsh(x0, x1, off), fsh(x1, x2, off)}; }
// position of the j-th set bit of x (j < popc(x)): halving search on popcounts
__device__ __forceinline__ uint32_t select_bit(uint32_t x, uint32_t j) {
  uint32_t bpos = 0;
#pragma unroll
  for (uint32_t wdt = 16; wdt; wdt >>= 1) {
    const uint32_t c = (uint32_t)__popc((x >> bpos) & ((1u << wdt) - 1u));
    const bool up = j >= c;
    j -= up ? c : 0u;
    bpos += up ? wdt : 0u;
  }
  return bpos;
}
// bit i set <=> for every PAM position t: (pam[t] & base[i + s0 + t]) != 0, over a 96-bit string per plane; s0 + pamlen <= 64
__device__ __forceinline__ uint32_t pam_match96(const uint32_t (&X)[5][3], uint64_t pam, int pamlen, int s0) {
  uint32_t m = 0xffffffffu;
#pragma unroll 1
  for (int i = 0; i < pamlen; ++i) {
    const uint32_t nib = (uint32_t)(pam >> (4 * (pamlen - 1 - i))) & 15u;
    if (nib == 15u) continue;  // wave-uniform
    const int s = s0 + i;
    const int k = s >> 5;
    const uint32_t sh = (uint32_t)(s & 31);
    uint32_t lo = 0, hi = 0;
    if (nib & 1u) { lo |= X[0][k]; hi |= X[0][k + 1]; }
    if (nib & 2u) { lo |= X[1][k]; hi |= X[1][k + 1]; }
    if (nib & 4u) { lo |= X[2][k]; hi |= X[2][k + 1]; }
    if (nib & 8u) { lo |= X[3][k]; hi |= X[3][k + 1]; }
    m &= fsh(lo, hi, sh);
  }
  return m;
}

// REF PAM hits with window starts in [ra, rb), per strand: hp[w] = {hit bits of word w on strand 0, hits in the words before,
// the same for strand 1} - one 16-byte entry serves both strands
__device__ __forceinline__ uint32_t ref_hits_between(const uint4* __restrict__ hp, int s, uint32_t ra, uint32_t rb) {
  const uint4 ea = hp[ra >> 5], eb = hp[rb >> 5];
  const uint32_t ma = (1u << (ra & 31u)) - 1u, mb = (1u << (rb & 31u)) - 1u;
  return s ? (eb.w + (uint32_t)__popc(eb.z & mb)) - (ea.w + (uint32_t)__popc(ea.z & ma))
           : (eb.y + (uint32_t)__popc(eb.x & mb)) - (ea.y + (uint32_t)__popc(ea.x & ma));
}

struct VcRanges {  // per strand, on the window start q: scan range (hits) and scan range x is_pamhit_in_range (candidates)
  int slo[2], shi[2], lo[2], hi[2];
};
// candidates / hits of the clean window starts [pa, pb) of the row: REF's hits under the stretch's shift
__device__ __forceinline__ void vc_count_run(const VcArgs& va, const VcRanges& rg, int32_t pa, int32_t pb, int32_t r_base, uint32_t& cand,
                                             uint32_t& hits) {
  const int lomax = rg.lo[0] > rg.lo[1] ? rg.lo[0] : rg.lo[1], himin = rg.hi[0] < rg.hi[1] ? rg.hi[0] : rg.hi[1];
  if (pa >= lomax && pb <= himin) {  // inside every range of both strands (all but a row's first and last runs): two entries
    const uint32_t ra = (uint32_t)(pa + r_base), rb = (uint32_t)(pb + r_base);
    const uint4 ea = va.hp[ra >> 5], eb = va.hp[rb >> 5];
    const uint32_t ma = (1u << (ra & 31u)) - 1u, mb = (1u << (rb & 31u)) - 1u;
    const uint32_t hc = (eb.y + (uint32_t)__popc(eb.x & mb)) - (ea.y + (uint32_t)__popc(ea.x & ma)) +
                        (eb.w + (uint32_t)__popc(eb.z & mb)) - (ea.w + (uint32_t)__popc(ea.z & ma));
    hits += hc; cand += hc;
    return;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int a = pa > rg.slo[s] ? pa : rg.slo[s], b = pb < rg.shi[s] ? pb : rg.shi[s];
    const int a2 = pa > rg.lo[s] ? pa : rg.lo[s], b2 = pb < rg.hi[s] ? pb : rg.hi[s];
    uint32_t hcount = 0;
    if (a < b) hcount = ref_hits_between(va.hp, s, (uint32_t)(a + r_base), (uint32_t)(b + r_base));
    hits += hcount;
    if (a2 == a && b2 == b) cand += hcount;
    else if (a2 < b2) cand += ref_hits_between(va.hp, s, (uint32_t)(a2 + r_base), (uint32_t)(b2 + r_base));
  }
}

// bits [a, b) of a 96-bit string as three word masks
__device__ __forceinline__ void mask96(int a, int b, uint32_t (&m)[3]) {
  m[0] = range_mask(0, a, b); m[1] = range_mask(32, a, b); m[2] = range_mask(64, a, b);
}
// d |= v << s over 96 bits, -32 < s < 96 (s < 0: the low -s bits of v fall off)
__device__ __forceinline__ void put32(uint32_t (&d)[3], uint32_t v, int s) {
  if (s < 0) { d[0] |= v >> (-s); return; }
  const int k = s >> 5;
  const uint32_t sh = (uint32_t)(s & 31);
  const uint32_t lo = v << sh, hi = sh ? v >> (32u - sh) : 0u;
  if (k == 0) { d[0] |= lo; d[1] |= hi; }
  else if (k == 1) { d[1] |= lo; d[2] |= hi; }
  else d[2] |= lo;
}
// 96 bits of a REF plane from bit r: one 16-byte request.  A position outside REF can only be asked for under a mapping
// whose bits an allele overwrites or the row's end masks: the word index is clamped.
struct __attribute__((packed, aligned(4))) U4 { uint32_t a, b, c, d; };
__device__ __forceinline__ void fetch96(const uint32_t* __restrict__ plane, uint32_t ref_S, int32_t r, uint32_t (&x)[3]) {
  int w = r >> 5;
  w = w < 0 ? 0 : (w > (int)ref_S - 4 ? (int)ref_S - 4 : w);
  const U4 t = *reinterpret_cast<const U4*>(plane + w);
  const uint32_t sh = (uint32_t)(r & 31);
  x[0] = fsh(t.a, t.b, sh); x[1] = fsh(t.b, t.c, sh); x[2] = fsh(t.c, t.d, sh);
}

// The 96-bit string of the five planes from row position p0 on, when every record it touches is staged and no allele in
// reach is longer than a word (otherwise: false, and the caller takes the expansion's general word builder).  REF is
// fetched ONCE under the mapping in force at p0 - four independent 16-byte requests, one round trip - and again only
// behind an indel, the one kind of record that changes the mapping; SNVs and the alleles themselves are register work.
__device__ __forceinline__ bool vc_string_fast(const VcArgs& va, const HxVar* __restrict__ s_v, int n, int32_t p0, int32_t len,
                                               int32_t p_run, uint32_t (&X)[5][3], int32_t& shift_run) {
  int a = 0, b = n;
  while (a < b) { const int m = (a + b) >> 1; if (s_v[m].o <= p0) a = m + 1; else b = m; }
  const int k = a - 1;  // the last record starting at or before p0 (-1: the row starts in unmodified REF)
  int32_t r_base = 0, v_end = 0;
  HxVar v;
  if (k >= 0) {
    v = s_v[k];
    v_end = v.o + (int32_t)v.alt_len;
    r_base = (int32_t)v.rs - v_end;
    if (v.alt_len > 32u && p0 < v_end) return false;
  }
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) fetch96(va.ref[pl], va.ref_S, r_base + p0, X[pl]);
  X[4][0] = X[4][1] = X[4][2] = 0;
  if (k >= 0 && p0 < v_end) {  // the allele of record k reaches into the string: bits [0, na), na <= 32
    const int src = p0 - v.o;
    const uint32_t am = hx_low(v_end - p0);
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) X[pl][0] = (X[pl][0] & ~am) | ((v.m[pl] >> src) & am);
    X[4][0] = am;
  }
  shift_run = r_base;  // the REF shift in force at p_run (> p0): that of the last record starting at or before it
  if (VS_ABL == 6) return true;
  // Records starting inside the string, in two sweeps so that a wave does not run the expensive case once per record
  // ordinal: first the records that are not SNVs (one in ten: an indel re-maps everything behind its allele - a second
  // REF fetch - and its allele is up to a word long), then the SNVs, one bit per plane each.  Alleles never overlap and
  // every REF rewrite is done before the first SNV bit is set, so the order does not change the result.
  int j_end = k + 1;
  for (int j = k + 1; j < n; ++j) {
    const int32_t o = s_v[j].o;
    if (o - p0 >= 96) break;
    j_end = j + 1;
    const uint32_t al = s_v[j].alt_len;
    const int32_t rb = (int32_t)s_v[j].rs - (o + (int32_t)al);
    if (o <= p_run) shift_run = rb;
    if (al == 1u && rb == r_base) continue;  // a SNV: second sweep
    if (al > 32u) return false;
    v = s_v[j];
    const int s = o - p0, e = s + (int)al;
    uint32_t am[3];
    if (rb != r_base && e < 96) {
      uint32_t Y[3];
      mask96(e, 96, am);
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        fetch96(va.ref[pl], va.ref_S, rb + p0, Y);
        X[pl][0] = (X[pl][0] & ~am[0]) | (Y[0] & am[0]); X[pl][1] = (X[pl][1] & ~am[1]) | (Y[1] & am[1]); X[pl][2] = (X[pl][2] & ~am[2]) | (Y[2] & am[2]);
      }
    }
    r_base = rb;
    mask96(s, e, am);
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      X[pl][0] &= ~am[0]; X[pl][1] &= ~am[1]; X[pl][2] &= ~am[2];
      put32(X[pl], v.m[pl], s);
    }
    X[4][0] |= am[0]; X[4][1] |= am[1]; X[4][2] |= am[2];
  }
  {
    int32_t rb_prev = k >= 0 ? (int32_t)s_v[k].rs - (s_v[k].o + (int32_t)s_v[k].alt_len) : 0;
    for (int j = k + 1; j < j_end; ++j) {
      const int32_t o = s_v[j].o;
      const uint32_t al = s_v[j].alt_len;
      const int32_t rb = (int32_t)s_v[j].rs - (o + (int32_t)al);
      const bool snv = al == 1u && rb == rb_prev;
      rb_prev = rb;
      if (!snv) continue;
      const int s = o - p0;
      const uint32_t bit = 1u << (s & 31);
      const uint32_t b0 = s < 32 ? bit : 0u, b1 = (s >= 32 && s < 64) ? bit : 0u, b2 = s >= 64 ? bit : 0u;
      const uint4 m4 = *reinterpret_cast<const uint4*>(&s_v[j].m[0]);
      const uint32_t mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        const uint32_t on = 0u - (mm[pl] & 1u);
        X[pl][0] = (X[pl][0] & ~b0) | (b0 & on); X[pl][1] = (X[pl][1] & ~b1) | (b1 & on); X[pl][2] = (X[pl][2] & ~b2) | (b2 & on);
      }
      X[4][0] |= b0; X[4][1] |= b1; X[4][2] |= b2;
    }
  }
  uint32_t em[3];
  mask96(0, len - p0, em);  // the row ends inside the string
#pragma unroll
  for (int pl = 0; pl < 5; ++pl) { X[pl][0] &= em[0]; X[pl][1] &= em[1]; X[pl][2] &= em[2]; }
  return true;
}

