// hawk_hx.h - records of an expansion plan and the word builder shared by the expansion kernel (hawk_expand.hip) and the
// search that runs straight from a plan (hawk_vsearch.hip).  See hawk_expand.hip for what the records mean.
#pragma once
#include "hawk_bits.h"

#define HX_TW 1024        // output words a workgroup builds: 32768 positions, four consecutive words per thread
#define HX_MAXV 96        // carried variants staged per workgroup (32 B each; more: the rest from global memory)
#define HX_RW (HX_TW + 64)  // REF words staged per plane: the tile's image in REF plus up to 2048 net deleted bases

// o: output start; rs = r0 + span: where REF resumes behind the alt allele; m: the first 32 alt bases as plane bits
struct __attribute__((aligned(16))) HxVar { int32_t o; uint32_t rs; uint32_t alt_len; uint32_t alt_off; uint32_t m[4]; };
// what a workgroup of k_hx_build starts from: one 16-byte scalar load
struct __attribute__((aligned(16))) HxTile {
  uint32_t first_lo, first_hi;  // index (over all rows) of the first record to stage
  uint32_t n_flags;             // records to stage | HX_FITS | HX_ALL
  uint32_t ws;                  // first REF word of the staged window (a multiple of 4)
};
#define HX_FITS (1u << 31)  // the tile's REF image fits the staged window
#define HX_ALL (1u << 30)   // every record the tile needs is staged
#define HX_HEAD (1u << 29)  // the first staged record starts at or before the tile (otherwise the tile starts in unmodified REF)

__device__ __forceinline__ uint32_t hx_low(int n) { return n >= 32 ? 0xffffffffu : (n > 0 ? (1u << n) - 1u : 0u); }

// NW consecutive 32-base words of one row, starting at output position p0 (any position >= 0).  FAST: every record the
// words need is among the n staged ones `s_v`; otherwise records beyond the staged ones come from global memory
// (`first[j]`, j < j_end = the row's records from `first` on).  Records are addressed locally: j = 0 is the first staged
// one.  `ref32(pl, r)` returns 32 bits of REF plane pl from bit r (the caller decides where REF lives).
//
// A word is written as: REF copied under the mapping in force at its first position (the allele of variant k reaching
// into the word first, if it does), then one round per variant starting inside the word - its alt bases shifted out of
// the record's plane bits over what is there, and only when the variant changes the mapping (an indel; a SNV does not)
// the rest of the word copied again.  A round is ~40 instructions; the first version looped over pieces (copy, allele,
// copy ...: three rounds of both branches per variant, 1000 vector instructions per wave).
template <bool FAST, int NW, class RefFn>
__device__ __forceinline__ void hx_words_t(const uint8_t* __restrict__ alt_codes, const HxVar* __restrict__ s_v,
                                           const HxVar* __restrict__ first, int n, int j_end /* records of the row from `first` on */,
                                           bool head, int32_t p0, int32_t len, RefFn ref32,
                                           uint32_t (&oA)[NW], uint32_t (&oC)[NW], uint32_t (&oG)[NW], uint32_t (&oT)[NW], uint32_t (&oV)[NW]) {
  auto getv = [&](int j) -> HxVar { return FAST || j < n ? s_v[j] : first[j]; };
  // FAST: a variant behind the staged ones starts behind everything the words read - all that matters about it
  auto geto = [&](int j) -> int32_t {
    if (FAST) return j < n ? s_v[j].o : 0x7fffffff;
    return j < n ? s_v[j].o : (j < j_end ? first[j].o : 0x7fffffff);
  };
  // k: local index of the last carried variant with o <= p0 (-1: none; only possible when the first staged record
  // starts behind p0).  Records before the first staged one start before it.
  int a = head ? 1 : 0, b = FAST ? n : j_end;
  while (a < b) { const int m = (a + b) >> 1; if (geto(m) <= p0) a = m + 1; else b = m; }
  int k = a - 1;
  // the stretch in force, reloaded when k moves: alt allele [v.o, v_end), behind it REF copied with r = r_base + p
  HxVar v;
  v.o = 0; v.rs = 0; v.alt_len = 0; v.alt_off = 0; v.m[0] = v.m[1] = v.m[2] = v.m[3] = 0;
  int32_t v_end = 0, r_base = 0, next_o = 0x7fffffff;  // k < 0: an empty allele at position 0, REF copied 1:1
  auto load = [&]() {
    if (k >= 0) {
      v = getv(k);
      v_end = v.o + (int32_t)v.alt_len;
      r_base = (int32_t)v.rs - v_end;
    }
    next_o = geto(k + 1);  // output start of the following variant
  };
  load();
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) {
    const int32_t wp0 = p0 + 32 * wi;
    const int32_t wend = wp0 + 32 < len ? wp0 + 32 : len;
    if (next_o <= wp0) { ++k; load(); }  // a variant starting exactly at the word
    uint32_t xA = 0, xC = 0, xG = 0, xT = 0, xV = 0;
    int32_t cs = 0;  // first bit of the word the REF copy fills
    if (wp0 < v_end) {  // the allele of variant k reaches into (or starts at) the word
      const int src = wp0 - v.o, na = v_end - wp0;
      const uint32_t am = hx_low(na);
      if (v.alt_len <= 32u) {
        xA = (v.m[0] >> src) & am; xC = (v.m[1] >> src) & am; xG = (v.m[2] >> src) & am; xT = (v.m[3] >> src) & am;
      } else {  // an insertion longer than a word: base by base from the allele text
        for (int32_t q = wp0; q < v_end && q < wend; ++q) {
          const uint32_t c = alt_codes[v.alt_off + (uint32_t)(q - v.o)], bit = 1u << (q - wp0);
          if (c & 1u) xA |= bit; if (c & 2u) xC |= bit; if (c & 4u) xG |= bit; if (c & 8u) xT |= bit;
        }
      }
      xV = am;
      cs = na;
    }
    if (cs < 32) {
      const uint32_t r = (uint32_t)(r_base + wp0 + cs);
      xA |= ref32(0, r) << cs; xC |= ref32(1, r) << cs; xG |= ref32(2, r) << cs; xT |= ref32(3, r) << cs;
    }
    while (next_o < wend) {  // one round per variant starting inside the word
      const int32_t rb_prev = r_base;
      ++k; load();
      const int s = v.o - wp0;  // 1 .. 31
      const uint32_t am = hx_low((int)v.alt_len) << s;
      if (v.alt_len <= 32u) {
        xA = (xA & ~am) | (v.m[0] << s); xC = (xC & ~am) | (v.m[1] << s); xG = (xG & ~am) | (v.m[2] << s); xT = (xT & ~am) | (v.m[3] << s);
      } else {
        xA &= ~am; xC &= ~am; xG &= ~am; xT &= ~am;
        for (int32_t q = v.o; q < wend; ++q) {  // the allele covers the rest of the word
          const uint32_t c = alt_codes[v.alt_off + (uint32_t)(q - v.o)], bit = 1u << (q - wp0);
          if (c & 1u) xA |= bit; if (c & 2u) xC |= bit; if (c & 4u) xG |= bit; if (c & 8u) xT |= bit;
        }
      }
      xV |= am;
      const int e = v_end - wp0;  // where REF resumes
      if (r_base != rb_prev && e < 32) {  // an indel: the rest of the word maps elsewhere
        const uint32_t cm = 0xffffffffu << e, r = v.rs;
        xA = (xA & ~cm) | (ref32(0, r) << e); xC = (xC & ~cm) | (ref32(1, r) << e);
        xG = (xG & ~cm) | (ref32(2, r) << e); xT = (xT & ~cm) | (ref32(3, r) << e);
      }
    }
    const uint32_t wm = hx_low(wend - wp0);  // the row ends inside (or before) the word
    oA[wi] = xA & wm; oC[wi] = xC & wm; oG[wi] = xG & wm; oT[wi] = xT & wm; oV[wi] = xV & wm;
  }
}
