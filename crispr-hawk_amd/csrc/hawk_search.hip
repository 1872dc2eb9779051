// hawk_search.hip — the fused guide-search kernel (K2 + K3 + K4 in one) and its offset scan.
//
// Three kernels over the same tiles (one workgroup = 1024 plane words = 32 768 positions of one
// haplotype):
//   k_search_count  stream the planes the PAM names + V, PAM-match both strands, apply scan-range /
//                   in-range / REF-identical filters, classify the survivors against the REF
//                   haplotype (remove_redundant_guides); write ONE count per tile and the tile's
//                   hand-over list of valid survivors (window start, strand, has-REF; <= 512);
//   (k_mscan1-3: exclusive scan of the tile counts -> row offsets, totals)
//   k_emit_list     stage the tile's five plane slices in LDS and assemble one finished row per
//                   list entry - coordinates, flags, CFDon, packed window - at deterministic
//                   offsets, coalesced;
//   k_search_emit   tiles beyond the list (REF tiles): recompute the bits and emit (search_tile<1>).
// Survivors of a tile are first compacted into an LDS list so that row work is spread evenly
// over the 256 threads regardless of where in the tile the variants cluster; the tile's slice of
// the haplotype position map sits in LDS next to it.
//
// Reference semantics: search_guides.py:32-46 (match), 87-99 (scan), 395-420 (in range),
// 468-471 (REF-identical skip), 260-280 (coordinates), 340-369 (redundancy), 134-160 (window);
// scoring.py:352-387 + cfdscore.py:53-95 after annotation.py:27-51 (CFDon).
#include "hawk_rows.h"

#define CAP 512    // survivors staged per round (2 per thread)
#define TILE_WORDS (HAWK_BLOCK * HAWK_WPT)  // 1024 plane words per tile
#define LDS_OFF 4                            // tile word w lives at s_pl[p][LDS_OFF + w]; word -1 at [3]
#define LDS_ROW (TILE_WORDS + 8)             // + halo: 1 word before, 2 after (+ pad)
#ifndef COUNT_WAVES
#define COUNT_WAVES 7
#endif
#define LIST_CAP 512                         // valid survivors a tile may hand from the count pass to the emit pass

// 64 bits starting at tile-relative bit position bp (>= -32) of a staged plane slice
__device__ __forceinline__ W2 ext_lds(const uint32_t* pl, int bp) {
  const uint32_t x = (uint32_t)(bp + 32 * LDS_OFF);
  const uint32_t w = x >> 5, sh = x & 31u;
  const uint32_t a = pl[w], b = pl[w + 1], c = pl[w + 2];
  return W2{fsh(a, b, sh), fsh(b, c, sh)};
}
template <int PASS>
__device__ __forceinline__ void search_tile(const HapSetDev& hs, const ScanParams& p_in, const GuideParams& gp, const RefInfo& ri,
                                            const TileMeta* __restrict__ tmeta, uint32_t* __restrict__ counts,
                                            unsigned long long* __restrict__ shards,
                                            const uint64_t* __restrict__ offsets, const GuideCols& out, int* status,
                                            uint32_t* __restrict__ lists, const uint32_t tile, uint32_t* __restrict__ big_count,
                                            unsigned long long* __restrict__ big_list, const int round_only) {
  __shared__ __attribute__((aligned(16))) uint32_t s_pl[PASS == 1 ? HAWK_PLANES : 1][PASS == 1 ? LDS_ROW : 8];
  __shared__ uint32_t s_list[PASS == 1 ? CAP : 1];
  // count pass: every thread's survivor bits (4 words per strand) and packed exclusive offsets, so that phase C
  // can find "survivor number i of the tile" by search instead of every thread looping over its own bits
  __shared__ __attribute__((aligned(16))) uint32_t s_kw[PASS == 0 ? 2 : 1][PASS == 0 ? HAWK_BLOCK * 4 : 4];
  __shared__ uint32_t s_ex[PASS == 0 ? HAWK_BLOCK : 1];
  __shared__ uint32_t s_segrel[NSEG];
  __shared__ int64_t s_seggen[NSEG];
  __shared__ double s_cfd[PASS == 1 ? 336 : 1];
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  __shared__ uint32_t s_acc[4];
  const ScanParams& p = p_in;
  const uint32_t tid = threadIdx.x;
  // list mode: the count pass hands the valid survivors of small tiles to k_emit_list; this kernel's emit
  // pass then only serves the tiles whose list did not fit (REF tiles, very dense tiles)
  const bool list_mode = PASS == 0 && lists != nullptr;
  if (PASS == 1 && lists != nullptr && counts[tile] <= LIST_CAP) return;
  // The tile's row and index within the row are arithmetic (tmeta is laid out [row][tile]), so the plane loads below do
  // not wait for the TileMeta record: both go out together and one level of memory latency leaves the wave's lifetime.
  const uint32_t h = tile / p.bph, blk = tile - h * p.bph;
  const TileMeta tm = tmeta[tile];  // one scalar load: every other haplotype / tile scalar the workgroup needs
  const uint32_t u = blk * HAWK_BLOCK + tid;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
  const int haplen = (int)tm.hap_len;
  const int ss = tm.scan_start, se = tm.scan_stop;
  const bool isref = tm.is_ref != 0;
  const bool dedup = ri.index >= 0 && !isref;  // rows of this tile can be redundant with REF
  const bool stage = PASS == 1 || dedup || list_mode;  // phase C runs if the tile has survivors
  const bool lds_planes = PASS == 1;            // PASS 0 classifies its few survivors straight from L2/HBM
  const uint32_t w0 = blk * TILE_WORDS;         // first plane word of the tile
  const uint32_t tile_q0 = w0 * 32u;
  const uint32_t tile_end = tile_q0 + TILE_WORDS * 32u + 64u;  // rel < tile_end can be looked up
  if (tid < 4) s_acc[tid] = 0;

  // ---- every global load of the tile is issued here, before any of it is consumed -------
  uint32_t A[6] = {0, 0, 0, 0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0}, G[6] = {0, 0, 0, 0, 0, 0}, Tp[6] = {0, 0, 0, 0, 0, 0};
  uint32_t E[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0};
  const uint32_t k0 = tm.seg0, kend = tm.seg_end;
  const uint32_t ldmask = lds_planes ? 15u : p.need;
  Ld6 la, lc, lg, lt, lv;
  la.v = lc.v = lg.v = lt.v = lv.v = make_uint4(0, 0, 0, 0);
  la.t = lc.t = lg.t = lt.t = lv.t = make_uint2(0, 0);
  if (ldmask & 1u) load6_issue(hs.plane[0] + rowbase, u, hs.S, active, la);
  if (ldmask & 2u) load6_issue(hs.plane[1] + rowbase, u, hs.S, active, lc);
  if (ldmask & 4u) load6_issue(hs.plane[2] + rowbase, u, hs.S, active, lg);
  if (ldmask & 8u) load6_issue(hs.plane[3] + rowbase, u, hs.S, active, lt);
  // REF windows are never filtered, but whether this is a REF row is TileMeta's to say: the V plane is fetched either way
  load6_issue(hs.plane[4] + rowbase, u, hs.S, active, lv);
  uint32_t halo = 0, seg_r = 0xffffffffu;
  int64_t seg_g = 0;
  bool seg_in = false, ovf = false;
  if (stage) {
    if (lds_planes && tid < 3 * HAWK_PLANES) {
      const uint32_t pl = tid / 3, j = tid % 3;  // j: 0 -> word -1, 1 -> word TILE_WORDS, 2 -> TILE_WORDS+1
      const long long w = j == 0 ? (long long)w0 - 1 : (long long)w0 + TILE_WORDS + (j - 1);
      if (w >= 0 && w < (long long)hs.S) halo = hs.plane[pl][rowbase + (size_t)w];
    }
    if (tid < NSEG) {  // the tile's slice of the position map
      const uint32_t k = k0 + tid;
      if (k < kend) {
        seg_r = hs.seg_rel[k];
        seg_g = hs.seg_gen[k];
        seg_in = tid == 0 || seg_r < tile_end;
      }
    }
    ovf = k0 + NSEG < kend && hs.seg_rel[k0 + NSEG] < tile_end;  // rare: > NSEG segments in a tile
    if (PASS == 1 && gp.score_cfdon) for (uint32_t i = tid; i < 336; i += HAWK_BLOCK) s_cfd[i] = gp.cfd_mm[i];  // mm[320] then pam[16]
  }

  // ---- phase A: scan + filters on registers; stage the slices phase C reads --------------
  load6_finish(la, A); load6_finish(lc, C); load6_finish(lg, G); load6_finish(lt, Tp);
  uint32_t V[4] = {lv.v.x, lv.v.y, lv.v.z, lv.v.w};
  if (!isref) load6_finish(lv, E);
  if (stage && lds_planes) {
    *reinterpret_cast<uint4*>(&s_pl[0][LDS_OFF + 4 * tid]) = make_uint4(A[0], A[1], A[2], A[3]);
    *reinterpret_cast<uint4*>(&s_pl[1][LDS_OFF + 4 * tid]) = make_uint4(C[0], C[1], C[2], C[3]);
    *reinterpret_cast<uint4*>(&s_pl[2][LDS_OFF + 4 * tid]) = make_uint4(G[0], G[1], G[2], G[3]);
    *reinterpret_cast<uint4*>(&s_pl[3][LDS_OFF + 4 * tid]) = make_uint4(Tp[0], Tp[1], Tp[2], Tp[3]);
    *reinterpret_cast<uint4*>(&s_pl[4][LDS_OFF + 4 * tid]) = make_uint4(V[0], V[1], V[2], V[3]);
    if (tid < 3 * HAWK_PLANES) {
      const uint32_t pl = tid / 3, j = tid % 3;
      s_pl[pl][j == 0 ? LDS_OFF - 1 : LDS_OFF + TILE_WORDS + (j - 1)] = halo;
    }
  }
  if (stage) {
    if (tid < NSEG) {  // NSEG == WAVE: wave 0 holds the whole slice
      s_segrel[tid] = seg_in ? seg_r : 0xffffffffu;
      s_seggen[tid] = seg_g;
      if (PASS == 1) {  // the count pass searches the sentinel-padded slice with a fixed step count and needs no length
        const unsigned long long bm = __ballot(seg_in);
        if (tid == 0) s_acc[3] = (uint32_t)__popcll(bm);
      }
    }
  }
  uint32_t kF[4], kR[4];
  uint32_t cand = 0, hits = 0;
  {
    if (!isref) window_or(E, p.L);
    const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
    uint32_t mF[4], mR[4];
    pam_match(A, C, G, Tp, p.pam_fwd, p.pamlen, poF, mF);
    pam_match(A, C, G, Tp, p.pam_rev, p.pamlen, poR, mR);
    const int sloF = ss - poF, shiF = se - poF, sloR = ss - poR, shiR = se - poR;
    const int qmin = HAWK_PAD, qmax = haplen - p.L - HAWK_PAD + 1;  // is_pamhit_in_range on q
    const int loF = sloF > qmin ? sloF : qmin, hiF = shiF < qmax ? shiF : qmax;
    const int loR = sloR > qmin ? sloR : qmin, hiR = shiR < qmax ? shiR : qmax;
    const int base0 = (int)(u * 128u);
    // the fast path is taken per WAVE (a per-lane branch would cost exec-mask bookkeeping on the scalar unit in every wave)
    const bool interior = __all(base0 >= loF && base0 >= loR && base0 + 128 <= hiF && base0 + 128 <= hiR);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t f = mF[k], r = mR[k];
      if (!interior) {
        const int b = base0 + 32 * k;
        f &= range_mask(b, sloF, shiF);
        r &= range_mask(b, sloR, shiR);
        hits += __popc(f) + __popc(r);
        f &= range_mask(b, loF, hiF);
        r &= range_mask(b, loR, hiR);
        cand += __popc(f) + __popc(r);
      } else {
        cand += __popc(f) + __popc(r);  // interior words: every PAM hit is also in range
      }
      kF[k] = active ? (f & E[k]) : 0u;
      kR[k] = active ? (r & E[k]) : 0u;
    }
    if (interior) hits = cand;
    if (!active) { cand = 0; hits = 0; }
  }
  const uint32_t cF = __popc(kF[0]) + __popc(kF[1]) + __popc(kF[2]) + __popc(kF[3]);
  const uint32_t cR = __popc(kR[0]) + __popc(kR[1]) + __popc(kR[2]) + __popc(kR[3]);
  if (PASS == 0) {
    *reinterpret_cast<uint4*>(&s_kw[0][4 * tid]) = make_uint4(kF[0], kF[1], kF[2], kF[3]);
    *reinterpret_cast<uint4*>(&s_kw[1][4 * tid]) = make_uint4(kR[0], kR[1], kR[2], kR[3]);
  }
  // one scan for both strands: per-thread counts <= 128, workgroup totals <= 32768 < 2^16
  uint32_t TT;
  const uint32_t exFR = block_excl_scan<HAWK_BLOCK / WAVE>(cF | (cR << 16), s_w, &TT);  // barriers also publish the LDS staging
  if (PASS == 0) s_ex[tid] = exFR;  // published by the barrier in front of phase C
  const uint32_t TF = TT & 0xffffu, TR = TT >> 16, exF = exFR & 0xffffu, exR = exFR >> 16;
  const uint32_t T = TF + TR;

  uint32_t nvalid = 0;  // this thread's share (PASS 0)
  uint32_t lrun = 0;  // valid survivors listed so far (list mode)
  if (!stage || (PASS == 0 && !dedup && (!list_mode || T > LIST_CAP))) {
    if (tid == 0) nvalid = T;  // nothing can be redundant here and no list is wanted: the count is the survivor count
  } else if (T) {  // workgroup-uniform
    const int nloc = (int)s_acc[3];
    const size_t refbase = ri.index >= 0 ? (size_t)ri.index * hs.S : 0;
    // PASS 1 on one round of a REF tile (round_only >= 0): every survivor of a REF tile is a row, so round b starts
    // b * CAP rows into the tile and the rounds of one tile can run in different workgroups
    uint64_t row_base = PASS == 1 ? offsets[tile] + (round_only >= 0 ? (uint64_t)round_only * CAP : 0ull) : 0;
    const int L = p.L;
    const int W = L + 2 * HAWK_PAD;
    const uint32_t mlo = L >= 32 ? 0xffffffffu : ((1u << L) - 1u), mhi = L <= 32 ? 0u : ((1u << (L - 32)) - 1u);
    const uint32_t wlo = 0xffffffffu, whi = W >= 64 ? 0xffffffffu : ((1u << (W - 32)) - 1u);  // W = L + 20 > 32
    const int ncfd = gp.guidelen < 20 ? gp.guidelen : 20;
    const uint32_t cfdmask = (1u << ncfd) - 1u;

    const uint32_t base_lo = (PASS == 1 && round_only >= 0) ? (uint32_t)round_only * CAP : 0u;
    const uint32_t base_hi = (PASS == 1 && round_only >= 0) ? (base_lo + CAP < T ? base_lo + CAP : T) : T;
    for (uint32_t base = base_lo; base < base_hi; base += CAP) {
      // ---- phase B (emit pass): survivors -> LDS list, strand 0 first, each in position order ------
      // (the count pass needs no list: its phase C looks survivor number i up in s_ex / s_kw)
      if (PASS == 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t idx = s ? TF + exR : exF;
        const uint32_t cnt = s ? cR : cF;
        if (cnt && idx < base + CAP && idx + cnt > base) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint32_t x = s ? kR[k] : kF[k];
            while (x) {
              const uint32_t j = (uint32_t)__builtin_ctz(x);
              x &= x - 1;
              if (idx >= base && idx < base + CAP) s_list[idx - base] = ((uint32_t)s << 31) | ((4 * tid + k) * 32 + j);
              ++idx;
            }
          }
        }
      }
      }
      __syncthreads();
      const uint32_t n = T - base < CAP ? T - base : CAP;
      // ---- phase C: one survivor per thread per round ---------------------------------
#pragma unroll 1
      for (uint32_t k = 0; k < CAP / HAWK_BLOCK; ++k) {
        if (k * HAWK_BLOCK >= n) break;  // workgroup-uniform
        const uint32_t i = tid + k * HAWK_BLOCK;
        uint32_t valid = 0;
        uint32_t ql = 0, s = 0;  // ql: window start relative to the tile
        int64_t start = 0, stop = 0;
        bool has_ref = false;
        W2 core[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, rcore[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        // ---- count pass: survivor lookup, position map and verdict as straight-line code.  Every lane runs it (lanes past
        // the last survivor redo survivor `base` and are masked at the end): per-lane branches and data-dependent loops cost
        // exec-mask bookkeeping on the CU's one scalar unit in every wave, which is what this kernel was bound by.
        int64_t c_start = 0;
        bool c_has_ref = false;
        uint32_t c_valid = 0;
        if (PASS == 0 && (i & ~(uint32_t)(WAVE - 1)) < n) {  // wave-uniform: a wave whose 64 slots lie past the last survivor skips
          // survivor number g of the tile (strand 0 in position order, then strand 1): the thread that found it is the
          // last one whose exclusive offset is <= the survivor's rank within its strand
          const uint32_t g = base + (i < n ? i : 0u);
          s = g >= TF ? 1u : 0u;
          const uint32_t target = s ? g - TF : g;
          const uint32_t sh16 = s ? 16u : 0u;
          uint32_t lo = 0;
#pragma unroll
          for (uint32_t step = HAWK_BLOCK / 2; step; step >>= 1) lo += (((s_ex[lo + step] >> sh16) & 0xffffu) <= target) ? step : 0u;
          uint32_t j = target - ((s_ex[lo] >> sh16) & 0xffffu);  // its rank among that thread's bits
          const uint4 w4 = *reinterpret_cast<const uint4*>(&s_kw[s][4 * lo]);
          const uint32_t c0 = (uint32_t)__popc(w4.x), c1 = c0 + (uint32_t)__popc(w4.y), c2 = c1 + (uint32_t)__popc(w4.z);
          const uint32_t kw = (j >= c0 ? 1u : 0u) + (j >= c1 ? 1u : 0u) + (j >= c2 ? 1u : 0u);
          const uint32_t x = kw == 0 ? w4.x : kw == 1 ? w4.y : kw == 2 ? w4.z : w4.w;
          j -= kw == 0 ? 0u : kw == 1 ? c0 : kw == 2 ? c1 : c2;
          uint32_t bpos = 0;  // position of the j-th set bit of x: halving search on popcounts
#pragma unroll
          for (uint32_t wdt = 16; wdt; wdt >>= 1) {
            const uint32_t c = (uint32_t)__popc((x >> bpos) & ((1u << wdt) - 1u));
            const bool up = j >= c;
            j -= up ? c : 0u;
            bpos += up ? wdt : 0u;
          }
          ql = (4 * lo + kw) * 32 + bpos;
          const uint32_t q = tile_q0 + ql;
          if (ovf) {  // workgroup-uniform, rare: more than NSEG segments in the tile
            c_start = posmap_global(hs, h, q);
          } else {    // unused slots of s_segrel hold 0xffffffff: a fixed six-step search needs no bounds
            uint32_t sj = 0;
#pragma unroll
            for (uint32_t step = NSEG / 2; step; step >>= 1) sj += (s_segrel[sj + step] <= q) ? step : 0u;
            c_start = s_seggen[sj] + (int64_t)(q - s_segrel[sj]);
          }
          c_valid = 1;
          c_has_ref = isref;
          if (dedup) {  // workgroup-uniform
            // A REF guide shares (start, strand) iff REF has a candidate window starting at qr = start - startp: one bit of
            // the per-strand bitmaps k_ref_bits left in HBM (2 x 125 KB on C3, L2-resident).  The survivor is redundant
            // iff the four code planes agree as well: the planes the PAM names (just streamed, L2-hot) are compared for
            // every lane, the others only in the rare wave where those agree.
            const int64_t qr64 = c_start - ri.startp;
            const bool inr = qr64 >= 0 && qr64 < (int64_t)ri.n_bits;
            const uint32_t qr = inr ? (uint32_t)qr64 : 0u;
            const uint32_t rw = (s ? ri.bits[1] : ri.bits[0])[qr >> 5];
            c_has_ref = inr && ((rw >> (qr & 31u)) & 1u);
            bool same = c_has_ref;
#pragma unroll
            for (int pl = 0; pl < 4; ++pl)
              if ((p.need >> pl) & 1u) {  // wave-uniform
                const W2 b = ext_glb(hs.plane[pl] + refbase, qr);
                const W2 a = ext_glb(hs.plane[pl] + rowbase, q);
                same = same && ((a.lo ^ b.lo) & mlo) == 0 && ((a.hi ^ b.hi) & mhi) == 0;
              }
            if (__any(same)) {
#pragma unroll
              for (int pl = 0; pl < 4; ++pl)
                if (!((p.need >> pl) & 1u)) {
                  const W2 b = ext_glb(hs.plane[pl] + refbase, qr);
                  const W2 a = ext_glb(hs.plane[pl] + rowbase, q);
                  same = same && ((a.lo ^ b.lo) & mlo) == 0 && ((a.hi ^ b.hi) & mhi) == 0;
                }
            }
            c_valid = same ? 0u : 1u;
          }
        }
        if (i < n) {
          if (PASS == 0) {
            // filled in by the branch-free block in front of `if (i < n)`
          } else {
            const uint32_t e = s_list[i];
            s = e >> 31; ql = e & 0x7fffffffu;
          }
          const uint32_t q = tile_q0 + ql;
          if (PASS == 0) {
            start = c_start;
          } else if (ovf) {
            start = posmap_global(hs, h, q);
            if (PASS == 1) stop = posmap_global(hs, h, q + (uint32_t)L);
          } else {
            const int j = seg_find(s_segrel, nloc, q);
            start = s_seggen[j] + (int64_t)(q - s_segrel[j]);
            if (PASS == 1) {  // search_guides.py:260-280: stop = posmap[q + L]
              if (j + 1 >= nloc || s_segrel[j + 1] > q + (uint32_t)L) stop = start + L;
              else { const int j2 = seg_find(s_segrel, nloc, q + (uint32_t)L); stop = s_seggen[j2] + (int64_t)(q + (uint32_t)L - s_segrel[j2]); }
            }
          }
          valid = 1;
          if (PASS == 0) {
            has_ref = c_has_ref;
            valid = c_valid;
          } else {
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) {
            core[pl] = ext_lds(s_pl[pl], (int)ql);
            core[pl].lo &= mlo; core[pl].hi &= mhi;
            rcore[pl] = core[pl];
          }
          if (ri.index >= 0) {
            if (isref) {
              has_ref = true;
            } else {
              // is there a REF guide with the same (start, strand)?  REF's position map is the
              // identity, so its window starts at qr; it is a guide iff qr lies in REF's candidate
              // range for this strand and REF's PAM matches there.
              const int64_t qr = start - ri.startp;
              if (qr >= ri.lo[s] && qr < ri.hi[s]) {
                W2 rc[4];
#pragma unroll
                for (int pl = 0; pl < 4; ++pl) {
                  rc[pl] = ext_glb(hs.plane[pl] + refbase, (uint32_t)qr);
                  rc[pl].lo &= mlo; rc[pl].hi &= mhi;
                }
                // PAM test on REF's slice: for each PAM position the selected planes must have the bit
                const bool pamfirst = (p.right != 0) != (s != 0);
                const int po = pamfirst ? 0 : p.guidelen;
                const uint64_t pam = s ? p.pam_rev : p.pam_fwd;
                bool ok = true;
                for (int t = 0; t < p.pamlen; ++t) {
                  const uint32_t nib = (uint32_t)(pam >> (4 * (p.pamlen - 1 - t))) & 15u;
                  const int off = po + t;
                  uint32_t sel = 0;
                  if (off < 32) {
                    if (nib & 1u) sel |= rc[0].lo; if (nib & 2u) sel |= rc[1].lo;
                    if (nib & 4u) sel |= rc[2].lo; if (nib & 8u) sel |= rc[3].lo;
                  } else {
                    if (nib & 1u) sel |= rc[0].hi; if (nib & 2u) sel |= rc[1].hi;
                    if (nib & 4u) sel |= rc[2].hi; if (nib & 8u) sel |= rc[3].hi;
                  }
                  ok = ok && ((sel >> (off & 31)) & 1u);
                }
                if (ok) {
                  has_ref = true;
                  bool same = true;
#pragma unroll
                  for (int pl = 0; pl < 4; ++pl) {
                    rcore[pl] = rc[pl];
                    same = same && rc[pl].lo == core[pl].lo && rc[pl].hi == core[pl].hi;
                  }
                  if (same) valid = 0;  // an alt guide whose upper-cased spacer+PAM equals REF's is redundant
                }
              }
            }
          }
          }
        }
        if (PASS == 0) {
          if (list_mode) {  // compact the valid survivors into the tile's hand-over list
            uint32_t tot;
            const uint32_t ex = block_rank<HAWK_BLOCK / WAVE>(valid != 0, s_w, &tot);
            if (valid && lrun + ex < LIST_CAP) lists[(size_t)tile * LIST_CAP + lrun + ex] = ql | (s << 15) | ((uint32_t)has_ref << 16);
            lrun += tot;
            if (tid == 0) nvalid += tot;
          } else {
            nvalid += valid;
          }
        } else {
          uint32_t tot;
          const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(valid, s_w, &tot);
          if (valid) {
            const uint64_t o = row_base + ex;
            if (o >= out.cap) { atomicExch(status, -3 /* HAWK_E_CAPACITY: offsets and counts disagree */); continue; }
            const bool pamfirst = (p.right != 0) != (s != 0);
            const uint32_t q = tile_q0 + ql;
            out.hap[o] = h;
            out.pos[o] = pamfirst ? q : q + (uint32_t)p.guidelen;
            out.strand[o] = (uint8_t)s;
            out.start[o] = start;
            out.stop[o] = stop;
            out.flags[o] = has_ref ? 1 : 0;
#pragma unroll
            for (int pl = 0; pl < HAWK_PLANES; ++pl) {
              W2 w = ext_lds(s_pl[pl], (int)ql - HAWK_PAD);
              w.lo &= wlo; w.hi &= whi;
              out.win[(size_t)pl * out.cap + o] = (uint64_t)w.lo | ((uint64_t)w.hi << 32);
            }
            double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
            if (gp.score_cfdon && has_ref) {
              bool err;
              score = cfdon_from_slices(core, rcore, s, L, cfdmask, s_cfd, err);
              if (err && gp.score_cfdon == 1) atomicExch(status, -5 /* HAWK_E_CFD; score_cfdon == 2 leaves NaN = "NA" */);
            }
            out.cfdon[o] = score;
          }
          row_base += tot;
        }
      }
      __syncthreads();  // the list is rewritten by the next round
    }
  }
  if (PASS == 0) {
    const uint32_t w1s = wave_sum(cand | (hits << 16));  // per-wave sums < 2^16
    // in list mode (and whenever phase C did not run) thread 0 alone holds the tile's count
    const bool spread = stage && !list_mode && T != 0 && !(PASS == 0 && !dedup);
    const uint32_t w0s = spread ? wave_sum(nvalid) : nvalid;  // workgroup-uniform choice
    __syncthreads();  // s_w is free again (the last scan's second barrier lies behind every reader)
    if ((tid & (WAVE - 1)) == 0) { s_w[tid / WAVE] = w0s; s_kw[0][tid / WAVE] = w1s; }
    __syncthreads();
    if (tid == 0) {
      uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
      for (int wv = 0; wv < HAWK_BLOCK / WAVE; ++wv) { a0 += s_w[wv]; a1 += s_kw[0][wv] & 0xffffu; a2 += s_kw[0][wv] >> 16; }
      counts[tile] = a0;
      // tiles whose rows do not fit the hand-over list (REF tiles, very dense ones) are named for k_search_emit, which
      // then runs over them alone instead of launching one workgroup per tile to find out
      // (a REF tile - thousands of rows, none ever dropped - is named once per round of CAP survivors, so that its rounds
      // spread over workgroups: 31 REF tiles of ~4000 rows were a 50 us critical path of their own)
      if (list_mode && a0 > LIST_CAP) {
        const uint32_t nr = isref ? (T + CAP - 1) / CAP : 1u;
        const uint32_t at = atomicAdd(big_count, nr);
        for (uint32_t r = 0; r < nr; ++r)
          big_list[at + r] = (unsigned long long)tile | ((unsigned long long)(isref ? r + 1u : 0u) << 32);
      }
      if (a1 | a2) {
        atomicAdd(&shards[(tile & 255u) * 2 + 0], (unsigned long long)a1);
        atomicAdd(&shards[(tile & 255u) * 2 + 1], (unsigned long long)a2);
      }
    }
  }
}

// The two passes as kernels.  The count pass is latency- and issue-bound, so it is held to the
// register budget of 8 waves per SIMD; the emit pass is bounded by its LDS staging (6 workgroups per CU).
__global__ __launch_bounds__(HAWK_BLOCK) __attribute__((amdgpu_waves_per_eu(COUNT_WAVES, 8)))
void k_search_count(HapSetDev hs, ScanParams p, GuideParams gp, RefInfo ri, const TileMeta* __restrict__ tmeta,
                    uint32_t* __restrict__ counts, unsigned long long* __restrict__ shards, int* status,
                    uint32_t* __restrict__ lists, uint32_t* __restrict__ big_count, unsigned long long* __restrict__ big_list) {
  search_tile<0>(hs, p, gp, ri, tmeta, counts, shards, nullptr, GuideCols{}, status, lists, blockIdx.x, big_count, big_list, -1);
}
__global__ __launch_bounds__(HAWK_BLOCK)
void k_search_emit(HapSetDev hs, ScanParams p, GuideParams gp, RefInfo ri, const TileMeta* __restrict__ tmeta,
                   uint32_t* __restrict__ counts, const uint64_t* __restrict__ offsets, GuideCols out, int* status,
                   uint32_t* __restrict__ lists, const uint32_t* __restrict__ big_count, const unsigned long long* __restrict__ big_list) {
  if (lists == nullptr) {  // no hand-over lists: every tile recomputes and emits here
    search_tile<1>(hs, p, gp, ri, tmeta, counts, nullptr, offsets, out, status, lists, blockIdx.x, nullptr, nullptr, -1);
    return;
  }
  const uint32_t n_big = *big_count;
#pragma unroll 1
  for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {  // workgroup-uniform
    const unsigned long long e = big_list[i];
    search_tile<1>(hs, p, gp, ri, tmeta, counts, nullptr, offsets, out, status, lists, (uint32_t)e, nullptr, nullptr, (int)(e >> 32) - 1);
    __syncthreads();  // the tile's LDS staging is rewritten by the next one
  }
}

// ---------------------------------------------------------------------------------------
// k_emit_list — the emit pass for tiles whose valid survivors fit the hand-over list.
//
// The count pass already did the PAM match, the range / variant-window filters and the REF
// classification, and left (window start, strand, has-REF) per valid survivor in `lists`.  What is
// left is row assembly: stage the tile's five plane slices and its position-map slice in LDS, then
// one thread per row looks up the coordinates, cuts the padded window and the spacer+PAM core out
// of LDS, fetches REF's core when a REF guide shares the key, scores CFDon and stores the row at
// offsets[tile] + i.  No scans, no list building, no barriers after the staging one.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HAWK_BLOCK) void k_emit_list(HapSetDev hs, ScanParams p, GuideParams gp, RefInfo ri,
                                                          const TileMeta* __restrict__ tmeta,
                                                          const uint32_t* __restrict__ counts,
                                                          const uint64_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ lists, GuideCols out, int* status) {
  __shared__ __attribute__((aligned(16))) uint32_t s_pl[HAWK_PLANES][LDS_ROW];
  __shared__ uint32_t s_segrel[NSEG];
  __shared__ int64_t s_seggen[NSEG];
  __shared__ double s_cfd[336];
  __shared__ uint32_t s_nloc;
  const uint32_t tid = threadIdx.x;
  // Workgroups are dealt to the 8 XCDs round-robin, each XCD with its own L2: give every XCD a contiguous
  // run of tiles, so that the rows of neighbouring tiles (which share cache lines at their seams) meet in one L2.
  const uint32_t per = gridDim.x >> 3;
  const uint32_t tile = blockIdx.x < per * 8u ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t n = counts[tile];
  if (n == 0 || n > LIST_CAP) return;  // workgroup-uniform
  const TileMeta tm = tmeta[tile];
  const uint32_t h = tm.h, blk = tm.blk;
  const uint32_t u = blk * HAWK_BLOCK + tid;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
  const bool isref = tm.is_ref != 0;
  const uint32_t w0 = blk * TILE_WORDS;
  const uint32_t tile_q0 = w0 * 32u;
  const uint32_t tile_end = tile_q0 + TILE_WORDS * 32u + 64u;
  if (tid == 0) s_nloc = 0;

  // the survivor entries first: their latency hides behind the staging
  const uint32_t e0 = tid < n ? lists[(size_t)tile * LIST_CAP + tid] : 0u;
  const uint32_t e1 = tid + HAWK_BLOCK < n ? lists[(size_t)tile * LIST_CAP + tid + HAWK_BLOCK] : 0u;
  uint4 v[HAWK_PLANES];
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl)
    v[pl] = active ? *reinterpret_cast<const uint4*>(hs.plane[pl] + rowbase + 4 * (size_t)u) : make_uint4(0, 0, 0, 0);
  uint32_t halo = 0;
  if (tid < 3 * HAWK_PLANES) {
    const uint32_t pl = tid / 3, j = tid % 3;  // j: 0 -> word -1, 1 -> word TILE_WORDS, 2 -> TILE_WORDS+1
    const long long w = j == 0 ? (long long)w0 - 1 : (long long)w0 + TILE_WORDS + (j - 1);
    if (w >= 0 && w < (long long)hs.S) halo = hs.plane[pl][rowbase + (size_t)w];
  }
  const uint32_t k0 = tm.seg0, kend = tm.seg_end;
  uint32_t seg_r = 0xffffffffu;
  int64_t seg_g = 0;
  bool seg_in = false;
  if (tid < NSEG) {
    const uint32_t k = k0 + tid;
    if (k < kend) {
      seg_r = hs.seg_rel[k];
      seg_g = hs.seg_gen[k];
      seg_in = tid == 0 || seg_r < tile_end;
    }
  }
  const bool ovf = k0 + NSEG < kend && hs.seg_rel[k0 + NSEG] < tile_end;
  if (gp.score_cfdon) for (uint32_t i = tid; i < 336; i += HAWK_BLOCK) s_cfd[i] = gp.cfd_mm[i];
#pragma unroll
  for (int pl = 0; pl < HAWK_PLANES; ++pl) *reinterpret_cast<uint4*>(&s_pl[pl][LDS_OFF + 4 * tid]) = v[pl];
  if (tid < 3 * HAWK_PLANES) {
    const uint32_t pl = tid / 3, j = tid % 3;
    s_pl[pl][j == 0 ? LDS_OFF - 1 : LDS_OFF + TILE_WORDS + (j - 1)] = halo;
  }
  if (tid < NSEG) {
    s_segrel[tid] = seg_in ? seg_r : 0xffffffffu;
    s_seggen[tid] = seg_g;
    if (seg_in) atomicAdd(&s_nloc, 1u);
  }
  __syncthreads();
  const int nloc = (int)s_nloc;
  const size_t refbase = ri.index >= 0 ? (size_t)ri.index * hs.S : 0;
  const int L = p.L;
  const int W = L + 2 * HAWK_PAD;
  const uint32_t mlo = L >= 32 ? 0xffffffffu : ((1u << L) - 1u), mhi = L <= 32 ? 0u : ((1u << (L - 32)) - 1u);
  const uint32_t whi = W >= 64 ? 0xffffffffu : ((1u << (W - 32)) - 1u);  // W = L + 20 > 32
  const int ncfd = gp.guidelen < 20 ? gp.guidelen : 20;
  const uint32_t cfdmask = (1u << ncfd) - 1u;
  const uint64_t row0 = offsets[tile];

#pragma unroll 1
  for (uint32_t k = 0; k < LIST_CAP / HAWK_BLOCK; ++k) {
    const uint32_t i = tid + k * HAWK_BLOCK;
    if (i >= n) break;
    const uint32_t e = k ? e1 : e0;
    const uint32_t ql = e & 0x7fffu, s = (e >> 15) & 1u;
    const bool has_ref = ((e >> 16) & 1u) != 0;
    const uint32_t q = tile_q0 + ql;
    int64_t start, stop;
    if (ovf) {
      start = posmap_global(hs, h, q);
      stop = posmap_global(hs, h, q + (uint32_t)L);
    } else {  // search_guides.py:260-280: start = posmap[q], stop = posmap[q + L]
      const int j = seg_find(s_segrel, nloc, q);
      start = s_seggen[j] + (int64_t)(q - s_segrel[j]);
      if (j + 1 >= nloc || s_segrel[j + 1] > q + (uint32_t)L) stop = start + L;
      else { const int j2 = seg_find(s_segrel, nloc, q + (uint32_t)L); stop = s_seggen[j2] + (int64_t)(q + (uint32_t)L - s_segrel[j2]); }
    }
    const uint64_t o = row0 + i;
    if (o >= out.cap) { atomicExch(status, -3 /* HAWK_E_CAPACITY: offsets and counts disagree */); continue; }
    const bool pamfirst = (p.right != 0) != (s != 0);
    out.hap[o] = h;
    out.pos[o] = pamfirst ? q : q + (uint32_t)p.guidelen;
    out.strand[o] = (uint8_t)s;
    out.start[o] = start;
    out.stop[o] = stop;
    out.flags[o] = has_ref ? 1 : 0;
    W2 core[4], rcore[4];
#pragma unroll
    for (int pl = 0; pl < HAWK_PLANES; ++pl) {
      W2 w = ext_lds(s_pl[pl], (int)ql - HAWK_PAD);
      w.hi &= whi;
      out.win[(size_t)pl * out.cap + o] = (uint64_t)w.lo | ((uint64_t)w.hi << 32);
      if (pl < 4) {  // the spacer+PAM core is the window without its pads
        core[pl].lo = fsh(w.lo, w.hi, HAWK_PAD) & mlo;
        core[pl].hi = (w.hi >> HAWK_PAD) & mhi;
        rcore[pl] = core[pl];
      }
    }
    double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
    if (gp.score_cfdon && has_ref) {
      if (!isref) {
        const uint32_t qr = (uint32_t)(start - ri.startp);  // REF's position map is the identity
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
          rcore[pl] = ext_glb(hs.plane[pl] + refbase, qr);
          rcore[pl].lo &= mlo; rcore[pl].hi &= mhi;
        }
      }
      bool err;
      score = cfdon_from_slices(core, rcore, s, L, cfdmask, s_cfd, err);
      if (err && gp.score_cfdon == 1) atomicExch(status, -5 /* HAWK_E_CFD; score_cfdon == 2 leaves NaN = "NA" */);
    }
    out.cfdon[o] = score;
  }
}

// REF's candidate windows as two bitmaps (bit q of strand s: REF has a guide whose window starts at q) - what "a REF
// guide shares (start, strand)" (search_guides.py:340-369) is tested against, one bit per survivor, in the count pass.
__global__ __launch_bounds__(HAWK_BLOCK) void k_ref_bits(HapSetDev hs, ScanParams p, RefInfo ri, uint32_t* __restrict__ bitsF,
                                                          uint32_t* __restrict__ bitsR) {
  const uint32_t u = blockIdx.x * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)ri.index * hs.S;
  uint32_t A[6], C[6], G[6], Tp[6];
  load6(hs.plane[0] + rowbase, u, hs.S, active, A);
  load6(hs.plane[1] + rowbase, u, hs.S, active, C);
  load6(hs.plane[2] + rowbase, u, hs.S, active, G);
  load6(hs.plane[3] + rowbase, u, hs.S, active, Tp);
  const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
  uint32_t mF[4], mR[4];
  pam_match(A, C, G, Tp, p.pam_fwd, p.pamlen, poF, mF);
  pam_match(A, C, G, Tp, p.pam_rev, p.pamlen, poR, mR);
  if (!active) return;
  const int base0 = (int)(u * 128u);
  uint4 f, r;
  f.x = mF[0] & range_mask(base0, ri.lo[0], ri.hi[0]);       r.x = mR[0] & range_mask(base0, ri.lo[1], ri.hi[1]);
  f.y = mF[1] & range_mask(base0 + 32, ri.lo[0], ri.hi[0]);  r.y = mR[1] & range_mask(base0 + 32, ri.lo[1], ri.hi[1]);
  f.z = mF[2] & range_mask(base0 + 64, ri.lo[0], ri.hi[0]);  r.z = mR[2] & range_mask(base0 + 64, ri.lo[1], ri.hi[1]);
  f.w = mF[3] & range_mask(base0 + 96, ri.lo[0], ri.hi[0]);  r.w = mR[3] & range_mask(base0 + 96, ri.lo[1], ri.hi[1]);
  *reinterpret_cast<uint4*>(bitsF + 4 * (size_t)u) = f;
  *reinterpret_cast<uint4*>(bitsR + 4 * (size_t)u) = r;
}
void hawk_launch_ref_bits(hipStream_t st, const HapSetDev& hs, const ScanParams& p, const RefInfo& ri, uint32_t* bitsF, uint32_t* bitsR) {
  const uint32_t nb = (hs.S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hipLaunchKernelGGL(k_ref_bits, dim3(nb), dim3(HAWK_BLOCK), 0, st, hs, p, ri, bitsF, bitsR);
}

void hawk_launch_search(hipStream_t st, int pass, const HapSetDev& hs, const ScanParams& p, const GuideParams& gp,
                        const RefInfo& ri, const TileMeta* tmeta, uint32_t* counts, unsigned long long* shards,
                        const uint64_t* offsets, GuideCols out, int* status, uint32_t* lists, uint32_t* big_count,
                        unsigned long long* big_list, hipEvent_t mid, uint32_t n_tiles) {
  const uint32_t ntile = n_tiles == 0xffffffffu ? hs.n_hap * p.bph : n_tiles;  // a plan view: the REF row's tiles only
  if (!ntile) { if (pass == 1 && mid) (void)hipEventRecord(mid, st); return; }
  const dim3 grid(ntile), block(HAWK_BLOCK);
  if (pass == 0) {
    hipLaunchKernelGGL(k_search_count, grid, block, 0, st, hs, p, gp, ri, tmeta, counts, shards, status, lists, big_count, big_list);
  } else {
    // lists != nullptr: small tiles are assembled from their hand-over lists, the named big ones recompute
    if (lists) hipLaunchKernelGGL(k_emit_list, grid, block, 0, st, hs, p, gp, ri, tmeta, counts, offsets, lists, out, status);
    if (mid) (void)hipEventRecord(mid, st);  // timing: k_emit_list ends here
    // list mode: a walk over the work list (<= 128 entries per REF tile, usually ~8; one per other big tile)
    const dim3 egrid(lists ? (ntile * 8u < 2048u ? ntile * 8u : 2048u) : ntile);
    hipLaunchKernelGGL(k_search_emit, egrid, block, 0, st, hs, p, gp, ri, tmeta, counts, offsets, out, status, lists, big_count, big_list);
  }
}

// ---------------------------------------------------------------------------------------
// multi-workgroup exclusive scan of u32 counts -> u64 offsets (3 small launches)
// ---------------------------------------------------------------------------------------
#define MS_TILE 1024  // entries per workgroup (4 per thread)

__global__ __launch_bounds__(HAWK_BLOCK) void k_mscan1(const uint32_t* __restrict__ counts, uint64_t n,
                                                        unsigned long long* __restrict__ partial) {
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const uint64_t i0 = (uint64_t)blockIdx.x * MS_TILE + threadIdx.x * 4;
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i0 + k < n) s += counts[i0 + k];
  s = (unsigned long long)wave_sum((uint32_t)s) + ((unsigned long long)wave_sum((uint32_t)(s >> 32)) << 32);
  if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&s_sum, s);
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_sum;
}

// single workgroup: exclusive scan of the partials in place, totals, shard sums
__global__ __launch_bounds__(1024) void k_mscan2(unsigned long long* __restrict__ partial, uint64_t nb,
                                                  const unsigned long long* __restrict__ shards, ScanTotals* totals) {
  __shared__ unsigned long long s_part[1024];
  __shared__ unsigned long long s_aux[2];
  const uint32_t t = threadIdx.x;
  if (t < 2) s_aux[t] = 0;
  unsigned long long carry = 0;
  for (uint64_t b0 = 0; b0 < nb; b0 += 1024) {
    const unsigned long long v = b0 + t < nb ? partial[b0 + t] : 0;
    s_part[t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const unsigned long long x = t >= (uint32_t)d ? s_part[t - d] : 0;
      __syncthreads();
      s_part[t] += x;
      __syncthreads();
    }
    if (b0 + t < nb) partial[b0 + t] = carry + s_part[t] - v;
    const unsigned long long tot = s_part[1023];
    __syncthreads();
    carry += tot;
  }
  if (shards && t < 256) {
    atomicAdd(&s_aux[0], shards[2 * t]);
    atomicAdd(&s_aux[1], shards[2 * t + 1]);
  }
  __syncthreads();
  if (t == 0) {
    totals->n_keep = carry;
    totals->n_keep_fwd = 0;
    totals->n_cand = s_aux[0];
    totals->n_hits = s_aux[1];
  }
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_mscan3(const uint32_t* __restrict__ counts, uint64_t n,
                                                        const unsigned long long* __restrict__ partial_off,
                                                        uint64_t* __restrict__ offsets) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  const uint64_t i0 = (uint64_t)blockIdx.x * MS_TILE + threadIdx.x * 4;
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = i0 + k < n ? counts[i0 + k] : 0u;
  uint32_t tot;
  const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(c[0] + c[1] + c[2] + c[3], s_w, &tot);
  uint64_t run = partial_off[blockIdx.x] + ex;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) offsets[i0 + k] = run;
    run += c[k];
  }
}

// k_mscan2 + k_mscan3 in one launch while the partials are few: every workgroup adds up the partials before its own
// (<= 4096 values out of L2) instead of waiting for a single-workgroup scan kernel in between; the last workgroup
// leaves the totals, workgroup 0 the shard sums
__global__ __launch_bounds__(HAWK_BLOCK) void k_mscan23(const uint32_t* __restrict__ counts, uint64_t n,
                                                         const unsigned long long* __restrict__ partial, uint32_t nb,
                                                         const unsigned long long* __restrict__ shards, uint64_t* __restrict__ offsets,
                                                         ScanTotals* __restrict__ totals) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  __shared__ unsigned long long s_sum[3];
  if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long pre = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += HAWK_BLOCK) pre += partial[j];
  pre = (unsigned long long)wave_sum((uint32_t)pre) + ((unsigned long long)wave_sum((uint32_t)(pre >> 32)) << 32);
  if ((threadIdx.x & (WAVE - 1)) == 0 && pre) atomicAdd(&s_sum[0], pre);
  if (blockIdx.x == 0 && shards) {  // 256 (candidates, hits) partial sums of the count pass
    atomicAdd(&s_sum[1], shards[2 * threadIdx.x]);
    atomicAdd(&s_sum[2], shards[2 * threadIdx.x + 1]);
  }
  const uint64_t i0 = (uint64_t)blockIdx.x * MS_TILE + threadIdx.x * 4;
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = i0 + k < n ? counts[i0 + k] : 0u;
  uint32_t tot;
  const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(c[0] + c[1] + c[2] + c[3], s_w, &tot);  // has the barrier s_sum needs
  uint64_t run = s_sum[0] + ex;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) offsets[i0 + k] = run;
    run += c[k];
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x == nb - 1) { totals->n_keep = s_sum[0] + tot; totals->n_keep_fwd = 0; }
    if (blockIdx.x == 0) { totals->n_cand = s_sum[1]; totals->n_hits = s_sum[2]; }
  }
}

// a small region (<= 2048 tiles; C1 has one): the whole scan in ONE workgroup and one launch - thread t owns a run of
// consecutive counts, the runs' sums are scanned across the workgroup.  (Runs of 19 - one rank's eighth of C3 in one
// workgroup - were tried: 25 us of dependent loads against 14 us for the two-launch scan.)
__global__ __launch_bounds__(1024) void k_mscan_one(const uint32_t* __restrict__ counts, uint32_t n, uint32_t ipt,
                                                     const unsigned long long* __restrict__ shards, uint64_t* __restrict__ offsets,
                                                     ScanTotals* __restrict__ totals) {
  __shared__ uint32_t s_w[1024 / WAVE];
  __shared__ unsigned long long s_aux[2];
  const uint32_t t = threadIdx.x;
  if (t < 2) s_aux[t] = 0;
  const uint32_t i0 = t * ipt, i1 = i0 + ipt < n ? i0 + ipt : n;
  uint32_t sum = 0;  // a tile keeps <= 65536 rows and there are <= 32768 tiles here: the total fits 32 bits
  for (uint32_t i = i0; i < i1; ++i) sum += counts[i];
  uint32_t tot;
  const uint32_t ex = block_excl_scan<1024 / WAVE>(sum, s_w, &tot);
  uint64_t run = ex;
  for (uint32_t i = i0; i < i1; ++i) { offsets[i] = run; run += counts[i]; }
  if (shards && t < 256) {
    atomicAdd(&s_aux[0], shards[2 * t]);
    atomicAdd(&s_aux[1], shards[2 * t + 1]);
  }
  __syncthreads();
  if (t == 0) {
    totals->n_keep = tot;
    totals->n_keep_fwd = 0;
    totals->n_cand = s_aux[0];
    totals->n_hits = s_aux[1];
  }
}

// The same two launches with 16 entries per thread (4096 per workgroup) for scans of millions of counts - the cluster
// instances of a plan view (hawk_csearch.hip): the single-workgroup scan of k_mscan2 in between would cost more than both.
#define MSW_IPT 16
#define MSW_TILE (HAWK_BLOCK * MSW_IPT)
__global__ __launch_bounds__(HAWK_BLOCK) void k_mscan1w(const uint32_t* __restrict__ counts, uint64_t n, unsigned long long* __restrict__ partial) {
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const uint64_t i0 = (uint64_t)blockIdx.x * MSW_TILE + threadIdx.x * MSW_IPT;
  unsigned long long s = 0;
  if (i0 + MSW_IPT <= n) {
#pragma unroll
    for (int k = 0; k < MSW_IPT / 4; ++k) { const uint4 v = *reinterpret_cast<const uint4*>(counts + i0 + 4 * k); s += (unsigned long long)v.x + v.y + v.z + v.w; }
  } else {
    for (int k = 0; k < MSW_IPT; ++k) if (i0 + k < n) s += counts[i0 + k];
  }
  s = (unsigned long long)wave_sum((uint32_t)s) + ((unsigned long long)wave_sum((uint32_t)(s >> 32)) << 32);
  if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&s_sum, s);
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_sum;
}
__global__ __launch_bounds__(HAWK_BLOCK) void k_mscan23w(const uint32_t* __restrict__ counts, uint64_t n, const unsigned long long* __restrict__ partial,
                                                          uint32_t nb, const unsigned long long* __restrict__ shards, uint64_t* __restrict__ offsets,
                                                          ScanTotals* __restrict__ totals) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  __shared__ unsigned long long s_sum[3];
  if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long pre = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += HAWK_BLOCK) pre += partial[j];
  pre = (unsigned long long)wave_sum((uint32_t)pre) + ((unsigned long long)wave_sum((uint32_t)(pre >> 32)) << 32);
  if ((threadIdx.x & (WAVE - 1)) == 0 && pre) atomicAdd(&s_sum[0], pre);
  if (blockIdx.x == 0 && shards) {
    atomicAdd(&s_sum[1], shards[2 * threadIdx.x]);
    atomicAdd(&s_sum[2], shards[2 * threadIdx.x + 1]);
  }
  const uint64_t i0 = (uint64_t)blockIdx.x * MSW_TILE + threadIdx.x * MSW_IPT;
  uint32_t c[MSW_IPT];
  uint32_t sum = 0;
  if (i0 + MSW_IPT <= n) {
#pragma unroll
    for (int k = 0; k < MSW_IPT / 4; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(counts + i0 + 4 * k);
      c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < MSW_IPT; ++k) c[k] = i0 + k < n ? counts[i0 + k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < MSW_IPT; ++k) sum += c[k];
  uint32_t tot;
  const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(sum, s_w, &tot);  // has the barrier s_sum needs
  uint64_t run = s_sum[0] + ex;
  if (i0 + MSW_IPT <= n) {
#pragma unroll
    for (int k = 0; k < MSW_IPT; k += 2) {
      const uint64_t a = run, b = run + c[k];
      *reinterpret_cast<ulonglong2*>(offsets + i0 + k) = make_ulonglong2(a, b);
      run = b + c[k + 1];
    }
  } else {
#pragma unroll
    for (int k = 0; k < MSW_IPT; ++k) {
      if (i0 + k < n) offsets[i0 + k] = run;
      run += c[k];
    }
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x == nb - 1) { totals->n_keep = s_sum[0] + tot; totals->n_keep_fwd = 0; }
    if (blockIdx.x == 0) { totals->n_cand = s_sum[1]; totals->n_hits = s_sum[2]; }
  }
}

void hawk_launch_mscan(hipStream_t st, const uint32_t* counts, uint64_t n, unsigned long long* partial,
                       const unsigned long long* shards, uint64_t* offsets, ScanTotals* totals) {
  if (n <= 2048 && shards) {
    const uint32_t ipt = (uint32_t)((n + 1023) / 1024);
    hipLaunchKernelGGL(k_mscan_one, dim3(1), dim3(1024), 0, st, counts, (uint32_t)n, ipt ? ipt : 1u, shards, offsets, totals);
    return;
  }
  const uint32_t nb = (uint32_t)((n + MS_TILE - 1) / MS_TILE);
  const uint32_t nbw = (uint32_t)((n + MSW_TILE - 1) / MSW_TILE);
  if (nb > 4096 && nbw <= 4096 && shards && (reinterpret_cast<uintptr_t>(counts) & 15) == 0 && (reinterpret_cast<uintptr_t>(offsets) & 15) == 0) {
    hipLaunchKernelGGL(k_mscan1w, dim3(nbw), dim3(HAWK_BLOCK), 0, st, counts, n, partial);
    hipLaunchKernelGGL(k_mscan23w, dim3(nbw), dim3(HAWK_BLOCK), 0, st, counts, n, partial, nbw, shards, offsets, totals);
    return;
  }
  hipLaunchKernelGGL(k_mscan1, dim3(nb), dim3(HAWK_BLOCK), 0, st, counts, n, partial);
  if (nb <= 4096 && shards) {
    hipLaunchKernelGGL(k_mscan23, dim3(nb), dim3(HAWK_BLOCK), 0, st, counts, n, partial, nb, shards, offsets, totals);
    return;
  }
  hipLaunchKernelGGL(k_mscan2, dim3(1), dim3(1024), 0, st, partial, (uint64_t)nb, shards, totals);
  hipLaunchKernelGGL(k_mscan3, dim3(nb), dim3(HAWK_BLOCK), 0, st, counts, n, partial, offsets);
}
