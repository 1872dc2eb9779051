// hawk_vsearch.hip — the guide search straight from an expansion plan: no haplotype plane is ever written.
//
// A haplotype row of a plan is REF + the variants the chromosome copy carries.  Two facts of the reference carry the design:
//   * search_guides.py:468-471 drops every window of a non-REF haplotype whose spacer+PAM holds no variant base, so a row
//     contributes guide rows only where a window [q, q + L) touches an alt allele: the "dirty" window starts.  They lie in
//     the few 32-position words around the row's variants (C3: ~11 % of all words);
//   * everywhere else the haplotype is a verbatim copy of a REF stretch, so its PAM hits - which only the job's totals
//     (candidates, hits) still need - are REF's PAM hits under the stretch's shift: one difference of prefix counts over
//     REF's hit bitmap per clean run, instead of a PAM match per position.
// Per (row, tile of 32 768 positions) one workgroup: the tile's records are staged in LDS (as k_hx_build does), every record
// marks the words its alleles dirty in a 1024-bit map, and each dirty word becomes one thread's task - 96 bits of the five
// planes around the word are assembled in registers by the expansion's own word builder (hx_words_t, REF read from L2),
// PAM-matched on both strands, range- and variant-filtered.  Survivors are then spread over the threads exactly as in
// hawk_search.hip (survivor number i of the tile: strand 0 in position order, then strand 1) and classified against REF
// (remove_redundant_guides, search_guides.py:340-369); PASS 0 counts rows per tile, PASS 1 - after the offset scan -
// repeats the work and writes the rows: coordinates, packed windows, CFDon.  Table, totals and row order are those of
// hawk_search on the materialised planes (tests/test_gpu_vsearch.py holds them equal column for column).
//
// The REF row itself (every candidate is a row) goes through the plane kernels of hawk_search.hip on the plan's REF planes.
#include "hawk_vc.h"

#define VC_BLOCK 128         // threads per workgroup: a C3 tile has ~110 dirty words
#define VC_MAXV 128          // records staged per tile (32 B each)
#define VC_SLOTS VC_BLOCK    // dirty words per chunk: one per thread
#define VC_BACK (HAWK_PAD + 1)
#define VC_REACH 64          // a word's string reads <= 32 + L + PAD - 1 < 96 positions from the word's start - PAD

template <int PASS>
__global__ __launch_bounds__(VC_BLOCK) void k_vsearch(HapSetDev hs, VcArgs va, ScanParams p, GuideParams gp, RefInfo ri,
                                                       const TileMeta* __restrict__ tmeta, uint32_t* __restrict__ counts,
                                                       uint32_t* __restrict__ counts0, unsigned long long* __restrict__ shards,
                                                       const uint64_t* __restrict__ offsets, GuideCols out, int* status, uint32_t tile0) {
  __shared__ HxVar s_v[VC_MAXV];
  __shared__ uint32_t s_str[5][3][VC_SLOTS];  // the slots' 96-bit strings: position 32 w - PAD + i at bit i
  __shared__ uint32_t s_kw[2][VC_SLOTS];      // kept window starts of the slot's word, per strand
  __shared__ uint32_t s_ex[VC_SLOTS];         // packed exclusive survivor offsets (strand 0 | strand 1 << 16)
  __shared__ uint32_t s_wl[VC_SLOTS];         // the slot's word within the tile
  __shared__ uint32_t s_bm[32], s_bmpre[33];  // the tile's dirty words as a bitmap, set bits in the words before
  __shared__ uint16_t s_words[HX_TW];         // ... and as a list
  __shared__ uint32_t s_segrel[NSEG + 1];
  __shared__ int64_t s_seggen[NSEG];
  __shared__ double s_cfd[PASS == 1 ? 336 : 1];
  __shared__ uint32_t s_w[VC_BLOCK / WAVE];
  __shared__ uint32_t s_red[VC_BLOCK / WAVE][4];
  __shared__ uint32_t s_n;
  const uint32_t tid = threadIdx.x;
  const HxVar* __restrict__ vrecs = static_cast<const HxVar*>(va.recs_);
  const HxTile* __restrict__ vtiles = static_cast<const HxTile*>(va.tiles_);
  uint32_t tile = tile0 + blockIdx.x;
  if (PASS == 1) {  // XCD-aware: every XCD takes a contiguous run of tiles (rows of neighbouring tiles share cache lines)
    const uint32_t per = gridDim.x >> 3;
    if (blockIdx.x < per * 8u) tile = tile0 + (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  }
  const uint32_t h = tile / p.bph, blk = tile - h * p.bph;
  const TileMeta tm = tmeta[tile];
  const int32_t haplen = (int32_t)tm.hap_len;
  const int32_t p_lo = (int32_t)(blk * (uint32_t)(HX_TW * 32));
  uint32_t n_rows_tile = 0, n_rows0 = 0;  // PASS 1: the tile's rows, those of strand 0 (they come first)
  if (PASS == 1) {
    n_rows_tile = counts[tile];
    if (n_rows_tile == 0) return;  // workgroup-uniform
    n_rows0 = counts0[tile];
  }
  if (p_lo >= haplen || tm.scan_stop <= tm.scan_start) {  // the row ends before this tile, or scans nothing (a row collapsed onto another)
    if (PASS == 0 && tid == 0) { counts[tile] = 0; counts0[tile] = 0; }
    return;
  }
  const HxTile ht = vtiles[tile];
  const int32_t p_hi = p_lo + HX_TW * 32;
  const int nwt = (haplen - p_lo + 31) / 32 < HX_TW ? (haplen - p_lo + 31) / 32 : HX_TW;  // words of the tile that exist
  // The tile index names the last record starting at or before the tile.  A word's string starts PAD positions in front
  // of the word, so the records of those positions count too: at most PAD of them plus the one in force before them -
  // staging starts VC_BACK records earlier.  The first staged record then either is the row's first or starts in front of
  // every string of the tile, which is what the builders' search (from local index 0) relies on.
  const HxVar* first = vrecs + (((uint64_t)ht.first_hi << 32) | ht.first_lo);
  {
    const uint64_t back = (uint64_t)(first - (vrecs + va.hv_off[h]));
    first -= back < VC_BACK ? back : VC_BACK;
  }
  const int avail = (int)(vrecs + va.hv_off[h + 1] - first);  // the row's records from `first` on
  const int L = p.L;
  if (tid == 0) s_n = 0;
  if (tid < 32) s_bm[tid] = 0;
  __syncthreads();
  // ---- the tile's records (everything a word's string can reach: VC_REACH positions past the tile) and its position-map slice
  if (tid < VC_MAXV && (int)tid < avail) {
    const HxVar v = first[tid];
    s_v[tid] = v;
    if (v.o < p_hi + VC_REACH) atomicAdd(&s_n, 1u);
  }
  const uint32_t tile_q0 = (uint32_t)p_lo;
  const uint32_t tile_end = tile_q0 + HX_TW * 32u + 64u;
  const uint32_t k0 = tm.seg0, kend = tm.seg_end;
  if (tid < NSEG) {
    const uint32_t k = k0 + tid;
    uint32_t seg_r = 0xffffffffu;
    int64_t seg_g = 0;
    if (k < kend) {
      seg_r = hs.seg_rel[k];
      seg_g = hs.seg_gen[k];
      if (!(tid == 0 || seg_r < tile_end)) seg_r = 0xffffffffu;
    }
    s_segrel[tid] = seg_r;
    s_seggen[tid] = seg_g;
  }
  if (tid == NSEG) s_segrel[NSEG] = 0xffffffffu;
  const bool ovf = k0 + NSEG < kend && hs.seg_rel[k0 + NSEG] < tile_end;  // rare: > NSEG segments in a tile
  if (PASS == 1 && gp.score_cfdon) for (uint32_t i = tid; i < 336; i += VC_BLOCK) s_cfd[i] = gp.cfd_mm[i];
  __syncthreads();
  const int n = (int)s_n;                                 // staged records within reach (a prefix: records are sorted)
  const bool all = n < VC_MAXV || avail <= VC_MAXV;       // nothing within reach lies beyond the staged ones
  // ---- dirty words: a record dirties the window starts [o - L + 1, o + alt_len - 1]
  {
    const int jn = all ? n : avail;
    for (int j = (int)tid; j < jn; j += VC_BLOCK) {
      int32_t o; uint32_t al;
      if (j < n) { o = s_v[j].o; al = s_v[j].alt_len; } else { const HxVar v = first[j]; o = v.o; al = v.alt_len; }
      if (o >= p_hi + L) break;  // sorted: nothing further reaches back into the tile
      int32_t lo = o - L + 1 - p_lo, hi = o + (int32_t)al - 1 - p_lo;
      lo = lo < 0 ? 0 : lo;
      hi = hi > 32 * nwt - 1 ? 32 * nwt - 1 : hi;
      if (hi < lo) continue;
      const int wlo = lo >> 5, whi = hi >> 5;
      for (int bw = wlo >> 5; bw <= (whi >> 5); ++bw) {
        const int a = wlo > bw * 32 ? wlo - bw * 32 : 0, b = whi < bw * 32 + 31 ? whi - bw * 32 : 31;
        const uint32_t m = (b == 31 ? 0xffffffffu : ((1u << (b + 1)) - 1u)) & ~((1u << a) - 1u);
        atomicOr(&s_bm[bw], m);
      }
    }
  }
  __syncthreads();
  if (tid < WAVE) {  // prefix popcounts of the 32 bitmap words (wave 0)
    const uint32_t c = tid < 32 ? (uint32_t)__popc(s_bm[tid]) : 0u;
    const uint32_t inc = wave_incl_scan(c);
    if (tid < 32) s_bmpre[tid] = inc - c;
    if (tid == 31) s_bmpre[32] = inc;
  }
  __syncthreads();
  if (tid < 32) {  // the dirty words as a list: task g is word s_words[g]
    uint32_t m = s_bm[tid], o = s_bmpre[tid];
    while (m) { s_words[o++] = (uint16_t)(tid * 32u + (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
  }
  __syncthreads();
  const uint32_t n_tasks = s_bmpre[32];
  const uint32_t n_chunks = (n_tasks + VC_SLOTS - 1) / VC_SLOTS;

  // ranges on the window start q (search_guides.py:49-84, 395-420): as phase A of hawk_search.hip
  VcRanges rg;
  {
    const int ss = tm.scan_start, se = tm.scan_stop;
    const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
    const int qmin = HAWK_PAD, qmax = haplen - L - HAWK_PAD + 1;
    rg.slo[0] = ss - poF; rg.shi[0] = se - poF; rg.slo[1] = ss - poR; rg.shi[1] = se - poR;
    rg.lo[0] = rg.slo[0] > qmin ? rg.slo[0] : qmin; rg.hi[0] = rg.shi[0] < qmax ? rg.shi[0] : qmax;
    rg.lo[1] = rg.slo[1] > qmin ? rg.slo[1] : qmin; rg.hi[1] = rg.shi[1] < qmax ? rg.shi[1] : qmax;
  }
  const int poF = p.right ? 0 : p.guidelen, poR = p.right ? p.guidelen : 0;
  const uint32_t mlo = L >= 32 ? 0xffffffffu : ((1u << L) - 1u), mhi = L <= 32 ? 0u : ((1u << (L - 32)) - 1u);
  const int W = L + 2 * HAWK_PAD;
  const uint32_t whi = W >= 64 ? 0xffffffffu : ((1u << (W - 32)) - 1u);  // W = L + 20 > 32
  const int ncfd = gp.guidelen < 20 ? gp.guidelen : 20;
  const uint32_t cfdmask = (1u << ncfd) - 1u;
  // the REF planes for the general word builder; reads of bits no word keeps are clamped
  auto ref32 = [&](int pl, uint32_t r) -> uint32_t {
    const uint32_t w = (r >> 5) < va.ref_S - 2 ? (r >> 5) : va.ref_S - 2;
    return ext32_glb(va.ref[pl], (w << 5) | (r & 31u));
  };
  // REF shift of the copied stretch that holds position pa (pa is not inside an alt allele): rs - (o + alt_len) of the last
  // record starting at or before pa
  auto shift_at = [&](int32_t pa) -> int32_t {
    int a = 0, b = all ? n : avail;
    while (a < b) {
      const int m = (a + b) >> 1;
      const int32_t o = m < n ? s_v[m].o : first[m].o;
      if (o <= pa) a = m + 1; else b = m;
    }
    const int k = a - 1;
    if (k < 0) return 0;
    const HxVar v = k < n ? s_v[k] : first[k];
    return (int32_t)v.rs - (v.o + (int32_t)v.alt_len);
  };
  // first dirty word of the tile behind word wl (HX_TW: none)
  auto next_dirty = [&](int wl) -> int {
    const int w1 = wl + 1;
    if (w1 >= HX_TW) return HX_TW;
    int bw = w1 >> 5;
    uint32_t m = s_bm[bw] & (0xffffffffu << (w1 & 31));
    while (!m && ++bw < 32) m = s_bm[bw];
    return m ? bw * 32 + __builtin_ctz(m) : HX_TW;
  };

  uint32_t cand = 0, hits = 0, nvalid0 = 0, nvalid1 = 0;
  // rows of a tile: strand 0 in position order, then strand 1 - the strand-0 rows of ALL chunks come first, so the emit pass
  // takes the tile's strand-0 total from the count pass and fills both runs in one sweep over the chunks
  uint64_t row0 = 0, row1 = 0;
  if (PASS == 1) { row0 = offsets[tile]; row1 = row0 + n_rows0; }
  if (PASS == 0 && tid == 0) {  // the clean run in front of the tile's first dirty word (the whole tile when it has none)
    const int nd = next_dirty(-1);
    const int32_t pb = p_lo + 32 * (nd < nwt ? nd : nwt);
    if (pb > p_lo) vc_count_run(va, rg, p_lo, pb, shift_at(p_lo), cand, hits);
  }

#pragma unroll 1
  for (uint32_t chunk = 0; chunk < n_chunks; ++chunk) {
    // ---- phase A: one dirty word per thread ---------------------------------------------------
    const uint32_t g = chunk * VC_SLOTS + tid;
    const bool active = g < n_tasks;
    uint32_t kF = 0, kR = 0, wl = 0;
    if (active) {
      wl = s_words[g];
      const int32_t q0 = p_lo + 32 * (int32_t)wl;  // row position of the word
      uint32_t X[5][3];
      const int32_t p0 = q0 - HAWK_PAD;
      const int32_t p0c = p0 < 0 ? 0 : p0;
      int32_t shift_run = 0;
      bool have_shift = true;
      if (!(all && vc_string_fast(va, s_v, n, p0c, haplen, q0 + 32, X, shift_run))) {
        have_shift = false;
        if (all) hx_words_t<true, 3>(va.alt_codes, s_v, first, n, n, false, p0c, haplen, ref32, X[0], X[1], X[2], X[3], X[4]);
        else hx_words_t<false, 3>(va.alt_codes, s_v, first, n, avail, false, p0c, haplen, ref32, X[0], X[1], X[2], X[3], X[4]);
      }
      if (p0 < 0) {  // the row's first word: there is nothing in front of position 0
#pragma unroll
        for (int pl = 0; pl < 5; ++pl) {
          X[pl][2] = fsh(X[pl][1], X[pl][2], 32 - HAWK_PAD);
          X[pl][1] = fsh(X[pl][0], X[pl][1], 32 - HAWK_PAD);
          X[pl][0] = X[pl][0] << HAWK_PAD;
        }
      }
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) { s_str[pl][0][tid] = X[pl][0]; s_str[pl][1][tid] = X[pl][1]; s_str[pl][2][tid] = X[pl][2]; }
      // E: window starts whose spacer+PAM holds a variant base (search_guides.py:468-471): sliding OR of V over L bits
      uint32_t v0 = X[4][0], v1 = X[4][1], v2 = X[4][2];
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
      uint32_t f = pam_match96(X, p.pam_fwd, p.pamlen, HAWK_PAD + poF);
      uint32_t rv = pam_match96(X, p.pam_rev, p.pamlen, HAWK_PAD + poR);
      f &= range_mask(q0, rg.slo[0], rg.shi[0]);
      rv &= range_mask(q0, rg.slo[1], rg.shi[1]);
      if (PASS == 0) hits += __popc(f) + __popc(rv);
      f &= range_mask(q0, rg.lo[0], rg.hi[0]);
      rv &= range_mask(q0, rg.lo[1], rg.hi[1]);
      if (PASS == 0) cand += __popc(f) + __popc(rv);
      kF = f & E;
      kR = rv & E;
      if (PASS == 0) {  // the clean run behind the word, up to the next dirty word (or the end of the tile)
        const int nd = next_dirty((int)wl);
        const int32_t pa = q0 + 32, pb = p_lo + 32 * (nd < nwt ? nd : nwt);
        if (pb > pa) vc_count_run(va, rg, pa, pb, have_shift ? shift_run : shift_at(pa), cand, hits);
      }
    }
    s_kw[0][tid] = kF; s_kw[1][tid] = kR; s_wl[tid] = wl;
    // ---- phase B: one scan for both strands (per-slot counts <= 32, chunk totals <= 4096) -----
    const uint32_t cF = (uint32_t)__popc(kF), cR = (uint32_t)__popc(kR);
    uint32_t TT;
    const uint32_t exFR = block_excl_scan<VC_BLOCK / WAVE>(cF | (cR << 16), s_w, &TT);  // its barriers publish the slots
    s_ex[tid] = exFR;
    __syncthreads();
    const uint32_t TF = TT & 0xffffu, TR = TT >> 16, T = TF + TR;
    // ---- phase C: survivors number i and i + VC_BLOCK of the chunk per thread (strand 0 in position order, then strand 1).
    // Two at a time so that their L2 gathers (REF's candidate bit, REF's core) are in flight together: the kernel is bound
    // by round trips per resident wave, not by bytes or instructions.
#pragma unroll 1
    for (uint32_t base = 0; base < T; base += 2 * VC_BLOCK) {
      uint32_t valid[2] = {0, 0}, sst[2] = {0, 0}, qq[2] = {0, 0}, qrr[2] = {0, 0};
      int64_t startv[2] = {0, 0};
      bool inrv[2] = {false, false}, has_refv[2] = {false, false};
      W2 win[2][5], core[2][4], rcore[2][4];
      uint32_t rw[2] = {0, 0};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int pl = 0; pl < 5; ++pl) win[u][pl] = W2{0, 0};
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) { core[u][pl] = W2{0, 0}; rcore[u][pl] = W2{0, 0}; }
      }
      // LDS work of both survivors, then every gather of both
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t i = base + (uint32_t)u * VC_BLOCK + tid;
        if (((base + (uint32_t)u * VC_BLOCK + tid) & ~(uint32_t)(WAVE - 1)) < T) {  // wave-uniform: a wave wholly past the last survivor skips
          const uint32_t gsv = i < T ? i : 0u;  // lanes past the end redo survivor 0 and are masked below
          const uint32_t s = gsv >= TF ? 1u : 0u;
          const uint32_t target = s ? gsv - TF : gsv;
          const uint32_t sh16 = s ? 16u : 0u;
          uint32_t slot = 0;
#pragma unroll
          for (uint32_t step = VC_SLOTS / 2; step; step >>= 1) slot += (((s_ex[slot + step] >> sh16) & 0xffffu) <= target) ? step : 0u;
          const uint32_t bpos = select_bit(s_kw[s][slot], target - ((s_ex[slot] >> sh16) & 0xffffu));
          const uint32_t q = tile_q0 + 32u * s_wl[slot] + bpos;
          int64_t start;
          if (ovf) {
            start = posmap_global(hs, h, q);
          } else {
            uint32_t sj = 0;
#pragma unroll
            for (uint32_t step = NSEG / 2; step; step >>= 1) sj += (s_segrel[sj + step] <= q) ? step : 0u;
            start = s_seggen[sj] + (int64_t)(q - s_segrel[sj]);
          }
          // the padded window [q - PAD, q + L + PAD) sits at bits [bpos, bpos + W) of the slot's string
#pragma unroll
          for (int pl = 0; pl < 5; ++pl) {
            win[u][pl] = ext96(s_str[pl][0][slot], s_str[pl][1][slot], s_str[pl][2][slot], bpos);
            win[u][pl].hi &= whi;
            if (pl < 4) {
              core[u][pl].lo = fsh(win[u][pl].lo, win[u][pl].hi, HAWK_PAD) & mlo;
              core[u][pl].hi = (win[u][pl].hi >> HAWK_PAD) & mhi;
            }
          }
          // a REF guide shares (start, strand) iff REF has a candidate window starting at qr = start - startp (k_ref_bits);
          // the row is redundant iff the four code planes agree as well (search_guides.py:340-369).  REF's core is fetched
          // whether or not the bit turns out set: one round trip instead of two.
          const int64_t qr64 = start - ri.startp;
          const bool inr = qr64 >= 0 && qr64 < (int64_t)ri.n_bits;
          const uint32_t qr = inr ? (uint32_t)qr64 : 0u;
          rw[u] = (s ? ri.bits[1] : ri.bits[0])[qr >> 5];
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) rcore[u][pl] = ext_glb(va.ref[pl], qr);
          sst[u] = s; qq[u] = q; qrr[u] = qr; startv[u] = start; inrv[u] = inr;
          valid[u] = i < T ? 1u : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool has_ref = inrv[u] && ((rw[u] >> (qrr[u] & 31u)) & 1u);
        bool same = has_ref;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
          rcore[u][pl].lo &= mlo; rcore[u][pl].hi &= mhi;
          same = same && rcore[u][pl].lo == core[u][pl].lo && rcore[u][pl].hi == core[u][pl].hi;
          if (!has_ref) rcore[u][pl] = core[u][pl];
        }
        has_refv[u] = has_ref;
        if (same) valid[u] = 0;
      }
      if (PASS == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          nvalid0 += (valid[u] && !sst[u]) ? 1u : 0u;
          nvalid1 += (valid[u] && sst[u]) ? 1u : 0u;
        }
      } else {
        // ranks of the rows among the chunk's rows of their strand: one scan over four 8-bit fields (<= 128 per field)
        const uint32_t f = ((valid[0] && !sst[0]) ? 1u : 0u) | ((valid[0] && sst[0]) ? 1u << 8 : 0u) |
                           ((valid[1] && !sst[1]) ? 1u << 16 : 0u) | ((valid[1] && sst[1]) ? 1u << 24 : 0u);
        uint32_t tot;
        const uint32_t ex = block_excl_scan<VC_BLOCK / WAVE>(f, s_w, &tot);
        const uint32_t t0F = tot & 0xffu, t0R = (tot >> 8) & 0xffu, t1F = (tot >> 16) & 0xffu, t1R = tot >> 24;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (!valid[u]) continue;
          const uint32_t s = sst[u], q = qq[u];
          const uint32_t rk = u == 0 ? (s ? (ex >> 8) & 0xffu : ex & 0xffu) : (s ? t0R + (ex >> 24) : t0F + ((ex >> 16) & 0xffu));
          const uint64_t o = (s ? row1 : row0) + rk;
          if (o >= out.cap) {
            atomicExch(status, -3 /* HAWK_E_CAPACITY: offsets and counts disagree */);
            continue;
          }
          int64_t stop;  // search_guides.py:260-280: stop = posmap[q + L]
          if (ovf) {
            stop = posmap_global(hs, h, q + (uint32_t)L);
          } else {
            uint32_t sj = 0;
#pragma unroll
            for (uint32_t step = NSEG / 2; step; step >>= 1) sj += (s_segrel[sj + step] <= q + (uint32_t)L) ? step : 0u;
            stop = s_seggen[sj] + (int64_t)(q + (uint32_t)L - s_segrel[sj]);
          }
          const bool pamfirst = (p.right != 0) != (s != 0);
          out.hap[o] = h;
          out.pos[o] = pamfirst ? q : q + (uint32_t)p.guidelen;
          out.strand[o] = (uint8_t)s;
          out.start[o] = startv[u];
          out.stop[o] = stop;
          out.flags[o] = has_refv[u] ? 1 : 0;
#pragma unroll
          for (int pl = 0; pl < HAWK_PLANES; ++pl) out.win[(size_t)pl * out.cap + o] = (uint64_t)win[u][pl].lo | ((uint64_t)win[u][pl].hi << 32);
          double score = __longlong_as_double(0x7ff8000000000000ll);  // NaN -> "NA"
          if (gp.score_cfdon && has_refv[u]) {
            bool err;
            score = cfdon_from_slices(core[u], rcore[u], s, L, cfdmask, s_cfd, err);
            if (err && gp.score_cfdon == 1) atomicExch(status, -5 /* HAWK_E_CFD; score_cfdon == 2 leaves NaN = "NA" */);
          }
          out.cfdon[o] = score;
        }
        row0 += t0F + t1F; row1 += t0R + t1R;
      }
    }
    __syncthreads();  // the slots are rewritten by the next chunk
  }
  if (PASS == 0) {
    const uint32_t a0 = wave_sum(nvalid0), a1 = wave_sum(cand), a2 = wave_sum(hits), a3 = wave_sum(nvalid1);
    if ((tid & (WAVE - 1)) == 0) { s_red[tid / WAVE][0] = a0; s_red[tid / WAVE][1] = a1; s_red[tid / WAVE][2] = a2; s_red[tid / WAVE][3] = a3; }
    __syncthreads();
    if (tid == 0) {
      uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
      for (int wv = 0; wv < VC_BLOCK / WAVE; ++wv) { t0 += s_red[wv][0]; t1 += s_red[wv][1]; t2 += s_red[wv][2]; t3 += s_red[wv][3]; }
      counts[tile] = t0 + t3;
      counts0[tile] = t0;
      if (t1 | t2) {
        atomicAdd(&shards[(tile & 255u) * 2 + 0], (unsigned long long)t1);
        atomicAdd(&shards[(tile & 255u) * 2 + 1], (unsigned long long)t2);
      }
    }
  }
}

void hawk_launch_vsearch(hipStream_t st, int pass, const HapSetDev& hs, const VcArgs& va, const ScanParams& p, const GuideParams& gp,
                         const RefInfo& ri, const TileMeta* tmeta, uint32_t* counts, uint32_t* counts0, unsigned long long* shards,
                         const uint64_t* offsets, GuideCols out, int* status, uint32_t tile0, uint32_t n_tiles) {
  if (!n_tiles) return;
  if (pass == 0)
    hipLaunchKernelGGL(k_vsearch<0>, dim3(n_tiles), dim3(VC_BLOCK), 0, st, hs, va, p, gp, ri, tmeta, counts, counts0, shards, offsets, out, status, tile0);
  else
    hipLaunchKernelGGL(k_vsearch<1>, dim3(n_tiles), dim3(VC_BLOCK), 0, st, hs, va, p, gp, ri, tmeta, counts, counts0, shards, offsets, out, status, tile0);
}


// ---------------------------------------------------------------------------------------
// REF's PAM hits per strand, indexed by window start and NOT cut to any range (clean stretches of the haplotypes read them
// under their own ranges): hp[w] = {hit bits of word w on strand 0, hits in the words before, the same for strand 1}, w <= S.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(HAWK_BLOCK) void k_ref_hits(HapSetDev hs, ScanParams p, int32_t ref_index, uint4* __restrict__ hp) {
  const uint32_t u = blockIdx.x * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)ref_index * hs.S;
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
#pragma unroll
  for (int k = 0; k < 4; ++k) hp[4 * (size_t)u + k] = make_uint4(mF[k], 0u, mR[k], 0u);
}
// single workgroup: the prefix counts behind the hit words (S is a few 10^4 words: one pass of 1024-entry scans)
__global__ __launch_bounds__(1024) void k_ref_hits_prefix(uint32_t S, uint4* __restrict__ hp) {
  __shared__ uint32_t s_w[1024 / WAVE];
  uint32_t carryF = 0, carryR = 0;
  for (uint32_t b0 = 0; b0 <= S; b0 += 1024) {  // entry S closes the table: hit word 0, count = all hits
    const uint32_t w = b0 + threadIdx.x;
    uint32_t xF = 0, xR = 0;
    if (w < S) { const uint4 e = hp[w]; xF = e.x; xR = e.z; }
    const uint32_t c = (uint32_t)__popc(xF) | ((uint32_t)__popc(xR) << 16);  // <= 32 each, 1024 entries: sums < 2^16
    uint32_t tot;
    const uint32_t ex = block_excl_scan<1024 / WAVE>(c, s_w, &tot);
    if (w <= S) hp[w] = make_uint4(xF, carryF + (ex & 0xffffu), xR, carryR + (ex >> 16));
    carryF += tot & 0xffffu; carryR += tot >> 16;
  }
}
void hawk_launch_ref_hits(hipStream_t st, const HapSetDev& hs, const ScanParams& p, int32_t ref_index, void* hp) {
  const uint32_t nb = (hs.S / 4 + HAWK_BLOCK - 1) / HAWK_BLOCK;
  hipLaunchKernelGGL(k_ref_hits, dim3(nb), dim3(HAWK_BLOCK), 0, st, hs, p, ref_index, (uint4*)hp);
  hipLaunchKernelGGL(k_ref_hits_prefix, dim3(1), dim3(1024), 0, st, hs.S, (uint4*)hp);
}
