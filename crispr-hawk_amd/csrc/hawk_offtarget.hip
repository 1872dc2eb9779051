// hawk_offtarget.hip — K7: mismatch-tolerant off-target enumeration on the GPU.
//
// Replaces the reference's external call `crispritz.py search <index> pam.txt guides.txt -mm M
// -bDNA 0 -bRNA 0` (offtargets.py:222-293; CRISPRitz 2.6.6 is not vendored, so the semantics
// are restated from the call site and from the fields the reference consumes, offtarget.py:
// 77-101): every genome window, both strands, whose PAM positions hold an unambiguous base
// inside the PAM's IUPAC set and whose spacer differs from a guide in at most `max_mm`
// positions (an ambiguous genome base counts as a mismatch).
//
// The genome is a plane set like the haplotypes (contigs cut into equal rows on the host),
// made one-hot: A,C,G,T planes keep only unambiguous bases, plane 4 marks ambiguous ones.
//   k_ot_onehot   planes -> one-hot + ambiguity mask (once per genome)
//   k_scan_raw    (hawk_kernels.hip) PAM hits of both strands as bitmaps, indexed by window start
//   k_ot_sites    hit bits -> site records: the guidelen+pamlen window in guide orientation
//                 (strand 1 reverse-complemented) as a 2-bit code + ambiguity mask
//   k_ot_match    all pairs site x guide: XOR, fold to one bit per base, popcount, <= max_mm
//                 survivors appended through a wave-aggregated atomic (small guide sets)
//   k_ot_match_seeded  pigeonhole filter: with the spacer cut into max_mm + 1 blocks, a pair within max_mm
//                 mismatches agrees exactly in one block; the guides are bucketed per block by the block's
//                 bases (host, counting sort), a site compares only against the buckets its own blocks name:
//                 ~ (max_mm + 1) / 4^blocklen of the all-pairs work
//   k_ot_gather   hit sites -> compact array for one download
#include "hawk_bits.h"

// reverse the low L (<= 32) bits
__device__ __forceinline__ uint32_t rev32(uint32_t v, int L) { return __brev(v) >> (32 - L); }
// spread the 32 bits of x to the even bit positions of a 64-bit word
__device__ __forceinline__ uint64_t spread(uint32_t x) {
  uint64_t v = x;
  v = (v | (v << 16)) & 0x0000ffff0000ffffull;
  v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
  v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_onehot(uint32_t* pA, uint32_t* pC, uint32_t* pG, uint32_t* pT, uint32_t* pV,
                                                           uint64_t nwords) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= nwords) return;
  const uint32_t a = pA[i], c = pC[i], g = pG[i], t = pT[i];
  const uint32_t multi = (a & c) | ((a | c) & (g | t)) | (g & t);  // two or more planes set: N, R, Y, ...
  pA[i] = a & ~multi; pC[i] = c & ~multi; pG[i] = g & ~multi; pT[i] = t & ~multi;
  pV[i] = multi;
}
void hawk_launch_ot_onehot(hipStream_t st, uint32_t* const* plane, uint64_t nwords) {
  if (!nwords) return;
  hipLaunchKernelGGL(k_ot_onehot, dim3((uint32_t)((nwords + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, plane[0],
                     plane[1], plane[2], plane[3], plane[4], nwords);
}

// one workgroup per 1024-word tile of a genome row, as k_emit_hits: bit q of keepF/keepR = a
// window starting at q has its PAM (strand 0: fwd pattern, strand 1: reverse complement).
__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_sites(HapSetDev hs, ScanParams p, const uint32_t* __restrict__ keepF,
                                                          const uint32_t* __restrict__ keepR,
                                                          const uint64_t* __restrict__ offsets, OtSite* __restrict__ sites) {
  __shared__ uint32_t s_w[HAWK_BLOCK / WAVE];
  const uint32_t h = blockIdx.x / p.bph, blk = blockIdx.x % p.bph;
  const uint32_t u = blk * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
  const int L = p.L;
  const uint32_t lmask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint32_t* keep = s ? keepR : keepF;
    uint4 kw = make_uint4(0, 0, 0, 0);
    if (active) kw = *reinterpret_cast<const uint4*>(keep + rowbase + 4 * (size_t)u);
    const uint32_t c = __popc(kw.x) + __popc(kw.y) + __popc(kw.z) + __popc(kw.w);
    uint32_t tot;
    const uint32_t ex = block_excl_scan<HAWK_BLOCK / WAVE>(c, s_w, &tot);
    if (tot == 0) continue;  // workgroup-uniform
    uint64_t o = offsets[((size_t)s * hs.n_hap + h) * p.bph + blk] + ex;
    const uint32_t w[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k];
      while (x) {
        const uint32_t j = (uint32_t)__builtin_ctz(x);
        x &= x - 1;
        const uint32_t q = (4 * u + k) * 32 + j;
        uint32_t a = ext_glb(hs.plane[0] + rowbase, q).lo & lmask, cc = ext_glb(hs.plane[1] + rowbase, q).lo & lmask;
        uint32_t g = ext_glb(hs.plane[2] + rowbase, q).lo & lmask, t = ext_glb(hs.plane[3] + rowbase, q).lo & lmask;
        uint32_t nm = ext_glb(hs.plane[4] + rowbase, q).lo & lmask;
        if (s) {  // guide orientation = reverse complement of the + strand window
          const uint32_t ra = rev32(t, L), rc = rev32(g, L), rg = rev32(cc, L), rt = rev32(a, L);
          a = ra; cc = rc; g = rg; t = rt; nm = rev32(nm, L);
        }
        (void)a;
        OtSite st;
        st.code = spread(cc | t) | (spread(g | t) << 1);  // base i at bits 2i,2i+1: A0 C1 G2 T3
        st.nmask = nm;
        st.q = q | ((uint32_t)s << 31);
        st.row = h;
        sites[o++] = st;
      }
    }
  }
}
void hawk_launch_ot_sites(hipStream_t st, const HapSetDev& hs, const ScanParams& p, const uint32_t* keepF, const uint32_t* keepR,
                          const uint64_t* offsets, OtSite* sites) {
  hipLaunchKernelGGL(k_ot_sites, dim3(hs.n_hap * p.bph), dim3(HAWK_BLOCK), 0, st, hs, p, keepF, keepR, offsets, sites);
}

// All pairs.  Each thread owns one site; the guides stream through LDS in chunks and are read
// by every lane at the same address (LDS broadcast, conflict-free).  Per pair: XOR the 2-bit
// codes, fold each base's two bits into one, add the ambiguity bits, popcount.
#define OT_GCHUNK 1024
__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_match(const OtSite* __restrict__ sites, uint64_t n_sites,
                                                          const uint64_t* __restrict__ guides, uint32_t n_guides, int guidelen,
                                                          int sp0, int max_mm, OtHit* __restrict__ hits, uint64_t cap,
                                                          unsigned long long* __restrict__ n_hits) {
  __shared__ uint64_t s_g[OT_GCHUNK];
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  const bool have = i < n_sites;
  uint64_t code = 0;
  uint32_t nmsp = 0;
  const uint64_t smask = guidelen >= 32 ? ~0ull : ((1ull << (2 * guidelen)) - 1ull);
  bool live = false;
  if (have) {
    const OtSite st = sites[i];
    code = (st.code >> (2 * sp0)) & smask;
    nmsp = (st.nmask >> sp0) & (guidelen >= 32 ? 0xffffffffu : ((1u << guidelen) - 1u));
    live = __popc(nmsp) <= max_mm;  // more ambiguous bases than allowed mismatches: no guide can match
  }
  const uint64_t nm2 = spread(nmsp);  // ambiguity bits at the even positions, like the folded XOR
  for (uint32_t g0 = 0; g0 < n_guides; g0 += OT_GCHUNK) {
    const uint32_t ng = n_guides - g0 < OT_GCHUNK ? n_guides - g0 : OT_GCHUNK;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ng; t += HAWK_BLOCK) s_g[t] = guides[g0 + t];
    __syncthreads();
    if (!live) continue;
    for (uint32_t t = 0; t < ng; ++t) {
      const uint64_t x = code ^ s_g[t];
      const uint64_t m = ((x | (x >> 1)) & 0x5555555555555555ull) | nm2;
      const int mm = __popcll(m);
      if (mm <= max_mm) {
        const unsigned long long o = atomicAdd(n_hits, 1ull);  // the compiler aggregates this per wave
        if (o < cap) { OtHit hh; hh.site = i; hh.guide = g0 + t; hh.mm = (uint32_t)mm; hits[o] = hh; }
      }
    }
  }
}
void hawk_launch_ot_match(hipStream_t st, const OtSite* sites, uint64_t n_sites, const uint64_t* guides, uint32_t n_guides,
                          int guidelen, int sp0, int max_mm, OtHit* hits, uint64_t cap, unsigned long long* n_hits) {
  if (!n_sites || !n_guides) return;
  hipLaunchKernelGGL(k_ot_match, dim3((uint32_t)((n_sites + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, sites, n_sites,
                     guides, n_guides, guidelen, sp0, max_mm, hits, cap, n_hits);
}

// Seeded match.  Each thread owns one site and visits, per block, the bucket of guides whose key bases equal the
// site's (a gather from the L2-resident bucketed guide table).  A pair that agrees in the key bases of several
// blocks is met several times; it is reported at the first of them.
__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_match_seeded(const OtSite* __restrict__ sites, uint64_t n_sites, OtSeeds sd,
                                                                 const uint32_t* __restrict__ goff, const uint64_t* __restrict__ gcode,
                                                                 const uint32_t* __restrict__ gid, uint32_t n_guides, int guidelen,
                                                                 int sp0, int max_mm, OtHit* __restrict__ hits, uint64_t cap,
                                                                 unsigned long long* __restrict__ n_hits) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= n_sites) return;
  const OtSite st = sites[i];
  const uint64_t smask = guidelen >= 32 ? ~0ull : ((1ull << (2 * guidelen)) - 1ull);
  const uint64_t code = (st.code >> (2 * sp0)) & smask;
  const uint32_t nmsp = (st.nmask >> sp0) & (guidelen >= 32 ? 0xffffffffu : ((1u << guidelen) - 1u));
  if (__popc(nmsp) > max_mm) return;  // more ambiguous bases than allowed mismatches: no guide can match
  const uint64_t nm2 = spread(nmsp);
  for (int b = 0; b < sd.nb; ++b) {  // wave-uniform
    const int ks = sd.start[b], kl = sd.klen[b];
    if ((nmsp >> ks) & ((1u << kl) - 1u)) continue;  // an ambiguous base among the key bases: this block cannot agree
    const uint32_t key = (uint32_t)(code >> (2 * ks)) & ((1u << (2 * kl)) - 1u);
    const uint32_t lo = goff[sd.off_base[b] + key], hi = goff[sd.off_base[b] + key + 1];
    const uint64_t* gc = gcode + (size_t)b * n_guides;
    const uint32_t* gi = gid + (size_t)b * n_guides;
    for (uint32_t t = lo; t < hi; ++t) {
      const uint64_t x = code ^ gc[t];
      const uint64_t m = ((x | (x >> 1)) & 0x5555555555555555ull) | nm2;
      const int mm = __popcll(m);
      if (mm > max_mm) continue;
      bool earlier = false;
      for (int j = 0; j < b; ++j) earlier = earlier || (m & sd.pmask2[j]) == 0;
      if (earlier) continue;
      const unsigned long long o = atomicAdd(n_hits, 1ull);
      if (o < cap) { OtHit hh; hh.site = i; hh.guide = gi[t]; hh.mm = (uint32_t)mm; hits[o] = hh; }
    }
  }
}
void hawk_launch_ot_match_seeded(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtSeeds& sd, const uint32_t* goff,
                                 const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, int guidelen, int sp0, int max_mm,
                                 OtHit* hits, uint64_t cap, unsigned long long* n_hits) {
  if (!n_sites || !n_guides) return;
  hipLaunchKernelGGL(k_ot_match_seeded, dim3((uint32_t)((n_sites + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, sites,
                     n_sites, sd, goff, gcode, gid, n_guides, guidelen, sp0, max_mm, hits, cap, n_hits);
}

// The same with the guide buckets in LDS.  The global-gather version above moves a 64-byte sector from L2 for every
// 8-byte guide code a lane reads and runs into the L2 bandwidth (35 TB/s at the full C5 size); here the guides are
// taken OT_LDS_CHUNK at a time (blockIdx.y), each chunk bucketed on its own with 4 key bases per block, and a
// workgroup keeps the chunk's codes and bucket offsets in LDS while it walks OT_SITES_PER_WG sites.
// Tables per chunk c and block b: goff[(c * nb + b) * (OT_LDS_KEYS + 1) ...] (offsets inside the chunk),
// gcode / gid[(c * nb + b) * OT_LDS_CHUNK ...].
#define OT_SITES_PER_WG (HAWK_BLOCK * 16)
__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_match_seeded_lds(const OtSite* __restrict__ sites, uint64_t n_sites, OtSeeds sd,
                                                                     const uint32_t* __restrict__ goff, const uint64_t* __restrict__ gcode,
                                                                     const uint32_t* __restrict__ gid, uint32_t n_guides, int guidelen,
                                                                     int sp0, int max_mm, OtHit* __restrict__ hits, uint64_t cap,
                                                                     unsigned long long* __restrict__ n_hits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int nb = sd.nb;
  uint64_t* s_code = reinterpret_cast<uint64_t*>(s_raw);                                   // [nb][OT_LDS_CHUNK]
  uint16_t* s_off = reinterpret_cast<uint16_t*>(s_raw + (size_t)nb * OT_LDS_CHUNK * 8);    // [nb][OT_LDS_KEYS + 1]
  const uint32_t chunk = blockIdx.y;
  const uint32_t g0 = chunk * OT_LDS_CHUNK;
  const uint32_t ng = n_guides - g0 < OT_LDS_CHUNK ? n_guides - g0 : OT_LDS_CHUNK;
  const size_t tb = (size_t)chunk * nb;
  for (uint32_t t = threadIdx.x; t < (uint32_t)nb * OT_LDS_CHUNK; t += HAWK_BLOCK) {
    const uint32_t b = t / OT_LDS_CHUNK, j = t % OT_LDS_CHUNK;
    s_code[t] = j < ng ? gcode[(tb + b) * OT_LDS_CHUNK + j] : 0ull;
  }
  for (uint32_t t = threadIdx.x; t < (uint32_t)nb * (OT_LDS_KEYS + 1); t += HAWK_BLOCK)
    s_off[t] = (uint16_t)goff[tb * (OT_LDS_KEYS + 1) + t];
  __syncthreads();
  const uint64_t smask = guidelen >= 32 ? ~0ull : ((1ull << (2 * guidelen)) - 1ull);
  const uint64_t first = (uint64_t)blockIdx.x * OT_SITES_PER_WG;
#pragma unroll 1
  for (uint32_t r = 0; r < OT_SITES_PER_WG / HAWK_BLOCK; ++r) {
    const uint64_t i = first + (uint64_t)r * HAWK_BLOCK + threadIdx.x;
    if (i >= n_sites) break;  // sites are handed out in order: nothing behind it either
    const OtSite st = sites[i];
    const uint64_t code = (st.code >> (2 * sp0)) & smask;
    const uint32_t nmsp = (st.nmask >> sp0) & (guidelen >= 32 ? 0xffffffffu : ((1u << guidelen) - 1u));
    if (__popc(nmsp) > max_mm) continue;
    const uint64_t nm2 = spread(nmsp);
    for (int b = 0; b < nb; ++b) {
      const int ks = sd.start[b], kl = sd.klen[b];
      if ((nmsp >> ks) & ((1u << kl) - 1u)) continue;
      const uint32_t key = (uint32_t)(code >> (2 * ks)) & ((1u << (2 * kl)) - 1u);
      const uint32_t lo = s_off[b * (OT_LDS_KEYS + 1) + key], hi = s_off[b * (OT_LDS_KEYS + 1) + key + 1];
      for (uint32_t t = lo; t < hi; ++t) {
        const uint64_t x = code ^ s_code[b * OT_LDS_CHUNK + t];
        const uint64_t m = ((x | (x >> 1)) & 0x5555555555555555ull) | nm2;
        const int mm = __popcll(m);
        if (mm > max_mm) continue;
        bool earlier = false;
        for (int j = 0; j < b; ++j) earlier = earlier || (m & sd.pmask2[j]) == 0;
        if (earlier) continue;
        const unsigned long long o = atomicAdd(n_hits, 1ull);
        if (o < cap) { OtHit hh; hh.site = i; hh.guide = gid[(tb + b) * OT_LDS_CHUNK + t]; hh.mm = (uint32_t)mm; hits[o] = hh; }
      }
    }
  }
}
void hawk_launch_ot_match_seeded_lds(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtSeeds& sd, const uint32_t* goff,
                                     const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, uint32_t n_chunks, int guidelen,
                                     int sp0, int max_mm, OtHit* hits, uint64_t cap, unsigned long long* n_hits) {
  if (!n_sites || !n_guides) return;
  const size_t lds = (size_t)sd.nb * OT_LDS_CHUNK * 8 + (size_t)sd.nb * (OT_LDS_KEYS + 1) * 2;
  const dim3 grid((uint32_t)((n_sites + OT_SITES_PER_WG - 1) / OT_SITES_PER_WG), n_chunks);
  hipLaunchKernelGGL(k_ot_match_seeded_lds, grid, dim3(HAWK_BLOCK), lds, st, sites, n_sites, sd, goff, gcode, gid, n_guides, guidelen,
                     sp0, max_mm, hits, cap, n_hits);
}

// Pair seeds, candidates dealt evenly.  A wave takes 64 sites.  Per pair of blocks every lane looks its site's bucket up (two
// adjacent offsets); the wave's candidates - the concatenation of the 64 buckets - are then walked 64 at a time, candidate c by lane
// c: a 6-step search over the lanes' exclusive counts (LDS) names the owning site, whose code comes out of LDS, and the guide codes of
// consecutive candidates are consecutive words of the bucketed table.  No lane idles because its own bucket is shorter than a
// neighbour's (the per-lane bucket walks of k_ot_match_seeded(_lds) ran at a third of their lanes: profiles/r03_c5_pmc.json).
// A pair that agrees in more than two blocks is met in several tables and reported in the first (lowest two agreeing blocks).
__global__ __launch_bounds__(HAWK_BLOCK) void k_ot_match_pairs(const OtSite* __restrict__ sites, uint64_t n_sites, OtPairSeeds sd,
                                                                const uint32_t* __restrict__ goff, const uint64_t* __restrict__ gcode,
                                                                const uint32_t* __restrict__ gid, uint32_t n_guides, int guidelen,
                                                                int sp0, int max_mm, OtHit* __restrict__ hits, uint64_t cap,
                                                                unsigned long long* __restrict__ n_hits) {
  __shared__ uint64_t s_code[HAWK_BLOCK / WAVE][WAVE], s_nm2[HAWK_BLOCK / WAVE][WAVE];
  __shared__ uint32_t s_ex[HAWK_BLOCK / WAVE][WAVE + 1], s_lo[HAWK_BLOCK / WAVE][WAVE];
  const uint32_t wv = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const uint64_t i0 = (uint64_t)blockIdx.x * HAWK_BLOCK + wv * WAVE;  // the wave's first site
  if (i0 >= n_sites) return;                                            // wave-uniform; no workgroup barrier below
  const uint64_t i = i0 + lane;
  const uint64_t smask = guidelen >= 32 ? ~0ull : ((1ull << (2 * guidelen)) - 1ull);
  uint64_t code = 0;
  uint32_t nmsp = 0xffffffffu;
  if (i < n_sites) {
    const OtSite st = sites[i];
    code = (st.code >> (2 * sp0)) & smask;
    nmsp = (st.nmask >> sp0) & (guidelen >= 32 ? 0xffffffffu : ((1u << guidelen) - 1u));
  }
  const bool live = i < n_sites && __popc(nmsp) <= max_mm;  // more ambiguous bases than allowed mismatches: no guide can match
  const uint64_t nm2 = spread(nmsp);
  s_code[wv][lane] = code;
  s_nm2[wv][lane] = nm2;
  uint32_t blk_ok = 0;  // bit b: no ambiguous base among block b's key bases
  for (int b = 0; b < sd.nb; ++b) blk_ok |= (((nmsp >> sd.start[b]) & ((1u << sd.klen[b]) - 1u)) == 0 ? 1u : 0u) << b;
  __builtin_amdgcn_wave_barrier();
  // (Tried and dropped, same 5.9 ms at the full C5 size: all 15 tables' look-ups in flight before the first is used - in registers,
  // 73 VGPRs, or staged through LDS, 8.2 ms.  The wave's life is the chain per table - scan, owner search in LDS, guide load - at
  // ~19 candidates per table and wave; profiles/r04_c5_pmc.json.)
#pragma unroll 1
  for (int p = 0; p < sd.n_pairs; ++p) {  // wave-uniform
    {
    const int bi = sd.pi[p], bj = sd.pj[p];
    uint32_t lo = 0, cnt = 0;
    if (live && ((blk_ok >> bi) & 1u) && ((blk_ok >> bj) & 1u)) {
      const uint32_t ki = (uint32_t)(code >> (2 * sd.start[bi])) & ((1u << (2 * sd.klen[bi])) - 1u);
      const uint32_t kj = (uint32_t)(code >> (2 * sd.start[bj])) & ((1u << (2 * sd.klen[bj])) - 1u);
      const uint32_t* o = goff + sd.off_base[p] + (ki | (kj << (2 * sd.klen[bi])));
      lo = o[0];
      cnt = o[1] - lo;
    }
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1);
    if (T != 0) {
    s_ex[wv][lane] = inc - cnt;
    s_lo[wv][lane] = lo;
    __builtin_amdgcn_wave_barrier();
    const uint64_t* __restrict__ gc = gcode + (size_t)p * n_guides;
    const uint32_t* __restrict__ gi = gid + (size_t)p * n_guides;
#pragma unroll 1
    for (uint32_t c0 = 0; c0 < T; c0 += WAVE) {
      const uint32_t c = c0 + lane;
      if (c < T) {
        uint32_t l = 0;
#pragma unroll
        for (uint32_t step = WAVE / 2; step; step >>= 1) l += (s_ex[wv][l + step] <= c) ? step : 0u;  // the last lane whose candidates start at or before c
        const uint32_t t = s_lo[wv][l] + (c - s_ex[wv][l]);
        const uint64_t x = s_code[wv][l] ^ gc[t];
        const uint64_t m = ((x | (x >> 1)) & 0x5555555555555555ull) | s_nm2[wv][l];
        const int mm = __popcll(m);
        if (mm <= max_mm) {
          // the blocks that agree, lowest first: this table is the pair's first iff they are (bi, bj)
          int first = -1, second = -1;
          for (int b = 0; b < sd.nb; ++b)
            if ((m & sd.pmask2[b]) == 0) { if (first < 0) first = b; else if (second < 0) second = b; }
          if (first == bi && second == bj) {
            const unsigned long long o = atomicAdd(n_hits, 1ull);
            if (o < cap) { OtHit hh; hh.site = i0 + l; hh.guide = gi[t]; hh.mm = (uint32_t)mm; hits[o] = hh; }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    }
    }
  }
}
void hawk_launch_ot_match_pairs(hipStream_t st, const OtSite* sites, uint64_t n_sites, const OtPairSeeds& sd, const uint32_t* goff,
                                const uint64_t* gcode, const uint32_t* gid, uint32_t n_guides, int guidelen, int sp0, int max_mm,
                                OtHit* hits, uint64_t cap, unsigned long long* n_hits) {
  if (!n_sites || !n_guides) return;
  hipLaunchKernelGGL(k_ot_match_pairs, dim3((uint32_t)((n_sites + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, sites, n_sites, sd,
                     goff, gcode, gid, n_guides, guidelen, sp0, max_mm, hits, cap, n_hits);
}

__global__ __launch_bounds__(256) void k_ot_gather(const OtSite* __restrict__ sites, const OtHit* __restrict__ hits, uint64_t n_hits,
                                                   OtSite* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_hits) out[i] = sites[hits[i].site];
}
void hawk_launch_ot_gather(hipStream_t st, const OtSite* sites, const OtHit* hits, uint64_t n_hits, OtSite* out) {
  if (n_hits) hipLaunchKernelGGL(k_ot_gather, dim3((uint32_t)((n_hits + 255) / 256)), dim3(256), 0, st, sites, hits, n_hits, out);
}
