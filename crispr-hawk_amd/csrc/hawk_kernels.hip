// hawk_kernels.hip — K1 pack, K2 raw PAM scan (pam_search) and the stand-alone CFD kernel.
// The fused search (K2+K3+K4) lives in hawk_search.hip; shared device helpers in hawk_bits.h.
//
// Everything on this path is integer / bitwise work on bit-sliced sequence planes; there is no
// GEMM shape anywhere, so no MFMA: the kernels are HBM-streaming.  Wavefront = 64 lanes.
//
//   k_pack       K1  cased ASCII -> planes A,C,G,T,V (wave ballots: one v_cmp per plane per 64
//                    bases; encoder.py:18-57 + the lower-case-as-data rule of haplotype.py:120)
//   k_scan_raw   K2  PAM bitmask scan, both strands, hits at the PAM position inside
//                    [scan_start, scan_stop) (search_guides.py:32-46, 87-99)
//   k_emit_hits      deterministic compaction of the hit bits into ascending position lists
//   k_cfd        K4  compute_cfd on string triples (cfdscore.py:53-95)
#include "hawk_bits.h"

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
// K2: raw PAM scan (pam_search)
// ---------------------------------------------------------------------------------------
// bit q of keepF / keepR = the PAM / its reverse complement matches at relative position q,
// q in [scan_start, scan_stop).  counts: [strand][haplotype][workgroup] survivors.
__global__ __launch_bounds__(HAWK_BLOCK) void k_scan_raw(HapSetDev hs, ScanParams p, uint32_t* __restrict__ keepF,
                                                          uint32_t* __restrict__ keepR, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_acc[2];
  const uint32_t h = blockIdx.x / p.bph, blk = blockIdx.x % p.bph;
  const uint32_t u = blk * HAWK_BLOCK + threadIdx.x;
  const bool active = u < hs.S / 4;
  const size_t rowbase = (size_t)h * hs.S;
  const int ss = hs.scan_start[h], se = hs.scan_stop[h];
  if (threadIdx.x < 2) s_acc[threadIdx.x] = 0;
  uint32_t A[6] = {0, 0, 0, 0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0}, G[6] = {0, 0, 0, 0, 0, 0}, T[6] = {0, 0, 0, 0, 0, 0};
  if (p.need & 1u) load6(hs.plane[0] + rowbase, u, hs.S, active, A);
  if (p.need & 2u) load6(hs.plane[1] + rowbase, u, hs.S, active, C);
  if (p.need & 4u) load6(hs.plane[2] + rowbase, u, hs.S, active, G);
  if (p.need & 8u) load6(hs.plane[3] + rowbase, u, hs.S, active, T);
  uint32_t mF[4], mR[4];
  pam_match(A, C, G, T, p.pam_fwd, p.pamlen, p.poF, mF);
  pam_match(A, C, G, T, p.pam_rev, p.pamlen, p.poR, mR);
  const int base0 = (int)(u * 128u);
  uint32_t cF = 0, cR = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t rm = range_mask(base0 + 32 * k, ss, se);
    mF[k] = active ? (mF[k] & rm) : 0u;
    mR[k] = active ? (mR[k] & rm) : 0u;
    cF += __popc(mF[k]); cR += __popc(mR[k]);
  }
  if (active) {
    *reinterpret_cast<uint4*>(keepF + rowbase + 4 * (size_t)u) = make_uint4(mF[0], mF[1], mF[2], mF[3]);
    *reinterpret_cast<uint4*>(keepR + rowbase + 4 * (size_t)u) = make_uint4(mR[0], mR[1], mR[2], mR[3]);
  }
  const uint32_t w0 = wave_sum(cF | (cR << 16));  // per-wave sums <= 8192 each
  __syncthreads();
  if ((threadIdx.x & (WAVE - 1)) == 0) { atomicAdd(&s_acc[0], w0 & 0xffffu); atomicAdd(&s_acc[1], w0 >> 16); }
  __syncthreads();
  if (threadIdx.x == 0) {
    counts[(size_t)h * p.bph + blk] = s_acc[0];
    counts[((size_t)hs.n_hap + h) * p.bph + blk] = s_acc[1];
  }
}

void hawk_launch_scan_raw(hipStream_t st, const HapSetDev& hs, const ScanParams& p, uint32_t* keepF, uint32_t* keepR,
                          uint32_t* counts) {
  hipLaunchKernelGGL(k_scan_raw, dim3(hs.n_hap * p.bph), dim3(HAWK_BLOCK), 0, st, hs, p, keepF, keepR, counts);
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_emit_hits(HapSetDev hs, uint32_t bph, const uint32_t* __restrict__ keepF,
                                                           const uint32_t* __restrict__ keepR,
                                                           const uint64_t* __restrict__ offsets, uint64_t n_fwd_total,
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
    uint64_t o = offsets[((size_t)s * hs.n_hap + h) * bph + blk] + ex;
    const uint32_t w[4] = {kw.x, kw.y, kw.z, kw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t x = w[k];
      while (x) {
        const uint32_t j = (uint32_t)__builtin_ctz(x);
        x &= x - 1;
        const uint32_t q = (4 * u + k) * 32 + j;
        if (s == 0) hits_fwd[o] = q; else hits_rev[o - n_fwd_total] = q;
        ++o;
      }
    }
  }
}

void hawk_launch_emit_hits(hipStream_t st, const HapSetDev& hs, uint32_t bph, const uint32_t* keepF, const uint32_t* keepR,
                           const uint64_t* offsets, uint64_t n_fwd_total, uint32_t* hits_fwd, uint32_t* hits_rev) {
  hipLaunchKernelGGL(k_emit_hits, dim3(hs.n_hap * bph), dim3(HAWK_BLOCK), 0, st, hs, bph, keepF, keepR, offsets, n_fwd_total,
                     hits_fwd, hits_rev);
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
