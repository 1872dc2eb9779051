// hawk_meta.hip — the metadata of an expansion plan's rows, built on the device from the carried-variant lists the
// genotype inversion left there (hawk_vcf.hip): position-map segments (haplotype.py:90-159), the reverse look-ups the scan
// bounds need (search_guides.py:49-84), the per-tile records of the search kernels, and the checks the host used to run
// over every list entry.  Round 2 did all of this in numpy / host helpers of the library over lists that had first been
// downloaded - 80 of the 83 ms a C3 expansion took; what crosses PCIe now is a few words per ROW.
//   k_list_check   every list entry: ascending, non-overlapping within its row; an indel must not reach past the
//                  region's original length (the reference's clamp, haplotype.py:199-201)
//   k_seg_count    per row: segments its carried indels open (a deletion one, an insertion of n bases n + 1)
//   k_seg_fill     ... written behind the row's identity segment
//   k_rev_lookup   posmap_rev[g] of every row for two genomic positions (last relative position mapping to g, -1: deleted)
//   k_tile_meta    TileMeta of every (row, tile)
#include "hawk_bits.h"

// status bits of k_list_check
#define LC_OVERLAP 1u
#define LC_CLAMP 2u
#define LC_ORDER 4u

__global__ __launch_bounds__(256) void k_list_check(const uint64_t* __restrict__ row_off, uint32_t n_rows, const uint32_t* __restrict__ hv_idx,
                                                    const int32_t* __restrict__ hv_o, const int32_t* __restrict__ v_r0,
                                                    const int32_t* __restrict__ v_span, const int32_t* __restrict__ v_chain, uint32_t n_var,
                                                    uint32_t ref_len, int check_clamp, uint32_t* __restrict__ status) {
  // one workgroup per row (row r's entries: [row_off[r], row_off[r + 1]))
  const uint32_t r = blockIdx.x;
  const uint64_t lo = row_off[r], hi = row_off[r + 1];
  uint32_t bad = 0;
  for (uint64_t e = lo + threadIdx.x; e < hi; e += 256) {
    const uint32_t v = hv_idx[e];
    if (v >= n_var) { bad |= LC_ORDER; continue; }
    if (e > lo) {
      const uint32_t pv = hv_idx[e - 1];
      if (pv >= n_var || v <= pv) bad |= LC_ORDER;
      else if (v_r0[v] < v_r0[pv] + v_span[pv]) bad |= LC_OVERLAP;
    }
    if (check_clamp && v_chain[v] != 0 && (int64_t)hv_o[e] + v_span[v] > (int64_t)ref_len) bad |= LC_CLAMP;
  }
  if (bad) atomicOr(status, bad);
}

// segments one carried indel opens inside a row of `len` bases: rel = o + 1 + k, k < (chain < 0 ? 1 : chain + 1), rel < len
__device__ __forceinline__ uint32_t indel_segments(int32_t o, int32_t chain, int64_t len) {
  const int64_t n = chain < 0 ? 1 : (int64_t)chain + 1;
  const int64_t room = len - ((int64_t)o + 1);
  return (uint32_t)(room <= 0 ? 0 : (n < room ? n : room));
}

// row r's carried indels are the entries indel[ioff[r] .. ioff[r + 1]) (entry indices into hv_idx / hv_o)
__global__ __launch_bounds__(256) void k_seg_count(const uint64_t* __restrict__ ioff, const uint32_t* __restrict__ indel,
                                                   const uint32_t* __restrict__ hv_idx, const int32_t* __restrict__ hv_o,
                                                   const int32_t* __restrict__ v_chain, const uint32_t* __restrict__ hap_len,
                                                   uint32_t* __restrict__ seg_cnt) {
  __shared__ uint32_t s_sum;
  const uint32_t r = blockIdx.x;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const uint64_t lo = ioff[r], hi = ioff[r + 1];
  const int64_t len = hap_len[r];
  uint32_t c = 0;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t e = indel[i];
    c += indel_segments(hv_o[e], v_chain[hv_idx[e]], len);
  }
  c = wave_sum(c);
  if ((threadIdx.x & (WAVE - 1)) == 0 && c) atomicAdd(&s_sum, c);
  __syncthreads();
  if (threadIdx.x == 0) seg_cnt[r] = s_sum + 1u;  // + the identity segment every row starts with
}

// exclusive scan of n u32 counts into n + 1 u32 offsets, one workgroup (n = rows of a plan, chunks of its records): four
// consecutive counts per thread and round
__global__ __launch_bounds__(1024) void k_scan_u32(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ off) {
  __shared__ uint32_t s_w[1024 / WAVE];
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 4096) {
    const uint32_t i = b0 + threadIdx.x * 4;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = i + k < n ? cnt[i + k] : 0u;
    uint32_t tot;
    uint32_t ex = carry + block_excl_scan<1024 / WAVE>(c[0] + c[1] + c[2] + c[3], s_w, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < n) off[i + k] = ex;
      ex += c[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) off[n] = carry;
}

// the same for two count arrays at once (one launch: the dictionary's chunks open instances AND put some of them on a list).
// A thread takes sixteen CONSECUTIVE counts of either array (four 16-byte loads each, all in flight at once), scans them in
// registers, and the workgroup scans the threads' sums once per 16384 counts: a scan of 1.5 x 10^4 counts is latency and nothing
// else, so it is one round trip and one pair of barriers instead of fifteen.  Arrays and offsets 16-byte aligned, as the pool's are.
__global__ __launch_bounds__(1024) void k_scan2_u32(const uint32_t* __restrict__ cnt_a, const uint32_t* __restrict__ cnt_b, uint32_t n,
                                                    uint32_t* __restrict__ off_a, uint32_t* __restrict__ off_b) {
  __shared__ uint32_t s_w[2][1024 / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  uint32_t carry_a = 0, carry_b = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += 16 * 1024) {
    const uint32_t i0 = b0 + threadIdx.x * 16;
    uint32_t a[16], b[16];
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 va = reinterpret_cast<const uint4*>(cnt_a + i0)[q], vb = reinterpret_cast<const uint4*>(cnt_b + i0)[q];
        a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
        b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) { a[k] = i0 + k < n ? cnt_a[i0 + k] : 0u; b[k] = i0 + k < n ? cnt_b[i0 + k] : 0u; }
    }
    uint32_t sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const uint32_t xa = a[k], xb = b[k]; a[k] = sa; b[k] = sb; sa += xa; sb += xb; }  // exclusive, in place
    const uint32_t ia = wave_incl_scan(sa), ib = wave_incl_scan(sb);
    if (lane == WAVE - 1) { s_w[0][wv] = ia; s_w[1][wv] = ib; }
    __syncthreads();
    uint32_t pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < 1024 / WAVE; ++w) {
      const uint32_t xa = s_w[0][w], xb = s_w[1][w];
      if (w < wv) { pa += xa; pb += xb; }
      ta += xa; tb += xb;
    }
    __syncthreads();
    const uint32_t ea = carry_a + pa + ia - sa, eb = carry_b + pb + ib - sb;
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        reinterpret_cast<uint4*>(off_a + i0)[q] = make_uint4(ea + a[4 * q], ea + a[4 * q + 1], ea + a[4 * q + 2], ea + a[4 * q + 3]);
        reinterpret_cast<uint4*>(off_b + i0)[q] = make_uint4(eb + b[4 * q], eb + b[4 * q + 1], eb + b[4 * q + 2], eb + b[4 * q + 3]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) if (i0 + k < n) { off_a[i0 + k] = ea + a[k]; off_b[i0 + k] = eb + b[k]; }
    }
    carry_a += ta; carry_b += tb;
  }
  if (threadIdx.x == 0) { off_a[n] = carry_a; off_b[n] = carry_b; }
}

__global__ __launch_bounds__(256) void k_seg_fill(const uint64_t* __restrict__ ioff, const uint32_t* __restrict__ indel,
                                                  const uint32_t* __restrict__ hv_idx, const int32_t* __restrict__ hv_o,
                                                  const int32_t* __restrict__ v_r0, const int32_t* __restrict__ v_chain,
                                                  const uint32_t* __restrict__ hap_len, int64_t startp, const uint32_t* __restrict__ seg_off,
                                                  uint32_t* __restrict__ seg_rel, int64_t* __restrict__ seg_gen) {
  __shared__ uint32_t s_w[256 / WAVE];
  const uint32_t r = blockIdx.x;
  const uint64_t lo = ioff[r], hi = ioff[r + 1];
  const int64_t len = hap_len[r];
  uint32_t at = seg_off[r];
  if (threadIdx.x == 0) { seg_rel[at] = 0; seg_gen[at] = startp; }
  ++at;
  for (uint64_t b0 = lo; b0 < hi; b0 += 256) {  // workgroup-uniform trip count
    const uint64_t i = b0 + threadIdx.x;
    uint32_t c = 0;
    int32_t o = 0, ch = 0;
    int64_t pos = 0;
    if (i < hi) {
      const uint32_t e = indel[i];
      const uint32_t v = hv_idx[e];
      o = hv_o[e]; ch = v_chain[v]; pos = (int64_t)v_r0[v] + startp;
      c = indel_segments(o, ch, len);
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan<256 / WAVE>(c, s_w, &tot);
    for (uint32_t k = 0; k < c; ++k) {  // a deletion: one segment behind the deleted bases; an insertion: its bases all map to the anchor
      seg_rel[at + ex + k] = (uint32_t)(o + 1 + (int32_t)k);
      seg_gen[at + ex + k] = ch < 0 ? pos + 1 - ch : ((int32_t)k < ch ? pos : pos + 1);
    }
    at += tot;
  }
}

// posmap_rev[g] for two positions per row: the LAST relative position whose genomic position is g (the reference rebuilds
// the reverse dict by overwrite, haplotype.py:159), -1 where g is deleted from the row or outside it.  One wave per row.
__global__ __launch_bounds__(256) void k_rev_lookup(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_rel,
                                                    const int64_t* __restrict__ seg_gen, const uint32_t* __restrict__ hap_len, uint32_t n_rows,
                                                    int64_t g0, int64_t g1, int64_t* __restrict__ out0, int64_t* __restrict__ out1) {
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t r = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (r >= n_rows) return;  // wave-uniform
  const uint32_t lo = seg_off[r], hi = seg_off[r + 1];
  long long best0 = -1, best1 = -1;
  int64_t key0 = -1, key1 = -1;  // the segment index the match came from: the last segment wins
  for (uint32_t k = lo + lane; k < hi; k += WAVE) {
    const int64_t rel = seg_rel[k], gen = seg_gen[k];
    const int64_t end = k + 1 < hi ? (int64_t)seg_rel[k + 1] : (int64_t)hap_len[r];
    const int64_t last = gen + (end - rel) - 1;
    if (gen <= g0 && g0 <= last) { best0 = rel + (g0 - gen); key0 = k; }  // ascending k per lane: later overwrites
    if (gen <= g1 && g1 <= last) { best1 = rel + (g1 - gen); key1 = k; }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    const int64_t ok0 = __shfl_xor(key0, d), ok1 = __shfl_xor(key1, d);
    const long long ob0 = __shfl_xor(best0, d), ob1 = __shfl_xor(best1, d);
    if (ok0 > key0) { key0 = ok0; best0 = ob0; }
    if (ok1 > key1) { key1 = ok1; best1 = ob1; }
  }
  if (lane == 0) { out0[r] = best0; out1[r] = best1; }
}

__global__ __launch_bounds__(256) void k_tile_meta(const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_rel,
                                                   const uint32_t* __restrict__ hap_len, const uint8_t* __restrict__ is_ref,
                                                   const int32_t* __restrict__ scan_start, const int32_t* __restrict__ scan_stop,
                                                   uint32_t n_rows, uint32_t bph, TileMeta* __restrict__ tm) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)n_rows * bph) return;
  const uint32_t h = (uint32_t)(i / bph), blk = (uint32_t)(i % bph);
  const uint32_t q0 = blk * HAWK_BLOCK * 128u;
  uint32_t a = seg_off[h], b = seg_off[h + 1];  // first segment of the row with seg_rel > q0 (seg_rel[a] == 0 <= q0)
  const uint32_t end = b;
  while (a < b) { const uint32_t m = (a + b) >> 1; if (seg_rel[m] <= q0) a = m + 1; else b = m; }
  TileMeta t;
  t.h = h; t.blk = blk; t.hap_len = hap_len[h];
  t.scan_start = scan_start[h]; t.scan_stop = scan_stop[h]; t.is_ref = is_ref[h] ? 1u : 0u;
  t.seg0 = a - 1; t.seg_end = end;
  tm[i] = t;
}

void hawk_launch_list_check(hipStream_t st, const uint64_t* row_off, uint32_t n_rows, const uint32_t* hv_idx, const int32_t* hv_o,
                            const int32_t* v_r0, const int32_t* v_span, const int32_t* v_chain, uint32_t n_var, uint32_t ref_len,
                            int check_clamp, uint32_t* status) {
  if (n_rows) hipLaunchKernelGGL(k_list_check, dim3(n_rows), dim3(256), 0, st, row_off, n_rows, hv_idx, hv_o, v_r0, v_span, v_chain, n_var, ref_len,
                                 check_clamp, status);
}
void hawk_launch_segments(hipStream_t st, const uint64_t* ioff, const uint32_t* indel, const uint32_t* hv_idx, const int32_t* hv_o,
                          const int32_t* v_r0, const int32_t* v_chain, const uint32_t* hap_len, uint32_t n_rows, int64_t startp,
                          uint32_t* seg_cnt, uint32_t* seg_off, uint32_t* seg_rel /* null: count + offsets only */, int64_t* seg_gen) {
  if (!seg_rel) {
    hipLaunchKernelGGL(k_seg_count, dim3(n_rows), dim3(256), 0, st, ioff, indel, hv_idx, hv_o, v_chain, hap_len, seg_cnt);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, seg_cnt, n_rows, seg_off);
  } else {
    hipLaunchKernelGGL(k_seg_fill, dim3(n_rows), dim3(256), 0, st, ioff, indel, hv_idx, hv_o, v_r0, v_chain, hap_len, startp, seg_off, seg_rel, seg_gen);
  }
}
void hawk_launch_scan2_u32(hipStream_t st, const uint32_t* cnt_a, const uint32_t* cnt_b, uint32_t n, uint32_t* off_a, uint32_t* off_b) {
  hipLaunchKernelGGL(k_scan2_u32, dim3(1), dim3(1024), 0, st, cnt_a, cnt_b, n, off_a, off_b);
}
void hawk_launch_scan_u32(hipStream_t st, const uint32_t* cnt, uint32_t n, uint32_t* off) {
  hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, st, cnt, n, off);
}
void hawk_launch_rev_lookup(hipStream_t st, const uint32_t* seg_off, const uint32_t* seg_rel, const int64_t* seg_gen, const uint32_t* hap_len,
                            uint32_t n_rows, int64_t g0, int64_t g1, int64_t* out0, int64_t* out1) {
  hipLaunchKernelGGL(k_rev_lookup, dim3((n_rows + 3) / 4), dim3(256), 0, st, seg_off, seg_rel, seg_gen, hap_len, n_rows, g0, g1, out0, out1);
}
void hawk_launch_tile_meta(hipStream_t st, const uint32_t* seg_off, const uint32_t* seg_rel, const uint32_t* hap_len, const uint8_t* is_ref,
                           const int32_t* scan_start, const int32_t* scan_stop, uint32_t n_rows, uint32_t bph, TileMeta* tm) {
  const uint64_t n = (uint64_t)n_rows * bph;
  hipLaunchKernelGGL(k_tile_meta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seg_off, seg_rel, hap_len, is_ref, scan_start, scan_stop,
                     n_rows, bph, tm);
}
