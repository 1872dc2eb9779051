// hawk_vcf.hip — f3: VCF sample columns -> allele codes -> carried-variant lists, on the device.
//
// Reference: VariantRecord.read_vcf_line -> _genotypes_to_samples (variant.py:286-311, 558-619) splits
// every genotype string of every record in Python (2504 samples x 31 k records per Mb), and
// compute_haplotypes_phased (haplotypes.py:132-159) turns the per-variant sample sets into per-sample
// variant lists.  Here the raw text of the records goes to HBM once:
//   k_gt_parse   one workgroup per record: tabs are counted per 16-byte chunk, a workgroup scan gives
//                every chunk the index of its first field, and the thread that owns a field start parses
//                "a|b" (multi-digit alleles, '.' = missing, anything behind ':' ignored) into two bytes
//                codes[record][2*sample + copy]: 0 REF, k = k-th ALT, 255 missing;
//   k_gt_count / k_gt_fill   one wave per chromosome copy (column): variants are tested 64 at a time
//                (codes[line(j)][col] == allele(j)), the ballots kept; the fill pass turns them into the
//                ascending carried-variant list of the column plus, per entry, its offset in the
//                haplotype being built (r0 + running sum of the length changes): exactly the hv_idx / hv_o
//                inputs of hawk_hapset_expand.
#include "hawk_bits.h"

#define GT_CHUNK 16

// flags per record: 1 = a genotype without '|' (unphased separator or haploid), 2 = field count != n_samples,
// 4 = unexpected character inside a genotype
__global__ __launch_bounds__(256) void k_gt_parse(const uint8_t* __restrict__ text, const uint64_t* __restrict__ line_off,
                                                  const uint64_t* __restrict__ gt_off, uint32_t n_samples,
                                                  uint8_t* __restrict__ codes, uint8_t* __restrict__ flags) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_flag;
  const uint32_t tid = threadIdx.x;
  const uint64_t rec = blockIdx.x;
  const uint64_t lo = gt_off[rec];
  uint64_t hi = line_off[rec + 1];  // one past the '\n'
  while (hi > lo && (text[hi - 1] == '\n' || text[hi - 1] == '\r')) --hi;  // workgroup-uniform
  if (tid == 0) s_flag = 0;
  __syncthreads();
  uint8_t* out = codes + rec * 2ull * n_samples;
  uint32_t field_base = 0, myflag = 0;
  for (uint64_t base = lo; base < hi; base += 256ull * GT_CHUNK) {  // workgroup-uniform trip count
    const uint64_t a = base + (uint64_t)tid * GT_CHUNK;
    uint8_t c[GT_CHUNK];
    uint32_t ntab = 0;
#pragma unroll
    for (int k = 0; k < GT_CHUNK; ++k) {
      c[k] = a + k < hi ? text[a + k] : 0;
      ntab += c[k] == '\t';
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan<4>(ntab, s_w, &tot);
    uint32_t f = field_base + ex;  // index of the field the chunk's first byte belongs to
    uint8_t prev = a == lo ? (uint8_t)'\t' : (a < hi ? text[a - 1] : 0);
#pragma unroll 1
    for (int k = 0; k < GT_CHUNK; ++k) {
      if (a + k >= hi) break;
      if (prev == '\t') {  // a field starts here; f counts the tabs before it
        if (f < n_samples) {
          uint64_t p = a + k;
          uint32_t v[2] = {255u, 255u};
          int nal = 0;
          bool bar = false;
          for (; nal < 2; ++nal) {
            uint8_t ch = p < hi ? text[p] : (uint8_t)'\t';
            if (ch == '.') { v[nal] = 255u; ++p; }
            else if (ch >= '0' && ch <= '9') {
              uint32_t x = 0;
              while (p < hi && (ch = text[p]) >= '0' && ch <= '9') { x = x * 10u + (ch - '0'); if (x > 254u) x = 254u; ++p; }
              v[nal] = x;
            } else { myflag |= 4u; break; }
            ch = p < hi ? text[p] : (uint8_t)'\t';
            if (nal == 0) {
              if (ch == '|') { bar = true; ++p; }
              else if (ch == '/') { ++p; }
              else { ++nal; break; }  // haploid
            }
          }
          if (!bar) myflag |= 1u;
          const uint8_t endc = p < hi ? text[p] : (uint8_t)'\t';
          if (endc != '\t' && endc != ':') myflag |= (endc == '|' || endc == '/') ? 1u : 4u;  // more than two alleles -> not phased diploid
          out[2ull * f] = (uint8_t)v[0];
          out[2ull * f + 1] = (uint8_t)v[1];
        }
      }
      prev = c[k];
      f += c[k] == '\t';
    }
    field_base += tot;
  }
  // fields = tabs + 1 (an empty section has no field at all)
  const uint32_t nfields = hi > lo ? field_base + 1 : 0;
  if (nfields != n_samples) myflag |= 2u;
  if (myflag) atomicOr(&s_flag, myflag);
  __syncthreads();
  if (tid == 0) flags[rec] = (uint8_t)s_flag;
}

// ballots[col][chunk] = which of variants 64*chunk .. +63 the column carries;
// col_count[col], col_count[n_cols + col]: carried variants, carried indels (var_chain != 0) of the column (zeroed by the launcher).
// A wave takes 64 neighbouring columns and GT_CPW chunks of variants: lane = column, so a variant's 64 codes are one coalesced
// 64-byte read and every lane shifts its own ballot word together (the first version gave a wave one column and its lanes 64
// variants - 64 cache lines per load for 64 bytes of use: 0.74 ms on C3 where the matrix is 155 MB).
#define GT_CPW 4
__global__ __launch_bounds__(256) void k_gt_count(const uint8_t* __restrict__ codes, uint32_t n_cols, const uint32_t* __restrict__ var_line,
                                                  const uint8_t* __restrict__ var_allele, const int32_t* __restrict__ var_chain, uint32_t n_var,
                                                  uint32_t n_chunk, unsigned long long* __restrict__ ballots, uint32_t* __restrict__ col_count) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t col = blockIdx.x * 64u + lane;
  const uint32_t ch0 = (blockIdx.y * 4u + (threadIdx.x >> 6)) * GT_CPW;
  if (ch0 >= n_chunk) return;  // wave-uniform
  const bool live = col < n_cols;
  uint32_t cnt = 0, cnt_indel = 0;
  for (uint32_t ch = ch0; ch < ch0 + GT_CPW && ch < n_chunk; ++ch) {
    unsigned long long b = 0, bi = 0;
    const uint32_t j0 = ch * 64u, nj = j0 + 64u <= n_var ? 64u : n_var - j0;
    // the chunk's 64 variants: line, allele and chain read once by the lanes, handed round by v_readlane (compile-time lane: the loop
    // is unrolled) - the 64 byte loads of a lane then depend on nothing but each other's issue
    const uint32_t jl = j0 + (lane < nj ? lane : 0u);
    const uint32_t my_line = var_line[jl], my_meta = (uint32_t)var_allele[jl] | (var_chain[jl] != 0 ? 0x100u : 0u);
    uint8_t c[64];
#pragma unroll
    for (int jj = 0; jj < 64; ++jj) {
      const uint32_t line = (uint32_t)__builtin_amdgcn_readlane((int)my_line, jj);
      c[jj] = (live && (uint32_t)jj < nj) ? codes[(size_t)line * n_cols + col] : (uint8_t)254;  // 254: no allele of a variant (they are <= 253)
    }
#pragma unroll
    for (int jj = 0; jj < 64; ++jj) {
      const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)my_meta, jj);
      const bool carried = (uint32_t)jj < nj && c[jj] == (uint8_t)(meta & 0xffu);
      b |= (unsigned long long)carried << jj;
      bi |= (unsigned long long)(carried && (meta & 0x100u)) << jj;
    }
    if (live) ballots[(size_t)col * n_chunk + ch] = b;
    cnt += (uint32_t)__popcll(b);
    cnt_indel += (uint32_t)__popcll(bi);
  }
  if (live && cnt) atomicAdd(&col_count[col], cnt);
  if (live && cnt_indel) atomicAdd(&col_count[n_cols + col], cnt_indel);
}

__global__ __launch_bounds__(256) void k_gt_fill(uint32_t n_cols, const int32_t* __restrict__ var_r0, const int32_t* __restrict__ var_chain,
                                                 uint32_t n_var, uint32_t n_chunk, const unsigned long long* __restrict__ ballots,
                                                 const uint64_t* __restrict__ col_off, uint32_t* __restrict__ hv_idx,
                                                 int32_t* __restrict__ hv_o, int64_t* __restrict__ col_delta,
                                                 const uint64_t* __restrict__ indel_off, uint32_t* __restrict__ indel_entry) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t col = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (col >= n_cols) return;
  uint64_t o = col_off[col];
  uint64_t io = indel_off[col];  // where this column's carried indels are listed (by entry index: position-map segments come from them)
  int64_t run = 0;  // sum of the length changes of the variants already placed
  for (uint32_t ch = 0; ch < n_chunk; ++ch) {
    const unsigned long long b = ballots[(size_t)col * n_chunk + ch];
    if (b == 0) continue;  // wave-uniform
    const uint32_t j = ch * 64u + lane;
    const bool carried = (b >> lane) & 1ull;
    const int32_t chn = carried ? var_chain[j] : 0;
    const uint32_t inc = wave_incl_scan((uint32_t)chn);  // two's complement: sums of signed values
    const uint32_t rank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (carried) {
      hv_idx[o + rank] = j;
      hv_o[o + rank] = (int32_t)((int64_t)var_r0[j] + run + (int64_t)(int32_t)(inc - (uint32_t)chn));
    }
    const unsigned long long bi = __ballot(chn != 0);
    if (chn != 0) indel_entry[io + (uint32_t)__popcll(bi & ((1ull << lane) - 1ull))] = (uint32_t)(o + rank);
    io += (uint64_t)__popcll(bi);
    run += (int64_t)(int32_t)__builtin_amdgcn_readlane((int)inc, 63);
    o += (uint64_t)__popcll(b);
  }
  if (lane == 0) col_delta[col] = run;
}

void hawk_launch_gt_parse(hipStream_t st, const uint8_t* text, const uint64_t* line_off, const uint64_t* gt_off, uint64_t n_lines,
                          uint32_t n_samples, uint8_t* codes, uint8_t* flags) {
  if (n_lines) hipLaunchKernelGGL(k_gt_parse, dim3((unsigned)n_lines), dim3(256), 0, st, text, line_off, gt_off, n_samples, codes, flags);
}
void hawk_launch_gt_count(hipStream_t st, const uint8_t* codes, uint32_t n_cols, const uint32_t* var_line, const uint8_t* var_allele,
                          const int32_t* var_chain, uint32_t n_var, unsigned long long* ballots, uint32_t* col_count /* [2 * n_cols] */) {
  const uint32_t n_chunk = (n_var + 63u) / 64u;
  (void)hipMemsetAsync(col_count, 0, (size_t)2 * n_cols * 4, st);
  hipLaunchKernelGGL(k_gt_count, dim3((n_cols + 63u) / 64u, (n_chunk + 4u * GT_CPW - 1u) / (4u * GT_CPW)), dim3(256), 0, st, codes, n_cols, var_line,
                     var_allele, var_chain, n_var, n_chunk, ballots, col_count);
}
void hawk_launch_gt_fill(hipStream_t st, uint32_t n_cols, const int32_t* var_r0, const int32_t* var_chain, uint32_t n_var,
                         const unsigned long long* ballots, const uint64_t* col_off, uint32_t* hv_idx, int32_t* hv_o, int64_t* col_delta,
                         const uint64_t* indel_off, uint32_t* indel_entry) {
  const uint32_t n_chunk = (n_var + 63u) / 64u;
  hipLaunchKernelGGL(k_gt_fill, dim3((n_cols + 3u) / 4u), dim3(256), 0, st, n_cols, var_r0, var_chain, n_var, n_chunk, ballots, col_off,
                     hv_idx, hv_o, col_delta, indel_off, indel_entry);
}
