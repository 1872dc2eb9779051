// hawk_scorers.hip — K6: Seq-DeepCpf1 forward pass (scores/deepCpf1/seqdeepcpf1.py:22-92) and
// K5: Azimuth / Rule-Set-2 gradient-boosted-tree evaluation (scores/azimuth/model_comparison.py:
// 507-585 on the features of features/featurization.py).  Both are parameterised by weight arrays
// supplied by the caller: the reference downloads the real parameters at run time.
//
// K6 layout: one workgroup scores 64 guides (lane = guide, wavefront = a quarter of the output
// neurons).  34-mers are one-hot, so the 4->80 k=5 convolution is a gather of 5 weights per
// output (from LDS); ReLU and the pairwise average pool are fused into it; the 1200->80 layer
// consumes each pooled time step as soon as it exists, its weights are wave-uniform and come
// through the scalar cache.  fp32 throughout, fused multiply-adds, fixed summation order
// (ascending flattened index t*80+c, as the reference's flatten(transpose) lays it out).
#include "hawk_bits.h"

#define DC_G 64  // guides per workgroup

__global__ __launch_bounds__(HAWK_BLOCK) void k_deepcpf1(const char* __restrict__ seqs, uint64_t n,
                                                          const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ w3, const float* __restrict__ b3,
                                                          const float* __restrict__ w4, const float* __restrict__ b4,
                                                          float* __restrict__ out, int* status) {
  __shared__ float s_cw[80 * 4 * 5];
  __shared__ float s_cb[80];
  __shared__ float s_x[80][DC_G];    // pooled features of the current time step, [channel][guide]
  __shared__ float s_h[80][DC_G];    // hidden activations between layers
  __shared__ uint8_t s_idx[34][DC_G];
  const int tid = threadIdx.x, gi = tid & (DC_G - 1);
  const int og = __builtin_amdgcn_readfirstlane(tid >> 6);  // wavefront index: which quarter of the outputs
  for (int i = tid; i < 1600; i += HAWK_BLOCK) s_cw[i] = conv_w[i];
  if (tid < 80) s_cb[tid] = conv_b[tid];
  const uint64_t g = (uint64_t)blockIdx.x * DC_G + gi;
  const bool have = g < n;
  for (int j = og; j < 34; j += 4) {  // each wavefront encodes a quarter of the positions
    uint8_t v = 0;
    if (have) {
      switch (seqs[g * 34 + j] & 0xDF) {
        case 'A': v = 0; break; case 'C': v = 1; break; case 'G': v = 2; break; case 'T': v = 3; break;
        default: atomicExch(status, -4); break;  // KeyError in the reference (seqdeepcpf1.py:91)
      }
    }
    s_idx[j][gi] = v;
  }
  __syncthreads();
  float acc[20];
#pragma unroll
  for (int j = 0; j < 20; ++j) acc[j] = b1[og * 20 + j];
  for (int t = 0; t < 15; ++t) {
    // conv + ReLU + avg-pool for channels [og*20, og*20+20) of this guide at time step t
    uint8_t id[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) id[k] = s_idx[2 * t + k][gi];
    for (int cc = 0; cc < 20; ++cc) {
      const int c = og * 20 + cc;
      float a0 = s_cb[c], a1 = s_cb[c];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        a0 += s_cw[(c * 4 + id[k]) * 5 + k];
        a1 += s_cw[(c * 4 + id[k + 1]) * 5 + k];
      }
      s_x[c][gi] = 0.5f * (fmaxf(a0, 0.f) + fmaxf(a1, 0.f));
    }
    __syncthreads();
    // fc1 partial sums: outputs [og*20, og*20+20), inputs t*80 .. t*80+79 in ascending order
    for (int c = 0; c < 80; ++c) {
      const float x = s_x[c][gi];
#pragma unroll
      for (int j = 0; j < 20; ++j) acc[j] = fmaf(w1[(og * 20 + j) * 1200 + t * 80 + c], x, acc[j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 20; ++j) s_h[og * 20 + j][gi] = fmaxf(acc[j], 0.f);
  __syncthreads();
  float h2[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {  // fc2: 80 -> 40, ten outputs per wavefront
    float a = b2[og * 10 + j];
    for (int i = 0; i < 80; ++i) a = fmaf(w2[(og * 10 + j) * 80 + i], s_h[i][gi], a);
    h2[j] = fmaxf(a, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 10; ++j) s_x[og * 10 + j][gi] = h2[j];
  __syncthreads();
  float h3[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {  // fc3: 40 -> 40
    float a = b3[og * 10 + j];
    for (int i = 0; i < 40; ++i) a = fmaf(w3[(og * 10 + j) * 40 + i], s_x[i][gi], a);
    h3[j] = fmaxf(a, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 10; ++j) s_h[og * 10 + j][gi] = h3[j];
  __syncthreads();
  if (og == 0 && have) {  // output: 40 -> 1
    float a = b4[0];
    for (int i = 0; i < 40; ++i) a = fmaf(w4[i], s_h[i][gi], a);
    out[g] = a;
  }
}

void hawk_launch_deepcpf1(hipStream_t st, const char* seqs, uint64_t n, const float* w, float* out, int* status) {
  if (!n) return;
  // packed parameter block: conv_w[1600] conv_b[80] w1[96000] b1[80] w2[3200] b2[40] w3[1600] b3[40] w4[40] b4[1]
  const float* cw = w; const float* cb = cw + 1600; const float* w1 = cb + 80; const float* b1 = w1 + 96000;
  const float* w2 = b1 + 80; const float* b2 = w2 + 3200; const float* w3 = b2 + 40; const float* b3 = w3 + 1600;
  const float* w4 = b3 + 40; const float* b4 = w4 + 40;
  hipLaunchKernelGGL(k_deepcpf1, dim3((uint32_t)((n + DC_G - 1) / DC_G)), dim3(HAWK_BLOCK), 0, st, seqs, n, cw, cb, w1, b1, w2, b2,
                     w3, b3, w4, b4, out, status);
}

// ---------------------------------------------------------------------------------------
// K5: Azimuth (Rule Set 2) = 627 sequence features + gradient-boosted regression trees
// ---------------------------------------------------------------------------------------
// One thread per 30-mer.  The sequence is held as 2-bit codes in the Azimuth alphabet order
// A0 T1 C2 G3 (featurization.py:436-438); features are evaluated on demand where a tree asks for
// them (a depth-3 tree touches 3 of the 627): one-hots are bit tests, the position-independent
// counts are XOR/fold/popcount over the packed word, the four melting temperatures are
// precomputed in fp64 (nearest-neighbour sums + one log each).
__device__ __forceinline__ int az_code(char c) {
  switch (c & 0xDF) { case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3; default: return -1; }
}
__constant__ double c_nnH[16] = {-7.9, -7.2, -8.4, -7.8, -7.2, -7.9, -8.2, -8.5, -8.5, -7.8, -8.0, -10.6, -8.2, -8.4, -9.8, -8.0};
__constant__ double c_nnS[16] = {-22.2, -20.4, -22.4, -21.0, -21.3, -22.2, -22.2, -22.7, -22.7, -21.0, -19.9, -27.2, -22.2, -22.4, -24.4, -19.9};

// Biopython Tm_NN defaults (DNA_NN3, dnac1 = dnac2 = 25, Na = 50, saltcorr = 5) on codes[lo, lo+n)
__device__ double az_tm(uint64_t codes, int lo, int n) {
  double dh = 0.0, ds = 0.0;
  const int e0 = (int)(codes >> (2 * lo)) & 3, e1 = (int)(codes >> (2 * (lo + n - 1))) & 3;
  const int at = (e0 < 2) + (e1 < 2), gc = 2 - at;
  dh += 2.3 * at + 0.1 * gc;
  ds += 4.1 * at + -2.8 * gc;
  for (int i = 0; i + 1 < n; ++i) {
    const int d = (int)(codes >> (2 * (lo + i))) & 15;  // low 2 bits = first base, next 2 = second
    const int idx = (d & 3) * 4 + (d >> 2);
    dh += c_nnH[idx]; ds += c_nnS[idx];
  }
  const double k = (25.0 - 25.0 / 2.0) * 1e-9, R = 1.987;
  ds += 0.368 * (n - 1) * log(50.0 * 1e-3);
  return (1000.0 * dh) / (ds + R * log(k)) - 273.15;
}
// number of positions i < npos with the 2-bit code at i equal to c
__device__ __forceinline__ int az_count1(uint64_t codes, int c, int npos) {
  const uint64_t x = codes ^ (0x5555555555555555ull * (uint64_t)c);
  const uint64_t m = ~(x | (x >> 1)) & 0x5555555555555555ull & ((npos >= 32) ? ~0ull : ((1ull << (2 * npos)) - 1ull));
  return __popcll(m);
}

__device__ double az_feature(int idx, uint64_t codes, int gc, const double (&tm)[4]) {
  if (idx < 120) return ((int)(codes >> (2 * (idx >> 2))) & 3) == (idx & 3) ? 1.0 : 0.0;
  if (idx < 124) return (double)az_count1(codes, idx - 120, 30);
  if (idx < 588) {
    const int j = idx - 124, p = j >> 4, a = j & 15;
    const int d = (int)(codes >> (2 * p)) & 15;
    return ((d & 3) == (a >> 2) && (d >> 2) == (a & 3)) ? 1.0 : 0.0;
  }
  if (idx < 604) {
    const int a = idx - 588;
    const uint64_t x1 = codes ^ (0x5555555555555555ull * (uint64_t)(a >> 2));         // first base == a/4
    const uint64_t x2 = (codes >> 2) ^ (0x5555555555555555ull * (uint64_t)(a & 3));  // next base == a%4
    const uint64_t m = ~(x1 | (x1 >> 1)) & ~(x2 | (x2 >> 1)) & 0x5555555555555555ull & ((1ull << 58) - 1ull);  // 29 positions
    return (double)__popcll(m);
  }
  if (idx == 604) return gc > 10 ? 1.0 : 0.0;
  if (idx == 605) return gc < 10 ? 1.0 : 0.0;
  if (idx == 606) return (double)gc;
  if (idx < 623) {
    const int a = idx - 607;
    return (((int)(codes >> 48) & 3) == (a >> 2) && ((int)(codes >> 54) & 3) == (a & 3)) ? 1.0 : 0.0;  // s[24], s[27]
  }
  return tm[idx - 623];
}

__global__ __launch_bounds__(HAWK_BLOCK) void k_azimuth(const char* __restrict__ seqs, uint64_t n, uint32_t n_trees,
                                                         const int32_t* __restrict__ tree_off, const int32_t* __restrict__ feature,
                                                         const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                         const double* __restrict__ threshold, const double* __restrict__ value,
                                                         double init, double lr, double* __restrict__ out, double* feats_out,
                                                         int* status) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint64_t codes = 0;
  bool bad = false;
  for (int p = 0; p < 30; ++p) {
    const int c = az_code(seqs[i * 30 + p]);
    if (c < 0) bad = true;
    codes |= (uint64_t)(c & 3) << (2 * p);
  }
  if (bad) {  // alphabet.index(nucl) raises ValueError in the reference (featurization.py:471)
    atomicExch(status, -4);
    out[i] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  // GC count of s[4:24]: codes C=2, G=3 have the high bit set
  const int gc = __popcll((codes >> 8) & 0xAAAAAAAAAAull);
  double tm[4];
  tm[0] = az_tm(codes, 0, 30); tm[1] = az_tm(codes, 19, 5); tm[2] = az_tm(codes, 11, 8); tm[3] = az_tm(codes, 6, 5);
  if (feats_out)
    for (int f = 0; f < 627; ++f) feats_out[i * 627 + f] = az_feature(f, codes, gc, tm);
  double acc = init;
  for (uint32_t t = 0; t < n_trees; ++t) {
    const int base = tree_off[t];
    int node = base;
    int f;
    while ((f = feature[node]) >= 0) {
      const double x = (double)(float)az_feature(f, codes, gc, tm);  // sklearn casts X to float32
      node = base + (x <= threshold[node] ? left[node] : right[node]);
    }
    acc += lr * value[node];
  }
  out[i] = acc;
}

// az_tm on its own: sequences of `len` <= 32 bases (A, C, G, T only)
__global__ __launch_bounds__(256) void k_tm_nn(const char* __restrict__ seqs, uint32_t len, uint64_t n, double* __restrict__ out, int* status) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t codes = 0;
  bool bad = false;
  for (uint32_t k = 0; k < len; ++k) {
    const int c = az_code(seqs[i * len + k]);
    bad = bad || c < 0;
    codes |= (uint64_t)(c & 3) << (2 * k);
  }
  if (bad) { atomicExch(status, -4); out[i] = __longlong_as_double(0x7ff8000000000000ll); return; }
  out[i] = az_tm(codes, 0, (int)len);
}
void hawk_launch_tm_nn(hipStream_t st, const char* seqs, uint32_t len, uint64_t n, double* out, int* status) {
  hipLaunchKernelGGL(k_tm_nn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seqs, len, n, out, status);
}
void hawk_launch_azimuth(hipStream_t st, const char* seqs, uint64_t n, uint32_t n_trees, const int32_t* tree_off,
                         const int32_t* feature, const int32_t* left, const int32_t* right, const double* threshold,
                         const double* value, double init, double lr, double* out, double* feats_out, int* status) {
  if (!n) return;
  hipLaunchKernelGGL(k_azimuth, dim3((uint32_t)((n + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, seqs, n, n_trees,
                     tree_off, feature, left, right, threshold, value, init, lr, out, feats_out, status);
}

// ---- generic gradient-boosted-tree evaluation over a caller-supplied feature matrix (RS3: a LightGBM model over
// sglearn features, scores/crisprhawk_scores.py:47-62; the features come from the caller, the trees run here).
// cast_f32: sklearn compares float32(x) <= threshold, LightGBM compares the double as it is.
__global__ __launch_bounds__(HAWK_BLOCK) void k_gbt(const double* __restrict__ feats, uint64_t n, uint32_t nf, uint32_t n_trees,
                                                     const int32_t* __restrict__ tree_off, const int32_t* __restrict__ feature,
                                                     const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                     const double* __restrict__ threshold, const double* __restrict__ value, double init,
                                                     double lr, int cast_f32, double* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * HAWK_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double* x = feats + i * nf;
  double acc = init;
  for (uint32_t t = 0; t < n_trees; ++t) {
    const int base = tree_off[t];
    int node = base, f;
    while ((f = feature[node]) >= 0) {
      const double v = cast_f32 ? (double)(float)x[f] : x[f];
      node = base + (v <= threshold[node] ? left[node] : right[node]);
    }
    acc += lr * value[node];
  }
  out[i] = acc;
}
void hawk_launch_gbt(hipStream_t st, const double* feats, uint64_t n, uint32_t nf, uint32_t n_trees, const int32_t* tree_off,
                     const int32_t* feature, const int32_t* left, const int32_t* right, const double* threshold, const double* value,
                     double init, double lr, int cast_f32, double* out) {
  if (!n) return;
  hipLaunchKernelGGL(k_gbt, dim3((uint32_t)((n + HAWK_BLOCK - 1) / HAWK_BLOCK)), dim3(HAWK_BLOCK), 0, st, feats, n, nf, n_trees, tree_off,
                     feature, left, right, threshold, value, init, lr, cast_f32, out);
}
