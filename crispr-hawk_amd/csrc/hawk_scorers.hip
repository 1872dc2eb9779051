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
