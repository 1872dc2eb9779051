// hawk_api_vcf.hip - C ABI: VCF genotypes on the device (f3)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "hawk_host.h"

extern "C" {

int hawk_gt_parse(hawk_ctx* ctx, const uint8_t* text, uint64_t text_len, const uint64_t* line_off, const uint64_t* gt_off,
                  uint64_t n_lines, uint32_t n_samples, hawk_gt** out, float* kernel_ms) {
  if (!ctx || !out || !n_samples || (n_lines && (!text || !line_off || !gt_off))) return HAWK_E_INVALID;
  // every offset the kernel dereferences is checked here
  for (uint64_t i = 0; i < n_lines; ++i)
    if (line_off[i + 1] > text_len || line_off[i] >= line_off[i + 1] || gt_off[i] < line_off[i] || gt_off[i] > line_off[i + 1])
      return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_gt* g = new (std::nothrow) hawk_gt();
  if (!g) return HAWK_E_INVALID;
  g->ctx = ctx; g->n_lines = n_lines; g->n_samples = n_samples; g->n_var = 0; g->n_entries = 0;
  g->d_codes = nullptr; g->d_flags = nullptr; g->d_col_off = nullptr; g->d_idx = nullptr; g->d_o = nullptr; g->d_delta = nullptr;
  if (kernel_ms) *kernel_ms = 0.f;
  const size_t ncode = std::max<size_t>((size_t)n_lines * 2 * n_samples, 1);
  POOLCHK(&g->d_codes, ncode); POOLCHK(&g->d_flags, std::max<size_t>(n_lines, 1));
  if (n_lines) {
    uint8_t* d_text = nullptr; uint64_t *d_lo = nullptr, *d_go = nullptr;
    PoolScope tmp;
    TEMPCHK(tmp, &d_text, text_len); TEMPCHK(tmp, &d_lo, (n_lines + 1) * 8); TEMPCHK(tmp, &d_go, n_lines * 8);
    hipStream_t st = ctx->stream;
    HIPCHK(hipMemcpyAsync(d_text, text, text_len, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_lo, line_off, (n_lines + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_go, gt_off, n_lines * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(g->d_codes, 0xff, ncode, st));  // samples a short record does not reach read as missing
    HIPCHK(hipEventRecord(ctx->ev[0], st));
    hawk_launch_gt_parse(st, d_text, d_lo, d_go, n_lines, n_samples, g->d_codes, g->d_flags);
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
  }
  *out = g;
  return HAWK_OK;
}

int hawk_gt_from_codes(hawk_ctx* ctx, const uint8_t* codes, uint64_t n_lines, uint32_t n_samples, hawk_gt** out) {
  if (!ctx || !out || !n_samples || (n_lines && !codes)) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_gt* g = new (std::nothrow) hawk_gt();
  if (!g) return HAWK_E_INVALID;
  g->ctx = ctx; g->n_lines = n_lines; g->n_samples = n_samples; g->n_var = 0; g->n_entries = 0;
  g->d_codes = nullptr; g->d_flags = nullptr; g->d_col_off = nullptr; g->d_idx = nullptr; g->d_o = nullptr; g->d_delta = nullptr;
  const size_t ncode = std::max<size_t>((size_t)n_lines * 2 * n_samples, 1);
  POOLCHK(&g->d_codes, ncode); POOLCHK(&g->d_flags, std::max<size_t>(n_lines, 1));
  if (n_lines) {
    HIPCHK(hipMemcpyAsync(g->d_codes, codes, (size_t)n_lines * 2 * n_samples, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(g->d_flags, 0, n_lines, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  *out = g;
  return HAWK_OK;
}

void hawk_gt_destroy(hawk_gt* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  hawk_pool_free(g->d_codes); hawk_pool_free(g->d_flags);
  if (g->d_col_off) hawk_pool_free(g->d_col_off);
  if (g->d_idx) hawk_pool_free(g->d_idx);
  if (g->d_o) hawk_pool_free(g->d_o);
  if (g->d_delta) hawk_pool_free(g->d_delta);
  if (g->d_indel) hawk_pool_free(g->d_indel);
  if (g->d_ioff) hawk_pool_free(g->d_ioff);
  delete g;
}

int hawk_gt_codes(hawk_gt* g, uint8_t* codes, uint8_t* line_flags) {
  if (!g) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (codes && g->n_lines) HIPCHK(hipMemcpyAsync(codes, g->d_codes, (size_t)g->n_lines * 2 * g->n_samples, hipMemcpyDefault, g->ctx->stream));
  if (line_flags && g->n_lines) HIPCHK(hipMemcpyAsync(line_flags, g->d_flags, g->n_lines, hipMemcpyDefault, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

int hawk_gt_lists(hawk_gt* g, const uint32_t* var_line, const uint8_t* var_allele, const int32_t* var_r0, const int32_t* var_chain,
                  uint32_t n_var, uint64_t* col_off, int64_t* col_delta, float* kernel_ms) {
  if (!g || !col_off || (n_var && (!var_line || !var_allele || !var_r0 || !var_chain))) return HAWK_E_INVALID;
  for (uint32_t j = 0; j < n_var; ++j)
    if (var_line[j] >= g->n_lines || var_allele[j] == 0 || var_allele[j] == 255) return HAWK_E_INVALID;
  hawk_ctx* ctx = g->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t n_cols = 2 * g->n_samples, n_chunk = (n_var + 63u) / 64u;
  hipStream_t st = ctx->stream;
  if (g->d_col_off) { hawk_pool_free(g->d_col_off); g->d_col_off = nullptr; }
  if (g->d_idx) { hawk_pool_free(g->d_idx); g->d_idx = nullptr; }
  if (g->d_o) { hawk_pool_free(g->d_o); g->d_o = nullptr; }
  if (g->d_delta) { hawk_pool_free(g->d_delta); g->d_delta = nullptr; }
  if (g->d_indel) { hawk_pool_free(g->d_indel); g->d_indel = nullptr; }
  if (g->d_ioff) { hawk_pool_free(g->d_ioff); g->d_ioff = nullptr; }
  g->h_off.assign(n_cols + 1, 0); g->h_ioff.assign(n_cols + 1, 0); g->h_delta.assign(n_cols, 0);
  g->n_var = n_var; g->n_entries = 0; g->n_indel = 0;
  if (kernel_ms) *kernel_ms = 0.f;
  std::vector<uint64_t> off(n_cols + 1, 0);
  if (n_var == 0) {
    memcpy(col_off, off.data(), (n_cols + 1) * 8);
    if (col_delta) memset(col_delta, 0, (size_t)n_cols * 8);
    return HAWK_OK;
  }
  uint32_t *d_vl = nullptr, *d_cnt = nullptr; uint8_t* d_va = nullptr; int32_t *d_r0 = nullptr, *d_ch = nullptr;
  unsigned long long* d_bal = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_vl, (size_t)n_var * 4); TEMPCHK(tmp, &d_va, n_var); TEMPCHK(tmp, &d_r0, (size_t)n_var * 4);
  TEMPCHK(tmp, &d_ch, (size_t)n_var * 4); TEMPCHK(tmp, &d_cnt, (size_t)n_cols * 2 * 4);
  POOLCHK(&g->d_ioff, (size_t)(n_cols + 1) * 8);
  uint64_t* d_ioff = g->d_ioff;
  TEMPCHK(tmp, &d_bal, (size_t)n_cols * n_chunk * 8);
  POOLCHK(&g->d_col_off, (size_t)(n_cols + 1) * 8); POOLCHK(&g->d_delta, (size_t)n_cols * 8);
  HIPCHK(hipMemcpyAsync(d_vl, var_line, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_va, var_allele, n_var, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r0, var_r0, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ch, var_chain, (size_t)n_var * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hawk_launch_gt_count(st, g->d_codes, n_cols, d_vl, d_va, d_ch, n_var, d_bal, d_cnt);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  std::vector<uint32_t> cnt(2 * (size_t)n_cols);
  HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)n_cols * 2 * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<uint64_t> ioff(n_cols + 1, 0);
  for (uint32_t c = 0; c < n_cols; ++c) {  // 2 * n_samples values: host prefix sums
    off[c + 1] = off[c] + cnt[c];
    ioff[c + 1] = ioff[c] + cnt[n_cols + c];
  }
  const uint64_t ne = off[n_cols], ni = ioff[n_cols];
  POOLCHK(&g->d_idx, std::max<size_t>(ne, 1) * 4); POOLCHK(&g->d_o, std::max<size_t>(ne, 1) * 4);
  POOLCHK(&g->d_indel, std::max<size_t>(ni, 1) * 4);
  HIPCHK(hipMemcpyAsync(g->d_col_off, off.data(), (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_ioff, ioff.data(), (size_t)(n_cols + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  hawk_launch_gt_fill(st, n_cols, d_r0, d_ch, n_var, d_bal, g->d_col_off, g->d_idx, g->d_o, g->d_delta, d_ioff, g->d_indel);
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(g->h_delta.data(), g->d_delta, (size_t)n_cols * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (col_delta) memcpy(col_delta, g->h_delta.data(), (size_t)n_cols * 8);
  g->h_off = off; g->h_ioff = ioff;
  if (kernel_ms) {
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[1]); (void)hipEventElapsedTime(&b, ctx->ev[2], ctx->ev[3]);
    *kernel_ms = a + b;
  }
  memcpy(col_off, off.data(), (size_t)(n_cols + 1) * 8);
  g->n_entries = ne;
  g->n_indel = ni;
  return HAWK_OK;
}

int hawk_gt_lists_indels(hawk_gt* g, uint32_t* entry_idx, uint64_t cap, uint64_t* n_indel) {
  if (!g || !n_indel) return HAWK_E_INVALID;
  *n_indel = g->n_indel;
  const uint64_t k = std::min<uint64_t>(cap, g->n_indel);
  if (!k || !entry_idx) return HAWK_OK;
  HIPCHK(hipSetDevice(g->ctx->device));
  HIPCHK(hipMemcpyAsync(entry_idx, g->d_indel, k * 4, hipMemcpyDefault, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

int hawk_gt_lists_download(hawk_gt* g, uint32_t* hv_idx, int32_t* hv_o) {
  if (!g) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(g->ctx->device));
  if (g->n_entries) {
    if (hv_idx) HIPCHK(hipMemcpyAsync(hv_idx, g->d_idx, g->n_entries * 4, hipMemcpyDefault, g->ctx->stream));
    if (hv_o) HIPCHK(hipMemcpyAsync(hv_o, g->d_o, g->n_entries * 4, hipMemcpyDefault, g->ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  return HAWK_OK;
}

}  // extern "C"
