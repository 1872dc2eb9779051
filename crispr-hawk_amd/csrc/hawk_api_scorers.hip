// hawk_api_scorers.hip - C ABI: the scorers (GBT over supplied features, CFD triples, DeepCpf1, Tm_NN, Azimuth)
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

// ---------------------------------------------------------------------------- fused search
// ---------------------------------------------------------------------------- generic GBT over supplied features (RS3)
int hawk_gbt_predict(hawk_ctx* ctx, const double* feats, uint64_t n, uint32_t n_features, const hawk_gbt_model* m, int cast_f32,
                     double* out) {
  if (!ctx || !m || !m->tree_off || !m->feature || !m->left || !m->right || !m->threshold || !m->value || !n_features ||
      (n && (!feats || !out)))
    return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  for (uint32_t t = 0; t < m->n_trees; ++t) {  // every child index inside its tree and pointing forward, every feature in range
    const int32_t lo = m->tree_off[t], hi = m->tree_off[t + 1];
    if (lo < 0 || hi <= lo || (uint32_t)hi > m->n_nodes) return HAWK_E_INVALID;
    for (int32_t k = lo; k < hi; ++k) {
      if (m->feature[k] >= (int32_t)n_features) return HAWK_E_INVALID;
      if (m->feature[k] >= 0 && (m->left[k] <= k - lo || m->right[k] <= k - lo || m->left[k] >= hi - lo || m->right[k] >= hi - lo))
        return HAWK_E_INVALID;
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nn = m->n_nodes, nt = m->n_trees;
  double *d_x = nullptr, *d_th = nullptr, *d_v = nullptr, *d_o = nullptr;
  int32_t *d_off = nullptr, *d_f = nullptr, *d_l = nullptr, *d_r = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_x, n * n_features * 8); TEMPCHK(tmp, &d_off, (nt + 1) * 4); TEMPCHK(tmp, &d_f, nn * 4); TEMPCHK(tmp, &d_l, nn * 4); TEMPCHK(tmp, &d_r, nn * 4);
  TEMPCHK(tmp, &d_th, nn * 8); TEMPCHK(tmp, &d_v, nn * 8); TEMPCHK(tmp, &d_o, n * 8);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_x, feats, n * n_features * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, m->tree_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_f, m->feature, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_l, m->left, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r, m->right, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_th, m->threshold, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_v, m->value, nn * 8, hipMemcpyHostToDevice, st));
  hawk_launch_gbt(st, d_x, n, n_features, m->n_trees, d_off, d_f, d_l, d_r, d_th, d_v, m->init, m->learning_rate, cast_f32, d_o);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return HAWK_OK;
}

int hawk_cfd(hawk_ctx* ctx, const char* wt, const char* sg, uint32_t len, const char* pam2, uint64_t n,
             const double* cfd_mm, const double* cfd_pam, double* out) {
  if (!ctx || !cfd_mm || !cfd_pam || (n && (!wt || !sg || !pam2 || !out)) || len == 0) return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  char *d_wt = nullptr, *d_sg = nullptr, *d_p = nullptr;
  double *d_tab = nullptr, *d_out = nullptr;
  int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_wt, n * len); TEMPCHK(tmp, &d_sg, n * len); TEMPCHK(tmp, &d_p, n * 2);
  TEMPCHK(tmp, &d_tab, 336 * 8); TEMPCHK(tmp, &d_out, n * 8); TEMPCHK(tmp, &d_status, 4);
  HIPCHK(hipMemcpyAsync(d_wt, wt, n * len, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_sg, sg, n * len, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_p, pam2, n * 2, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_tab, cfd_mm, 320 * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_tab + 320, cfd_pam, 16 * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
  hawk_launch_cfd(ctx->stream, d_wt, d_sg, len, d_p, n, d_tab, d_tab + 320, d_out, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return status;
}
// ---------------------------------------------------------------------------- K6 DeepCpf1
int hawk_deepcpf1(hawk_ctx* ctx, const char* seqs34, uint64_t n, const float* weights, float* out) {
  if (!ctx || !weights || (n && (!seqs34 || !out))) return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nw = HAWK_DEEPCPF1_NPARAMS;
  char* d_s = nullptr; float *d_w = nullptr, *d_o = nullptr; int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_s, n * 34); TEMPCHK(tmp, &d_w, nw * 4); TEMPCHK(tmp, &d_o, n * 4); TEMPCHK(tmp, &d_status, 4);
  HIPCHK(hipMemcpyAsync(d_s, seqs34, n * 34, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(d_w, weights, nw * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, ctx->stream));
  hawk_launch_deepcpf1(ctx->stream, d_s, n, d_w, d_o, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return status;
}

// ---------------------------------------------------------------------------- K5 Azimuth
int hawk_tm_nn(hawk_ctx* ctx, const char* seqs, uint32_t len, uint64_t n, double* out) {
  if (!ctx || (n && (!seqs || !out))) return HAWK_E_INVALID;
  if (len < 2 || len > 32) return HAWK_E_UNSUPPORTED;
  if (!n) return HAWK_OK;
  HIPCHK(hipSetDevice(ctx->device));
  PoolScope tmp;
  char* d_s; double* d_o; int* d_status;
  TEMPCHK(tmp, &d_s, n * len); TEMPCHK(tmp, &d_o, n * 8); TEMPCHK(tmp, &d_status, 4);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_s, seqs, n * len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, st));
  hawk_launch_tm_nn(st, d_s, len, n, d_o, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return status;
}

int hawk_azimuth(hawk_ctx* ctx, const char* seqs30, uint64_t n, const hawk_gbt_model* m, double* out, double* feats_out) {
  if (!ctx || !m || !m->tree_off || !m->feature || !m->left || !m->right || !m->threshold || !m->value ||
      (n && (!seqs30 || !out)))
    return HAWK_E_INVALID;
  if (!n) return HAWK_OK;
  // validate the trees on the host: every child index inside its tree, every feature < 627
  for (uint32_t t = 0; t < m->n_trees; ++t) {
    const int32_t lo = m->tree_off[t], hi = m->tree_off[t + 1];
    if (lo < 0 || hi <= lo || (uint32_t)hi > m->n_nodes) return HAWK_E_INVALID;
    for (int32_t k = lo; k < hi; ++k) {
      if (m->feature[k] >= 627) return HAWK_E_INVALID;
      if (m->feature[k] >= 0 && (m->left[k] <= k - lo || m->right[k] <= k - lo || m->left[k] >= hi - lo || m->right[k] >= hi - lo))
        return HAWK_E_INVALID;  // children must point forward inside the tree: traversal terminates
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nn = m->n_nodes, nt = m->n_trees;
  char* d_s = nullptr; int32_t *d_off = nullptr, *d_f = nullptr, *d_l = nullptr, *d_r = nullptr;
  double *d_th = nullptr, *d_v = nullptr, *d_o = nullptr, *d_fo = nullptr; int* d_status = nullptr;
  PoolScope tmp;
  TEMPCHK(tmp, &d_s, n * 30); TEMPCHK(tmp, &d_off, (nt + 1) * 4); TEMPCHK(tmp, &d_f, nn * 4);
  TEMPCHK(tmp, &d_l, nn * 4); TEMPCHK(tmp, &d_r, nn * 4); TEMPCHK(tmp, &d_th, nn * 8); TEMPCHK(tmp, &d_v, nn * 8);
  TEMPCHK(tmp, &d_o, n * 8); TEMPCHK(tmp, &d_status, 4);
  if (feats_out) TEMPCHK(tmp, &d_fo, n * 627 * 8);
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_s, seqs30, n * 30, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, m->tree_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_f, m->feature, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_l, m->left, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_r, m->right, nn * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_th, m->threshold, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_v, m->value, nn * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_status, 0, 4, st));
  hawk_launch_azimuth(st, d_s, n, m->n_trees, d_off, d_f, d_l, d_r, d_th, d_v, m->init, m->learning_rate, d_o, d_fo, d_status);
  HIPCHK(hipGetLastError());
  int status = 0;
  HIPCHK(hipMemcpyAsync(out, d_o, n * 8, hipMemcpyDeviceToHost, st));
  if (feats_out) HIPCHK(hipMemcpyAsync(feats_out, d_fo, n * 627 * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return status;
}


}  // extern "C"
