// hawk_comm.hip — the one exchange of a multi-GPU job: every rank's guide table to one rank over
// RCCL (xGMI).  One process per GPU; haplotypes are sharded with no data-path collective, so this file is the only
// place a collective library is touched.  librccl.so is opened lazily (dlopen) so that a single-GPU host without
// RCCL can still load libhawk_hip.so; the Python side (crisprhawk_hip/parallel.py) carries the 128-byte unique id
// from rank 0 to the other ranks over its own TCP rendezvous and then calls straight into these entry points.
//
// All RCCL work is enqueued on the context's stream, behind the kernels that produced the columns: no extra
// synchronisation between search and exchange.  A direct all-to-one (grouped ncclSend / ncclRecv) is used rather
// than a padded all-gather: xGMI is point to point, every peer has its own link into the destination, and the
// tables differ in length.
#include <dlfcn.h>
#include <unistd.h>

#include <cstring>
#include <new>
#include <vector>

#include "hawk_host.h"

namespace {
// the slice of rccl.h this file needs (rccl.h:40-43, 187, 220, 260, 339, 460-463, 678, 700, 715)
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclUint8 = 1, kNcclUint64 = 5 };
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
};
Rccl g_rccl;
thread_local char g_comm_err[256] = "";

// RCCL prints a version banner on STDOUT when its first communicator comes up ("RCCL version : ...", five lines).  A
// caller whose stdout is a protocol (bench.py prints exactly one JSON line) must not get that: while RCCL initialises,
// file descriptor 1 points at stderr.
struct StdoutToStderr {
  int saved;
  StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
  ~StdoutToStderr() { fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

int rccl_load() {
  if (g_rccl.h) return HAWK_OK;
  const char* names[] = {getenv("HAWK_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if (n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) { snprintf(g_comm_err, sizeof(g_comm_err), "cannot load librccl.so: %s", dlerror()); return HAWK_E_COMM; }
  Rccl r;
  r.h = h;
#define SYM(field, name)                                                                     \
  *(void**)(&r.field) = dlsym(h, name);                                                      \
  if (!r.field) { snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so lacks %s", name); return HAWK_E_COMM; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(GetErrorString, "ncclGetErrorString") SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
  g_rccl = r;
  return HAWK_OK;
}
}  // namespace

#define NCCLCHK(expr)                                                                                              \
  do {                                                                                                             \
    int r_ = (expr);                                                                                               \
    if (r_ != 0) {                                                                                                 \
      snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return HAWK_E_COMM;                                                                                          \
    }                                                                                                              \
  } while (0)

struct hawk_comm {
  hawk_ctx* ctx;
  int world, rank;
  NcclComm nccl;
  DevBuf scratch, scratch2;
};

// hap column of a gathered slice: local haplotype 0 is REF on every rank and stays 0, the others move to the
// rank's block of the global haplotype list
__global__ __launch_bounds__(256) void k_hap_shift(uint32_t* __restrict__ hap, uint64_t n, uint32_t offset) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const uint32_t h = hap[i]; hap[i] = h ? h + offset : 0u; }
}

extern "C" {

const char* hawk_comm_last_error(void) { return g_comm_err; }

int hawk_comm_unique_id(uint8_t* id128) {
  if (!id128) return HAWK_E_INVALID;
  int rc = rccl_load();
  if (rc) return rc;
  NcclId id;
  {
    StdoutToStderr quiet;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != 0) { snprintf(g_comm_err, sizeof(g_comm_err), "ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); return HAWK_E_COMM; }
  }
  memcpy(id128, id.internal, 128);
  return HAWK_OK;
}

int hawk_comm_init(hawk_ctx* ctx, int world, int rank, const uint8_t* id128, hawk_comm** out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return HAWK_E_INVALID;
  int rc = rccl_load();
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  hawk_comm* c = new (std::nothrow) hawk_comm();
  if (!c) return HAWK_E_INVALID;
  c->ctx = ctx; c->world = world; c->rank = rank; c->nccl = nullptr;
  NcclId id;
  memcpy(id.internal, id128, 128);
  int r;
  {
    StdoutToStderr quiet;
    r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
  }
  if (r != 0) {
    snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank(world %d, rank %d): %s", world, rank, g_rccl.GetErrorString(r));
    delete c;
    return HAWK_E_COMM;
  }
  *out = c;
  return HAWK_OK;
}

void hawk_comm_destroy(hawk_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  c->scratch.release(); c->scratch2.release();
  delete c;
}

int hawk_comm_allgather_u64(hawk_comm* c, const uint64_t* mine, uint32_t k, uint64_t* all) {
  if (!c || !mine || !all || !k) return HAWK_E_INVALID;
  hawk_ctx* ctx = c->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if ((rc = c->scratch.reserve((size_t)k * 8)) || (rc = c->scratch2.reserve((size_t)k * 8 * c->world))) return rc;
  HIPCHK(hipMemcpyAsync(c->scratch.p, mine, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
  NCCLCHK(g_rccl.AllGather(c->scratch.p, c->scratch2.p, k, kNcclUint64, c->nccl, ctx->stream));
  HIPCHK(hipMemcpyAsync(all, c->scratch2.p, (size_t)k * 8 * c->world, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return HAWK_OK;
}

int hawk_comm_gatherv(hawk_comm* c, const void* send, uint64_t send_bytes, int send_on_device, void* recv,
                      const uint64_t* recv_off, int recv_on_device, int dst) {
  if (!c || dst < 0 || dst >= c->world || (send_bytes && !send)) return HAWK_E_INVALID;
  const bool root = c->rank == dst;
  if (root && (!recv_off || (recv_off[c->world] && !recv))) return HAWK_E_INVALID;
  hawk_ctx* ctx = c->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const void* s_dev = send;
  int rc;
  if (!send_on_device && send_bytes) {
    if ((rc = c->scratch.reserve(send_bytes))) return rc;
    HIPCHK(hipMemcpyAsync(c->scratch.p, send, send_bytes, hipMemcpyHostToDevice, st));
    s_dev = c->scratch.p;
  }
  char* r_dev = (char*)recv;
  if (root) {
    if (recv_off[c->rank + 1] - recv_off[c->rank] != send_bytes) return HAWK_E_INVALID;
    if (!recv_on_device) {
      if ((rc = c->scratch2.reserve(std::max<uint64_t>(recv_off[c->world], 1)))) return rc;
      r_dev = c->scratch2.as<char>();
    }
  }
  NCCLCHK(g_rccl.GroupStart());
  if (!root) {
    if (send_bytes) NCCLCHK(g_rccl.Send(s_dev, send_bytes, kNcclUint8, dst, c->nccl, st));
  } else {
    for (int r = 0; r < c->world; ++r) {
      const uint64_t nb = recv_off[r + 1] - recv_off[r];
      if (r != dst && nb) NCCLCHK(g_rccl.Recv(r_dev + recv_off[r], nb, kNcclUint8, r, c->nccl, st));
    }
  }
  NCCLCHK(g_rccl.GroupEnd());
  if (root) {
    if (send_bytes) HIPCHK(hipMemcpyAsync(r_dev + recv_off[dst], s_dev, send_bytes, hipMemcpyDeviceToDevice, st));
    if (!recv_on_device && recv_off[c->world])
      HIPCHK(hipMemcpyAsync(recv, r_dev, recv_off[c->world], hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return HAWK_OK;
}

int hawk_table_gather(hawk_comm* c, hawk_table* t, uint32_t hap_offset, int dst, hawk_table** merged, float* ms) {
  if (!c || !t || dst < 0 || dst >= c->world || hawk_table_stale(t)) return HAWK_E_INVALID;
  const bool root = c->rank == dst;
  if (root && !merged) return HAWK_E_INVALID;
  hawk_ctx* ctx = c->ctx;
  if (t->ctx != ctx) return HAWK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int W = c->world;
  // directory: rows, haplotype offset, candidates, hits of every rank
  uint64_t mine[4] = {t->n_rows, hap_offset, t->n_cand, t->n_hits};
  std::vector<uint64_t> all((size_t)4 * W);
  int rc = hawk_comm_allgather_u64(c, mine, 4, all.data());
  if (rc) return rc;
  std::vector<uint64_t> off(W + 1, 0);
  for (int r = 0; r < W; ++r) off[r + 1] = off[r] + all[4 * r];
  const uint64_t total = off[W];
  hawk_table* m = nullptr;
  GuideCols mc = {};
  if (root) {
    m = new (std::nothrow) hawk_table();
    if (!m) return HAWK_E_INVALID;
    m->hs = nullptr; m->ctx = ctx; m->gen = 0;
    m->n_rows = total; m->n_cand = 0; m->n_hits = 0;
    for (int r = 0; r < W; ++r) { m->n_cand += all[4 * r + 2]; m->n_hits += all[4 * r + 3]; }
    m->guidelen = t->guidelen; m->pamlen = t->pamlen; m->right = t->right; m->n_groups = 0; m->collapsed = false;
    if ((rc = hawk_reserve_cols(m->own, std::max<uint64_t>(total, 1), &mc))) { hawk_table_destroy(m); return rc; }
    m->cols = mc; m->cap = mc.cap;
  }
  const GuideCols& sc = t->cols;
  const uint64_t n = t->n_rows;
  // column pointers and widths: hap pos strand start stop flags cfdon win[0..4]
  const void* sp[12] = {sc.hap, sc.pos, sc.strand, sc.start, sc.stop, sc.flags, sc.cfdon, sc.win, sc.win + sc.cap, sc.win + 2 * sc.cap,
                        sc.win + 3 * sc.cap, sc.win + 4 * sc.cap};
  void* rp[12] = {mc.hap, mc.pos, mc.strand, mc.start, mc.stop, mc.flags, mc.cfdon, mc.win, mc.win + mc.cap, mc.win + 2 * mc.cap,
                  mc.win + 3 * mc.cap, mc.win + 4 * mc.cap};
  const size_t wd[12] = {4, 4, 1, 8, 8, 1, 8, 8, 8, 8, 8, 8};
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  int r0 = g_rccl.GroupStart();
  if (r0 == 0) {
    for (int k = 0; k < 12 && r0 == 0; ++k) {
      if (!root) {
        if (n) r0 = g_rccl.Send(sp[k], n * wd[k], kNcclUint8, dst, c->nccl, st);
      } else {
        for (int r = 0; r < W && r0 == 0; ++r) {
          const uint64_t nr = all[4 * r];
          if (r != dst && nr) r0 = g_rccl.Recv((char*)rp[k] + off[r] * wd[k], nr * wd[k], kNcclUint8, r, c->nccl, st);
        }
      }
    }
    const int r1 = g_rccl.GroupEnd();
    if (r0 == 0) r0 = r1;
  }
  if (r0 != 0) {
    snprintf(g_comm_err, sizeof(g_comm_err), "hawk_table_gather: %s", g_rccl.GetErrorString(r0));
    if (m) hawk_table_destroy(m);
    return HAWK_E_COMM;
  }
  if (root) {
    for (int k = 0; k < 12; ++k)
      if (n) HIPCHK(hipMemcpyAsync((char*)rp[k] + off[dst] * wd[k], sp[k], n * wd[k], hipMemcpyDeviceToDevice, st));
    for (int r = 0; r < W; ++r) {
      const uint64_t nr = all[4 * r];
      const uint32_t ho = (uint32_t)all[4 * r + 1];
      if (nr && ho) hipLaunchKernelGGL(k_hap_shift, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, mc.hap + off[r], nr, ho);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  HIPCHK(hipStreamSynchronize(st));
  if (ms) (void)hipEventElapsedTime(ms, ctx->ev[6], ctx->ev[7]);
  if (root) *merged = m;
  return HAWK_OK;
}

}  // extern "C"
