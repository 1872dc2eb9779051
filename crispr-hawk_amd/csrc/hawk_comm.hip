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
  // a failure inside the bracket must not leave the group open (later RCCL calls of this thread would queue into it and
  // never run): the status is carried to behind ncclGroupEnd
  int r0 = g_rccl.GroupStart();
  if (r0 == 0) {
    if (!root) {
      if (send_bytes) r0 = g_rccl.Send(s_dev, send_bytes, kNcclUint8, dst, c->nccl, st);
    } else {
      for (int r = 0; r < c->world && r0 == 0; ++r) {
        const uint64_t nb = recv_off[r + 1] - recv_off[r];
        if (r != dst && nb) r0 = g_rccl.Recv(r_dev + recv_off[r], nb, kNcclUint8, r, c->nccl, st);
      }
    }
    const int r1 = g_rccl.GroupEnd();
    if (r0 == 0) r0 = r1;
  }
  if (r0 != 0) {
    snprintf(g_comm_err, sizeof(g_comm_err), "hawk_comm_gatherv: %s", g_rccl.GetErrorString(r0));
    return HAWK_E_COMM;
  }
  if (root) {
    if (send_bytes) HIPCHK(hipMemcpyAsync(r_dev + recv_off[dst], s_dev, send_bytes, hipMemcpyDeviceToDevice, st));
    if (!recv_on_device && recv_off[c->world])
      HIPCHK(hipMemcpyAsync(recv, r_dev, recv_off[c->world], hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return HAWK_OK;
}

// The arithmetic of hawk_table_gather without any device or RCCL call (so that it can be driven on a CPU with any
// transport, tests/test_parallel_gloo.py): from the directory every rank contributed - {rows, haplotype offset, candidates,
// hits} per rank - the row offset of every rank's slice in the merged table, the totals, and the transfers of this rank:
// one (column, peer, byte offset in the merged column, bytes) per column and peer for the destination (recv; its own
// slice is a local copy and is listed with peer == dst), one per column for a sender (offset 0 = start of its own column).
static const size_t kColWidth[HAWK_GATHER_COLS] = {4, 4, 1, 8, 8, 1, 8, 8, 8, 8, 8, 8};  // hap pos strand start stop flags cfdon win[0..4]
int hawk_host_gather_plan(int world, int rank, int dst, const uint64_t* dir4, uint64_t* row_off, uint64_t* totals3,
                          hawk_gather_op* ops, uint32_t cap, uint32_t* n_ops) {
  if (world < 1 || rank < 0 || rank >= world || dst < 0 || dst >= world || !dir4 || !row_off || !totals3 || !n_ops) return HAWK_E_INVALID;
  row_off[0] = 0;
  totals3[1] = totals3[2] = 0;
  for (int r = 0; r < world; ++r) {
    row_off[r + 1] = row_off[r] + dir4[4 * r];
    totals3[1] += dir4[4 * r + 2];
    totals3[2] += dir4[4 * r + 3];
  }
  totals3[0] = row_off[world];
  uint32_t k_ops = 0;
  for (int k = 0; k < HAWK_GATHER_COLS; ++k) {
    if (rank != dst) {
      const uint64_t n = dir4[4 * rank];
      if (n) {
        if (ops && k_ops < cap) ops[k_ops] = hawk_gather_op{(uint32_t)k, (uint32_t)dst, 0, n * kColWidth[k]};
        ++k_ops;
      }
    } else {
      for (int r = 0; r < world; ++r) {
        const uint64_t n = dir4[4 * r];
        if (!n) continue;
        if (ops && k_ops < cap) ops[k_ops] = hawk_gather_op{(uint32_t)k, (uint32_t)r, row_off[r] * kColWidth[k], n * kColWidth[k]};
        ++k_ops;
      }
    }
  }
  *n_ops = k_ops;
  return (ops && k_ops > cap) ? HAWK_E_CAPACITY : HAWK_OK;
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
  uint64_t totals[3];
  std::vector<hawk_gather_op> ops((size_t)HAWK_GATHER_COLS * W);
  uint32_t n_ops = 0;
  if ((rc = hawk_host_gather_plan(W, c->rank, dst, all.data(), off.data(), totals, ops.data(), (uint32_t)ops.size(), &n_ops))) return rc;
  hawk_table* m = nullptr;
  GuideCols mc = {};
  int prep = HAWK_OK;  // what this rank has to say before any send is posted
  GuideCols sc = t->cols;
  const uint64_t n = t->n_rows;
  PoolScope tmp;  // a table in packed rows travels as columns (the merged table is columnar): cut them out first
  if (sc.rows) {
    GuideCols u;
    memset(&u, 0, sizeof(u));
    u.cap = std::max<uint64_t>(n, 1);
    int ra = tmp.alloc((void**)&u.hap, u.cap * 4);
    if (!ra) ra = tmp.alloc((void**)&u.pos, u.cap * 4);
    if (!ra) ra = tmp.alloc((void**)&u.strand, u.cap);
    if (!ra) ra = tmp.alloc((void**)&u.start, u.cap * 8);
    if (!ra) ra = tmp.alloc((void**)&u.stop, u.cap * 8);
    if (!ra) ra = tmp.alloc((void**)&u.flags, u.cap);
    if (!ra) ra = tmp.alloc((void**)&u.cfdon, u.cap * 8);
    if (!ra) ra = tmp.alloc((void**)&u.win, u.cap * 8 * HAWK_PLANES);
    if (!ra) hawk_launch_rows_unpack(st, sc.rows, n, sc.startp, u);
    sc = u;
    if (ra) prep = ra;  // said in the go / no-go word below
  }
  if (root) {
    m = new (std::nothrow) hawk_table();
    if (!m && !prep) prep = HAWK_E_INVALID;
    if (m) {
      m->hs = nullptr; m->ctx = ctx; m->gen = 0;
      m->n_rows = totals[0]; m->n_cand = totals[1]; m->n_hits = totals[2];
      m->guidelen = t->guidelen; m->pamlen = t->pamlen; m->right = t->right; m->n_groups = 0; m->collapsed = false;
      const int rr = hawk_reserve_cols(m->own, std::max<uint64_t>(totals[0], 1), &mc);
      if (rr) prep = rr;
      m->cols = mc; m->cap = mc.cap;
    }
  }
  // go / no-go: a destination that could not reserve the merged columns must not leave its peers blocked in ncclSend
  {
    uint64_t word = (uint64_t)(prep != HAWK_OK);
    std::vector<uint64_t> words(W);
    rc = hawk_comm_allgather_u64(c, &word, 1, words.data());
    bool stop = rc != HAWK_OK;
    for (int r = 0; r < W && !stop; ++r) stop = words[r] != 0;
    if (stop) {
      if (m) hawk_table_destroy(m);
      if (rc) return rc;
      if (prep) return prep;
      snprintf(g_comm_err, sizeof(g_comm_err), "hawk_table_gather: the destination rank could not reserve the merged table");
      return HAWK_E_COMM;
    }
  }
  const void* sp[HAWK_GATHER_COLS] = {sc.hap, sc.pos, sc.strand, sc.start, sc.stop, sc.flags, sc.cfdon, sc.win, sc.win + sc.cap, sc.win + 2 * sc.cap,
                                      sc.win + 3 * sc.cap, sc.win + 4 * sc.cap};
  void* rp[HAWK_GATHER_COLS] = {mc.hap, mc.pos, mc.strand, mc.start, mc.stop, mc.flags, mc.cfdon, mc.win, mc.win + mc.cap, mc.win + 2 * mc.cap,
                                mc.win + 3 * mc.cap, mc.win + 4 * mc.cap};
  // from here on every failure releases the merged table
  auto fail = [&](int code) { if (m) hawk_table_destroy(m); return code; };
#define HIPCHK_M(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      snprintf(hawk_hip_err_buf(), 256, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return fail(HAWK_E_HIP);                                                                 \
    }                                                                                          \
  } while (0)
  HIPCHK_M(hipEventRecord(ctx->ev[6], st));
  int r0 = g_rccl.GroupStart();
  if (r0 == 0) {
    for (uint32_t i = 0; i < n_ops && r0 == 0; ++i) {
      const hawk_gather_op& op = ops[i];
      if (!root) r0 = g_rccl.Send(sp[op.col], op.bytes, kNcclUint8, dst, c->nccl, st);
      else if ((int)op.peer != dst) r0 = g_rccl.Recv((char*)rp[op.col] + op.offset, op.bytes, kNcclUint8, (int)op.peer, c->nccl, st);
    }
    const int r1 = g_rccl.GroupEnd();
    if (r0 == 0) r0 = r1;
  }
  if (r0 != 0) {
    snprintf(g_comm_err, sizeof(g_comm_err), "hawk_table_gather: %s", g_rccl.GetErrorString(r0));
    return fail(HAWK_E_COMM);
  }
  if (root) {
    for (uint32_t i = 0; i < n_ops; ++i) {  // the destination's own slice: a local copy
      const hawk_gather_op& op = ops[i];
      if ((int)op.peer == dst && n) HIPCHK_M(hipMemcpyAsync((char*)rp[op.col] + op.offset, sp[op.col], op.bytes, hipMemcpyDeviceToDevice, st));
    }
    for (int r = 0; r < W; ++r) {
      const uint64_t nr = all[4 * r];
      const uint32_t ho = (uint32_t)all[4 * r + 1];
      if (nr && ho) hipLaunchKernelGGL(k_hap_shift, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, mc.hap + off[r], nr, ho);
    }
    HIPCHK_M(hipGetLastError());
  }
  HIPCHK_M(hipEventRecord(ctx->ev[7], st));
  HIPCHK_M(hipStreamSynchronize(st));
#undef HIPCHK_M
  if (ms) (void)hipEventElapsedTime(ms, ctx->ev[6], ctx->ev[7]);
  if (root) *merged = m;
  return HAWK_OK;
}

}  // extern "C"
