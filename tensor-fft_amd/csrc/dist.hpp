// dist.hpp — one transform spread over the GPUs of a node (include/tfft.h, tfft_dist_*; BASELINE configs[4b]).
// Included at the end of tfft.hip: it builds on create_plan / launch_chain of that file.
//
// Four-step FFT, N = N1 N2, x viewed as [N1][N2] (n = n1 N2 + n2), one process (or one device of a process) per rank:
//
//   in  ("columns"):    rank p owns x[n1 N2 + p C + c], c < C = N2 / P, stored [N1][C]
//   pre                 ONE radix-N1 column pass along the strided axis whose epilogue applies the four-step twiddle
//                       w_N^(k1 (p C + c)) (tfft_plan_opts.fourstep_n): Y[k1][c] into the send buffer; rows k1 of rank q's
//                       block [q K, (q + 1) K), K = N1 / P, are one contiguous chunk of K C samples per plane
//   exchange            chunk q of both planes goes to rank q: ONE ncclGroupStart / ncclSend + ncclRecv per peer and plane /
//                       ncclGroupEnd on the caller's stream (RCCL over xGMI: every link busy at once, 7 x 4 MiB per plane
//                       and GPU at N = 2^26, P = 8); the own chunk is a device-to-device copy
//   post                the N2-point row transforms read the receive buffer [p'][k][c] IN PLACE: row k is P segments of C
//                       contiguous samples (segmented input rows of the first column pass, colfft::Args::in_seg_*), so no
//                       re-order pass is needed; lengths whose row transform is a single LDS-resident kernel (N2 <= 32768)
//                       or starts with a radix-1024 pass keep a re-order pass [p'][k][c] -> [k][p' C + c] in front
//   out ("transposed"): rank q owns X[k1 + N1 k2], k1 in its block, stored [K][N2]
//
// The reference has nothing to compare with: its multi-GPU code is commented out and ran independent transforms per device
// (src/base/ComputeFFT.h:295-411, src/base/DataHandler.h:168-403).
//
// RCCL is bound at first use with dlopen (librccl.so.1): a process that never creates a communicator does not load it, and
// inside a PyTorch process the library PyTorch has already loaded is the one that gets used.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      r.error = std::string("dlopen(librccl.so.1) failed: ") + dlerror();
      return;
    }
    bool ok = true;
    auto bind = [&](auto& fn, const char* sym) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.handle, sym));
      if (!fn) {
        ok = false;
        r.error = std::string("librccl: missing symbol ") + sym;
      }
    };
    bind(r.GetUniqueId, "ncclGetUniqueId");
    bind(r.CommInitRank, "ncclCommInitRank");
    bind(r.CommInitAll, "ncclCommInitAll");
    bind(r.CommDestroy, "ncclCommDestroy");
    bind(r.CommCount, "ncclCommCount");
    bind(r.CommUserRank, "ncclCommUserRank");
    bind(r.Send, "ncclSend");
    bind(r.Recv, "ncclRecv");
    bind(r.GroupStart, "ncclGroupStart");
    bind(r.GroupEnd, "ncclGroupEnd");
    bind(r.GetErrorString, "ncclGetErrorString");
    bind(r.GetVersion, "ncclGetVersion");
    if (!ok) {
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return &r;
}

int rccl_ready() {
  Rccl* r = rccl();
  if (!r->handle) return fail(TFFT_ERR_COMM, r->error);
  return TFFT_OK;
}

int nccl_fail(ncclResult_t e, const char* what) {
  return fail(TFFT_ERR_COMM, std::string(what) + ": " + rccl()->GetErrorString(e));
}
#define TFFT_NCCL(call)                                       \
  do {                                                        \
    const ncclResult_t e_ = (call);                           \
    if (e_ != ncclSuccess) return nccl_fail(e_, #call);       \
  } while (0)

// N1 of the split: the column pass is ONE radix-256 or radix-512 kernel with the four-step twiddle in its epilogue, and every
// rank needs at least 64 columns (a cooperative workgroup's width). Prefer the N1 whose N2 the library transforms in the
// fewest passes: a single-kernel length (<= 2^15, 4096 first of all); otherwise N1 = 256 (256-byte row segments).
int dist_geometry(uint64_t n, int world, int rank, tfft_dist_geometry* g) {
  if (!g) return fail(TFFT_ERR_ARG, "null geometry pointer");
  if (!is_pow2(n)) return fail(TFFT_ERR_NOT_POW2, "Error! Input size has to be a power of 2!");
  if (world < 1 || !is_pow2(static_cast<uint64_t>(world))) return fail(TFFT_ERR_ARG, "the number of ranks has to be a power of 2");
  if (rank < 0 || rank >= world) return fail(TFFT_ERR_ARG, "rank outside [0, world)");
  const int lg = ilog2(n);
  const uint64_t p = static_cast<uint64_t>(world);
  const bool prefer512 = (lg - 9 == 12) || (lg - 8 > 15 && lg - 9 <= 15);
  const uint64_t order[2] = {prefer512 ? 512u : 256u, prefer512 ? 256u : 512u};
  uint64_t n1 = 0;
  for (uint64_t cand : order) {
    if (cand >= n) continue;
    const uint64_t n2 = n / cand;
    if (cand % p || n2 % p) continue;
    const uint64_t c = n2 / p;
    if (c < 64 || c % 64) continue;
    n1 = cand;
    break;
  }
  if (!n1)
    return fail(TFFT_ERR_ARG, "N = " + std::to_string(n) + " is too small for " + std::to_string(world) +
                                  " ranks: the distributed transform needs N >= 256 * 64 * ranks (64 columns per rank)");
  std::memset(g, 0, sizeof(*g));
  g->struct_size = static_cast<uint32_t>(sizeof(*g));
  g->n = n;
  g->n1 = n1;
  g->n2 = n / n1;
  g->cols = g->n2 / p;
  g->rows = n1 / p;
  g->chunk = g->rows * g->cols;
  g->world = world;
  g->rank = rank;
  g->fused = 1;
  // the row transform reads segments in place when it starts with a cooperative radix-256 / radix-512 column pass
  std::vector<Pass> passes;
  plan_passes(g->n2, 1, 0, passes);
  const Pass& f = passes[0];
  const uint64_t pitch = f.kind == PassKind::Col256 ? g->n2 / static_cast<uint64_t>(f.radix) : 0;
  const bool seg = world > 1 && passes.size() >= 2 && f.kind == PassKind::Col256 && (f.radix == 256 || f.radix == 512) &&
                   pitch % 128 == 0 && g->cols >= pitch;
  g->reorder = (world > 1 && !seg) ? 1 : 0;
  g->local_passes = 1 + g->reorder + static_cast<int>(passes.size());
  g->slabs = 1;
  return TFFT_OK;
}

}  // namespace

struct tfft_dist_plan {
  tfft_dist_geometry g{};
  int device = 0;
  ncclComm_t comm = nullptr;
  tfft_plan* col = nullptr;
  tfft_plan* row = nullptr;
  // one device block: send RE | send IM | recv RE | recv IM | (re-order RE | IM) | row-plan scratch; the four exchange
  // buffers may be replaced by caller-owned memory (tfft_dist_plan_set_buffers)
  void* block = nullptr;
  _Float16 *send_re = nullptr, *send_im = nullptr, *recv_re = nullptr, *recv_im = nullptr;
  _Float16 *tmp_re = nullptr, *tmp_im = nullptr;
  bool self_via_comm = false;   // TFFT_DIST_SELF_VIA_COMM: the own chunk goes through ncclSend / ncclRecv too
  bool caller_buffers = false;  // TFFT_DIST_CALLER_BUFFERS: no internal exchange block; tfft_dist_plan_set_buffers before the first exec
  // Round 5: the C columns of the column pass in S slabs (TFFT_DIST_SLABS_2 / _4). Slab s is its own launch of the column kernel
  // and its own ncclSend / ncclRecv group; tfft_dist_exec puts the groups on a second stream, each behind the event of its slab's
  // column pass, so slab s travels over xGMI while slab s + 1 is being computed. Buffers: send [q][s][k][c_s], receive
  // [p'][s][k][c_s] (what rank p' sent for this rank); the row transforms read P S segments of C / S samples per row.
  int slabs = 1;
  hipStream_t comm_stream = nullptr;
  std::vector<hipEvent_t> slab_done;
  hipEvent_t exchange_done = nullptr;
};

extern "C" {

namespace {
// tfft_dist_geometry is an OUT struct: the caller says how many bytes it owns (struct_size), the library fills at most that many.
constexpr size_t kGeometryMinBytes = offsetof(tfft_dist_geometry, local_passes) + sizeof(int);     // the round-4 layout
int geometry_out(const tfft_dist_geometry& g, tfft_dist_geometry* out) {
  if (!out) return fail(TFFT_ERR_ARG, "null geometry pointer");
  uint32_t sz = 0;
  std::memcpy(&sz, out, sizeof(sz));
  if (sz < kGeometryMinBytes || sz > 4096 || (sz % 8))
    return fail(TFFT_ERR_ARG, "tfft_dist_geometry.struct_size = " + std::to_string(sz) + ": set it to sizeof(tfft_dist_geometry) before the call "
                                  "(TFFT_DIST_GEOMETRY_INIT); this library writes " + std::to_string(sizeof(g)) + " bytes at most");
  const size_t copy = std::min<size_t>(sz, sizeof(g));
  std::memcpy(out, &g, copy);
  std::memcpy(out, &sz, sizeof(sz));            // the caller's size stays what it was
  return TFFT_OK;
}
}  // namespace

int tfft_dist_geometry_query(uint64_t n, int world, int rank, tfft_dist_geometry* out) {
  g_err.clear();
  tfft_dist_geometry g;
  const int rc = dist_geometry(n, world, rank, &g);
  if (rc) return rc;
  return geometry_out(g, out);
}

int tfft_dist_rccl_version(int* version) {
  g_err.clear();
  if (!version) return fail(TFFT_ERR_ARG, "null argument");
  int rc = rccl_ready();
  if (rc) return rc;
  TFFT_NCCL(rccl()->GetVersion(version));
  return TFFT_OK;
}

int tfft_dist_unique_id(void* id128) {
  g_err.clear();
  if (!id128) return fail(TFFT_ERR_ARG, "null id buffer");
  int rc = rccl_ready();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == TFFT_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  TFFT_NCCL(rccl()->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return TFFT_OK;
}

int tfft_dist_comm_create(int world, int rank, const void* id128, int device_id, void** comm) {
  g_err.clear();
  if (!id128 || !comm) return fail(TFFT_ERR_ARG, "null argument");
  *comm = nullptr;
  int rc = rccl_ready();
  if (rc) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  int prev = 0;
  TFFT_HIP(hipGetDevice(&prev));
  TFFT_HIP(hipSetDevice(device_id));
  ncclComm_t c = nullptr;
  const ncclResult_t e = rccl()->CommInitRank(&c, world, id, rank);
  (void)hipSetDevice(prev);
  if (e != ncclSuccess) return nccl_fail(e, "ncclCommInitRank");
  *comm = c;
  return TFFT_OK;
}

int tfft_dist_comm_create_all(int ndev, const int* devices, void** comms) {
  g_err.clear();
  if (ndev < 1 || !comms) return fail(TFFT_ERR_ARG, "bad argument");
  int rc = rccl_ready();
  if (rc) return rc;
  std::vector<ncclComm_t> c(static_cast<size_t>(ndev), nullptr);
  TFFT_NCCL(rccl()->CommInitAll(c.data(), ndev, devices));
  for (int i = 0; i < ndev; ++i) comms[i] = c[static_cast<size_t>(i)];
  return TFFT_OK;
}

int tfft_dist_comm_destroy(void* comm) {
  g_err.clear();
  if (!comm) return TFFT_OK;
  int rc = rccl_ready();
  if (rc) return rc;
  TFFT_NCCL(rccl()->CommDestroy(static_cast<ncclComm_t>(comm)));
  return TFFT_OK;
}

int tfft_dist_comm_info(void* comm, int* count, int* rank) {
  g_err.clear();
  if (!comm) return fail(TFFT_ERR_ARG, "null communicator");
  int rc = rccl_ready();
  if (rc) return rc;
  if (count) TFFT_NCCL(rccl()->CommCount(static_cast<ncclComm_t>(comm), count));
  if (rank) TFFT_NCCL(rccl()->CommUserRank(static_cast<ncclComm_t>(comm), rank));
  return TFFT_OK;
}

int tfft_dist_group_start(void) {
  g_err.clear();
  int rc = rccl_ready();
  if (rc) return rc;
  TFFT_NCCL(rccl()->GroupStart());
  return TFFT_OK;
}

int tfft_dist_group_end(void) {
  g_err.clear();
  int rc = rccl_ready();
  if (rc) return rc;
  TFFT_NCCL(rccl()->GroupEnd());
  return TFFT_OK;
}

void tfft_dist_plan_destroy(tfft_dist_plan* p) {
  if (!p) return;
  tfft_plan_destroy(p->col);
  tfft_plan_destroy(p->row);
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(p->device);
  if (p->block) (void)hipFree(p->block);
  for (hipEvent_t e : p->slab_done) (void)hipEventDestroy(e);
  if (p->exchange_done) (void)hipEventDestroy(p->exchange_done);
  if (p->comm_stream) (void)hipStreamDestroy(p->comm_stream);
  (void)hipSetDevice(prev);
  delete p;
}

int tfft_dist_plan_create(uint64_t n, int world, int rank, int device_id, void* comm, int flags, tfft_dist_plan** out) {
  g_err.clear();
  if (!out) return fail(TFFT_ERR_ARG, "null plan pointer");
  *out = nullptr;
  if (flags & ~(TFFT_DIST_SELF_VIA_COMM | TFFT_DIST_CALLER_BUFFERS | TFFT_DIST_SLABS_2 | TFFT_DIST_SLABS_4)) return fail(TFFT_ERR_ARG, "unknown flag");
  if ((flags & TFFT_DIST_SLABS_2) && (flags & TFFT_DIST_SLABS_4)) return fail(TFFT_ERR_ARG, "TFFT_DIST_SLABS_2 and TFFT_DIST_SLABS_4 exclude each other");
  if ((flags & TFFT_DIST_SELF_VIA_COMM) && !comm) return fail(TFFT_ERR_ARG, "TFFT_DIST_SELF_VIA_COMM needs a communicator");
  tfft_dist_geometry g;
  int rc = dist_geometry(n, world, rank, &g);
  if (rc) return rc;
  if (comm) {
    rc = rccl_ready();
    if (rc) return rc;
    int cnt = 0, me = -1;
    TFFT_NCCL(rccl()->CommCount(static_cast<ncclComm_t>(comm), &cnt));
    TFFT_NCCL(rccl()->CommUserRank(static_cast<ncclComm_t>(comm), &me));
    if (cnt != world || me != rank)
      return fail(TFFT_ERR_ARG, "communicator has " + std::to_string(cnt) + " ranks and this is its rank " + std::to_string(me) +
                                    ", the plan was asked for rank " + std::to_string(rank) + " of " + std::to_string(world));
  }
  tfft_dist_plan* p = new tfft_dist_plan;
  g.slabs = (flags & TFFT_DIST_SLABS_4) ? 4 : ((flags & TFFT_DIST_SLABS_2) ? 2 : 1);
  p->g = g;
  p->device = device_id;
  p->comm = static_cast<ncclComm_t>(comm);
  p->self_via_comm = (flags & TFFT_DIST_SELF_VIA_COMM) != 0;
  p->caller_buffers = (flags & TFFT_DIST_CALLER_BUFFERS) != 0;
  const bool two_sided = world > 1 || p->self_via_comm;      // separate receive buffers
  p->slabs = (flags & TFFT_DIST_SLABS_4) ? 4 : ((flags & TFFT_DIST_SLABS_2) ? 2 : 1);
  if (p->slabs > 1) {
    // a slab is whole 128-column blocks of ONE four-step radix-256 column pass, and the row transforms must be able to read the
    // received pieces in place (segments of C / S samples no shorter than a row of their first column pass)
    if (g.n1 != 256 || g.reorder || (g.cols / static_cast<uint64_t>(p->slabs)) % 128) {
      const std::string why = g.n1 != 256 ? "its column pass is radix 512" : (g.reorder ? "its row transforms need the re-order pass" : "a slab would be narrower than 128 columns");
      delete p;
      return fail(TFFT_ERR_ARG, "TFFT_DIST_SLABS_*: this geometry cannot overlap its exchange (" + why + "); create the plan without the flag");
    }
  }
  auto bail = [&](int code) {
    const std::string keep = g_err;
    tfft_dist_plan_destroy(p);
    g_err = keep;
    return code;
  };
  // column pass: n = N1 along the strided axis, C columns, four-step twiddle with this rank's column offset
  tfft_plan_opts co = TFFT_PLAN_OPTS_INIT;
  co.inner = g.cols;
  co.in_batch_stride = g.n1 * g.cols;
  co.out_batch_stride = g.n1 * g.cols;
  co.preserve_input = 1;
  co.fourstep_n = n;
  co.fourstep_col0 = static_cast<uint64_t>(rank) * g.cols;
  rc = create_plan(g.n1, 1, device_id, &co, InternalOpts{}, &p->col);
  if (rc) return bail(rc);
  // row transforms: K contiguous rows of N2, read from the receive buffer in place (segments) or behind a re-order pass
  tfft_plan_opts ro = TFFT_PLAN_OPTS_INIT;
  ro.in_batch_stride = g.reorder || world == 1 ? g.n2 : g.cols;
  ro.out_batch_stride = g.n2;
  ro.preserve_input = 1;
  InternalOpts ri;
  if (!g.reorder && (world > 1 || p->slabs > 1)) {
    ri.in_seg_len = g.cols / static_cast<uint64_t>(p->slabs);
    ri.in_seg_stride = g.chunk / static_cast<uint64_t>(p->slabs);
    ro.in_batch_stride = ri.in_seg_len;
  }
  rc = create_plan(g.n2, g.rows, device_id, &ro, ri, &p->row);
  if (rc) return bail(rc);
  if (p->slabs > 1) {
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < p->slabs; ++i) {
      hipEvent_t ev = nullptr;
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess) p->slab_done.push_back(ev);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->exchange_done, hipEventDisableTiming);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return bail(hip_fail(e, "stream / events of the overlapped exchange"));
  }
  const size_t plane = static_cast<size_t>(g.n / static_cast<uint64_t>(world)) * sizeof(_Float16);   // N / P halves
  const size_t row_ws = tfft_plan_workspace_bytes(p->row);
  // (TFFT_DIST_CALLER_BUFFERS: the exchange buffers come from the caller, e.g. tensors of a framework whose own collective
  // sends them; the plan then only owns the re-order and row-plan scratch)
  const size_t exch = p->caller_buffers ? 0 : plane * (two_sided ? 4 : 2);
  const size_t total = exch + (g.reorder ? 2 * plane : 0) + row_ws;
  uint8_t* b = nullptr;
  if (total) {
    int prev = 0;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipMalloc(&p->block, total);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return bail(hip_fail(e, "hipMalloc(distributed plan buffers)"));
    b = static_cast<uint8_t*>(p->block);
  }
  if (!p->caller_buffers) {
    p->send_re = reinterpret_cast<_Float16*>(b);
    p->send_im = reinterpret_cast<_Float16*>(b + plane);
    b += 2 * plane;
    if (two_sided) {
      p->recv_re = reinterpret_cast<_Float16*>(b);
      p->recv_im = reinterpret_cast<_Float16*>(b + plane);
      b += 2 * plane;
    } else {
      p->recv_re = p->send_re;       // one rank: nothing to exchange, the row pass reads what the column pass wrote
      p->recv_im = p->send_im;
    }
  }
  if (g.reorder) {
    p->tmp_re = reinterpret_cast<_Float16*>(b);
    p->tmp_im = reinterpret_cast<_Float16*>(b + plane);
    b += 2 * plane;
  }
  if (row_ws) {
    rc = tfft_plan_set_workspace(p->row, b, row_ws);
    if (rc) return bail(rc);
  }
  *out = p;
  return TFFT_OK;
}

int tfft_dist_plan_geometry(const tfft_dist_plan* p, tfft_dist_geometry* out) {
  g_err.clear();
  if (!p || !out) return fail(TFFT_ERR_ARG, "null argument");
  return geometry_out(p->g, out);
}

int tfft_dist_plan_buffers(const tfft_dist_plan* p, void** send_re, void** send_im, void** recv_re, void** recv_im) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  if (send_re) *send_re = p->send_re;
  if (send_im) *send_im = p->send_im;
  if (recv_re) *recv_re = p->recv_re;
  if (recv_im) *recv_im = p->recv_im;
  return TFFT_OK;
}

int tfft_dist_plan_set_buffers(tfft_dist_plan* p, void* send_re, void* send_im, void* recv_re, void* recv_im) {
  g_err.clear();
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  if (!send_re || !send_im || !recv_re || !recv_im) return fail(TFFT_ERR_ARG, "null buffer");
  if ((reinterpret_cast<uintptr_t>(send_re) | reinterpret_cast<uintptr_t>(send_im) | reinterpret_cast<uintptr_t>(recv_re) |
       reinterpret_cast<uintptr_t>(recv_im)) & 15)
    return fail(TFFT_ERR_ARG, "buffers must be 16-byte aligned");
  // validate first, commit on success: each buffer is the range [ptr, ptr + N / world halves); the two send planes must not
  // overlap each other, and with a real exchange no receive range may overlap a send range or the other receive range (ncclRecv
  // would write into bytes that are still being sent)
  const bool two_sided = p->g.world > 1 || p->self_via_comm;
  const size_t bytes = static_cast<size_t>(p->g.n / static_cast<uint64_t>(p->g.world)) * sizeof(_Float16);
  auto overlap = [bytes](const void* a, const void* b2) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b2);
    return x < y + bytes && y < x + bytes;
  };
  if (overlap(send_re, send_im)) return fail(TFFT_ERR_ARG, "the two send planes overlap");
  if (two_sided && (overlap(recv_re, recv_im) || overlap(recv_re, send_re) || overlap(recv_re, send_im) || overlap(recv_im, send_re) ||
                    overlap(recv_im, send_im)))
    return fail(TFFT_ERR_ARG, "send and receive buffers must be distinct, non-overlapping ranges of N / world halves each");
  p->send_re = static_cast<_Float16*>(send_re);
  p->send_im = static_cast<_Float16*>(send_im);
  // one rank and no collective: nothing moves between the phases, the row transforms read what the column pass wrote
  p->recv_re = two_sided ? static_cast<_Float16*>(recv_re) : p->send_re;
  p->recv_im = two_sided ? static_cast<_Float16*>(recv_im) : p->send_im;
  return TFFT_OK;
}

namespace {
int dist_check(const tfft_dist_plan* p) {
  if (!p) return fail(TFFT_ERR_ARG, "null plan");
  int cur = 0;
  TFFT_HIP(hipGetDevice(&cur));
  if (cur != p->device) return fail(TFFT_ERR_ARG, "plan was created for another device than the current one");
  if (!p->send_re || !p->recv_re)
    return fail(TFFT_ERR_ARG, "this plan was created with TFFT_DIST_CALLER_BUFFERS: call tfft_dist_plan_set_buffers before the first execution");
  return TFFT_OK;
}
}  // namespace

namespace {
// column pass of slab sl (all of it for S = 1) into the send buffers
int dist_launch_col(const tfft_dist_plan* p, int sl, const void* in_re, const void* in_im, hipStream_t s) {
  if (p->slabs == 1) return launch_chain(p->col, in_re, in_im, p->send_re, p->send_im, s);
  const uint64_t cs = p->g.cols / static_cast<uint64_t>(p->slabs);            // columns per slab
  g_slab.on = true;
  g_slab.col_first = static_cast<uint64_t>(sl) * cs;
  g_slab.col_count = cs;
  g_slab.out_pitch_shift = static_cast<uint32_t>(ilog2(cs));                 // row k of the piece: k (C / S)
  g_slab.out_seg_shift = static_cast<uint32_t>(ilog2(p->g.rows));            // k / K = destination rank q
  g_slab.out_seg_gap = p->g.chunk - p->g.chunk / static_cast<uint64_t>(p->slabs);   // q K C - q K C / S
  g_slab.out_base = static_cast<uint64_t>(sl) * (p->g.chunk / static_cast<uint64_t>(p->slabs));
  const int rc = launch_chain(p->col, in_re, in_im, p->send_re, p->send_im, s);
  g_slab = SlabCtx{};
  return rc;
}

// slab sl of chunk q (both planes) goes to rank q: one ncclGroupStart / Send + Recv per peer and plane / GroupEnd on s
int dist_exchange_slab(const tfft_dist_plan* p, int sl, hipStream_t s) {
  const int world = p->g.world, me = p->g.rank;
  if (world == 1 && !p->self_via_comm) return TFFT_OK;
  if (!p->comm) return fail(TFFT_ERR_COMM, "this plan was created without a communicator: run the exchange yourself between "
                                           "tfft_dist_exec_pre and tfft_dist_exec_post (chunk q of the send buffers goes to rank q)");
  const size_t chunk = static_cast<size_t>(p->g.chunk), piece = chunk / static_cast<size_t>(p->slabs), off = static_cast<size_t>(sl) * piece;
  // own chunk: device-to-device copy, ordered on the same stream (or, with TFFT_DIST_SELF_VIA_COMM, a send to itself)
  if (!p->self_via_comm) {
    TFFT_HIP(hipMemcpyAsync(p->recv_re + me * chunk + off, p->send_re + me * chunk + off, piece * sizeof(_Float16), hipMemcpyDeviceToDevice, s));
    TFFT_HIP(hipMemcpyAsync(p->recv_im + me * chunk + off, p->send_im + me * chunk + off, piece * sizeof(_Float16), hipMemcpyDeviceToDevice, s));
  }
  Rccl* r = rccl();
  TFFT_NCCL(r->GroupStart());
  for (int d = p->self_via_comm ? 0 : 1; d < world; ++d) {
    // peers in a rotated order: at step d rank r sends to r + d and receives from r - d, so the P ranks' first
    // transfers do not all target the same GPU
    const int to = (me + d) % world, from = (me - d + world) % world;
    ncclResult_t e = r->Send(p->send_re + to * chunk + off, piece, ncclHalf, to, p->comm, s);
    if (e == ncclSuccess) e = r->Send(p->send_im + to * chunk + off, piece, ncclHalf, to, p->comm, s);
    if (e == ncclSuccess) e = r->Recv(p->recv_re + from * chunk + off, piece, ncclHalf, from, p->comm, s);
    if (e == ncclSuccess) e = r->Recv(p->recv_im + from * chunk + off, piece, ncclHalf, from, p->comm, s);
    if (e != ncclSuccess) {
      (void)r->GroupEnd();
      return nccl_fail(e, "ncclSend / ncclRecv");
    }
  }
  TFFT_NCCL(r->GroupEnd());
  return TFFT_OK;
}
}  // namespace

int tfft_dist_exec_pre(const tfft_dist_plan* p, const void* in_re, const void* in_im, void* stream) {
  g_err.clear();
  int rc = dist_check(p);
  if (rc) return rc;
  if (!in_re || !in_im) return fail(TFFT_ERR_ARG, "null data pointer");
  if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im)) & 15) return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  for (int sl = 0; sl < p->slabs; ++sl) {
    rc = dist_launch_col(p, sl, in_re, in_im, static_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  return TFFT_OK;
}

int tfft_dist_exec_exchange(const tfft_dist_plan* p, void* stream) {
  g_err.clear();
  int rc = dist_check(p);
  if (rc) return rc;
  for (int sl = 0; sl < p->slabs; ++sl) {
    rc = dist_exchange_slab(p, sl, static_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  return TFFT_OK;
}

int tfft_dist_exec_post(const tfft_dist_plan* p, void* out_re, void* out_im, void* stream) {
  g_err.clear();
  int rc = dist_check(p);
  if (rc) return rc;
  if (!out_re || !out_im) return fail(TFFT_ERR_ARG, "null data pointer");
  if ((reinterpret_cast<uintptr_t>(out_re) | reinterpret_cast<uintptr_t>(out_im)) & 15) return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
  const _Float16 *src_re = p->recv_re, *src_im = p->recv_im;
  if (p->g.reorder) {
    rc = tfft_permute_twiddle(p->recv_re, p->recv_im, p->tmp_re, p->tmp_im, static_cast<uint64_t>(p->g.world), p->g.rows, p->g.cols, 0, 0, stream);
    if (rc) return rc;
    src_re = p->tmp_re;
    src_im = p->tmp_im;
  }
  return launch_chain(p->row, src_re, src_im, out_re, out_im, static_cast<hipStream_t>(stream));
}

int tfft_dist_exec(const tfft_dist_plan* p, const void* in_re, const void* in_im, void* out_re, void* out_im, void* stream) {
  int rc;
  if (p && p->slabs > 1 && (p->g.world > 1 || p->self_via_comm) && p->comm) {
    // overlapped: slab s's exchange on the plan's second stream behind the event of its column pass, while slab s + 1 computes on
    // the caller's stream; the row transforms wait for the last exchange. (The three separate calls run the same slabs one after
    // the other on one stream: same buffers, same results, separable phases.)
    g_err.clear();
    rc = dist_check(p);
    if (rc) return rc;
    if (!in_re || !in_im) return fail(TFFT_ERR_ARG, "null data pointer");
    if ((reinterpret_cast<uintptr_t>(in_re) | reinterpret_cast<uintptr_t>(in_im)) & 15) return fail(TFFT_ERR_ARG, "data pointers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int sl = 0; sl < p->slabs; ++sl) {
      rc = dist_launch_col(p, sl, in_re, in_im, s);
      if (rc) return rc;
      TFFT_HIP(hipEventRecord(p->slab_done[static_cast<size_t>(sl)], s));
      TFFT_HIP(hipStreamWaitEvent(p->comm_stream, p->slab_done[static_cast<size_t>(sl)], 0));
      rc = dist_exchange_slab(p, sl, p->comm_stream);
      if (rc) return rc;
    }
    TFFT_HIP(hipEventRecord(p->exchange_done, p->comm_stream));
    TFFT_HIP(hipStreamWaitEvent(s, p->exchange_done, 0));
    return tfft_dist_exec_post(p, out_re, out_im, stream);
  }
  rc = tfft_dist_exec_pre(p, in_re, in_im, stream);
  if (rc) return rc;
  rc = tfft_dist_exec_exchange(p, stream);
  if (rc) return rc;
  return tfft_dist_exec_post(p, out_re, out_im, stream);
}

}  // extern "C"
