// Vertex-partitioned Poisson sweep behind the C-ABI: a libglx-owned RCCL communicator and a
// rank-local sweep object whose whole iteration -- [boundary rows | pack | exchange | interior rows] x sweeps, stop
// test included -- is enqueued (and captured into device graphs) by the library.  The reference has no distributed
// code; the arithmetic is the sweep of graphlearning/ssl.py:667-670 row for row (sweep.hip), so results are
// bit-identical to the single-GPU path for any partition.
//
// Per sweep, on the sweep's stream S and an exchange stream X:
//   S: SpMM of the BOUNDARY rows (rows some peer gathers)      -> xout[0:nb]
//   S: pack the records the peers need into one send buffer      (send lists precomputed by the planner)
//   X: (after the pack) grouped ncclSend / ncclRecv, one pair per peer: a direct all-to-all-v -- xGMI is
//      point-to-point, every link carries its own pair, no ring -- landing in xout's halo region
//   S: SpMM of the INTERIOR rows meanwhile                     -> xout[nb:n_own]
//   S: waits for X before the next sweep.
// Stop test (ssl.py:667) without a host round trip per sweep: the kernels record the rank-local maxima of
// |deg w - vinf| per sweep; past min_iter the sweeps run in chunks of `check_every` on a ring of check_every+1 state
// buffers, ONE ncclAllReduce(MAX) per chunk makes the maxima global, the host reads them once per chunk and picks
// the first T that satisfies the test -- u_T is still in the ring, so the result is exactly the reference's.
// RCCL is bound at run time (dlopen): libglx loads and serves single-GPU callers without it.
#include "glx_internal.h"
#include <dlfcn.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <rccl/rccl.h>

// ---- RCCL binding -------------------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    // the copy a host framework (torch) has already mapped wins: same soname, one RCCL instance per process
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (api.handle) break;
    }
    for (int i = 0; i < 3 && !api.handle; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) return;
#define BIND(f) *(void**)(&api.f) = dlsym(api.handle, "nccl" #f); if (!api.f) return;
    BIND(GetUniqueId) BIND(CommInitRank) BIND(CommInitAll) BIND(CommDestroy) BIND(GroupStart) BIND(GroupEnd)
    BIND(Send) BIND(Recv) BIND(AllReduce) BIND(AllGather) BIND(GetErrorString)
#undef BIND
    ok = true;
  });
  return ok ? &api : nullptr;
}

#define GLX_NCCL(call)                                                                                     \
  do {                                                                                                     \
    ncclResult_t r_ = (call);                                                                              \
    if (r_ != ncclSuccess) {                                                                               \
      glx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, rccl_api()->GetErrorString(r_));          \
      return GLX_ERCCL;                                                                                    \
    }                                                                                                      \
  } while (0)

struct glx_dist_sweep;
struct glx_comm {
  ncclComm_t comm = nullptr;   // null: single rank, no transport needed
  int rank = 0, nranks = 1, device = 0;
  // sweeps created on this communicator and still alive.  Their captured device graphs hold RCCL kernels: ncclCommDestroy
  // waits for every such graph to be released (seen as a hang when a communicator was closed before its sweep, round 3), so
  // glx_dist_destroy releases them first and detaches the sweeps.
  std::vector<glx_dist_sweep*> sweeps;
  std::mutex mu;
};
static void detach_sweep(glx_dist_sweep* s);

extern "C" int glx_dist_unique_id(char id_out[128]) {
  GLX_CHECK(id_out, GLX_EINVAL, "glx_dist_unique_id: null output");
  RcclApi* a = rccl_api();
  GLX_CHECK(a, GLX_ERCCL, "glx_dist_unique_id: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
  ncclUniqueId id;
  GLX_NCCL(a->GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id_out, &id, 128);
  return GLX_OK;
}

extern "C" int glx_dist_init_rank(int nranks, int rank, const char id[128], int device, glx_comm** out) {
  GLX_CHECK(out, GLX_EINVAL, "glx_dist_init_rank: null output");
  *out = nullptr;
  GLX_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, GLX_EINVAL, "glx_dist_init_rank: rank %d of %d", rank, nranks);
  GLX_HIP(hipSetDevice(device));
  glx_comm* c = new glx_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device;
  if (id) {
    RcclApi* a = rccl_api();
    if (!a) { delete c; glx_set_error("glx_dist_init_rank: librccl.so.1 could not be loaded"); return GLX_ERCCL; }
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclResult_t r = a->CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) { glx_set_error("ncclCommInitRank -> %s", a->GetErrorString(r)); delete c; return GLX_ERCCL; }
  }   // id == NULL: a rank identity without a transport (one rank, or the stepwise form with the caller's transport)
  *out = c;
  return GLX_OK;
}

// one process driving all the GPUs of the node (SURVEY.md 8b signature): out[nranks]
extern "C" int glx_dist_init(int nranks, const int* devices, glx_comm** out) {
  GLX_CHECK(out && devices && nranks >= 1, GLX_EINVAL, "glx_dist_init: bad argument");
  RcclApi* a = rccl_api();
  GLX_CHECK(a, GLX_ERCCL, "glx_dist_init: librccl.so.1 could not be loaded");
  std::vector<ncclComm_t> comms(nranks);
  GLX_NCCL(a->CommInitAll(comms.data(), nranks, devices));
  for (int r = 0; r < nranks; ++r) {
    glx_comm* c = new glx_comm();
    c->comm = comms[r];
    c->rank = r;
    c->nranks = nranks;
    c->device = devices[r];
    out[r] = c;
  }
  return GLX_OK;
}

extern "C" int glx_dist_comm_info(const glx_comm* c, int32_t info[4]) {
  GLX_CHECK(c && info, GLX_EINVAL, "glx_dist_comm_info: null argument");
  info[0] = c->rank;
  info[1] = c->nranks;
  info[2] = c->device;
  info[3] = c->comm ? 1 : 0;
  return GLX_OK;
}

extern "C" int glx_dist_destroy(glx_comm* c) {
  if (!c) return GLX_OK;
  {
    std::vector<glx_dist_sweep*> live;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      live.swap(c->sweeps);
    }
    for (glx_dist_sweep* s : live) detach_sweep(s);      // graphs with RCCL kernels go before the communicator does
  }
  if (c->comm) {
    hipSetDevice(c->device);
    RcclApi* a = rccl_api();
    if (a) a->CommDestroy(c->comm);
  }
  delete c;
  return GLX_OK;
}

// ---- the rank-local sweep -------------------------------------------------------------------------------
static const int ERR_SHARDS = 64;
static const int ERR_SLOTS = 65;     // check_every <= 64

struct glx_dist_sweep {
  glx_comm* comm = nullptr;
  int device = 0, dtype = GLX_F64, C = 0;
  RecLayout L;
  int64_t n_own = 0, n_halo = 0, nb = 0, n_loc = 0, n_global = 0;
  glx_graph* part[2] = {nullptr, nullptr};   // boundary rows [0, nb), interior rows [nb, n_own)
  SellPlan* plan[2] = {nullptr, nullptr};
  uint8_t* flags[2] = {nullptr, nullptr};
  std::vector<void*> ring;                   // state buffers of n_loc records
  void* bias = nullptr;                      // n_own records
  void* init_rec = nullptr;                  // n_own records: u = 0 (ssl.py:645), w = w0
  double *deg = nullptr, *vinf = nullptr;
  void* dense = nullptr;                     // staging (n_own, C)
  // exchange
  bool exchange = false;                     // some rank imports something (or forced): every sweep has an exchange
  std::vector<int64_t> send_cnt, recv_cnt, send_off, recv_off;   // records, per peer
  int64_t n_send = 0;
  int32_t* send_idx = nullptr;               // [n_send] local row of every record sent, grouped by destination
  void* sendbuf = nullptr;
  unsigned long long *err = nullptr, *h_err = nullptr;   // [ERR_SLOTS * 64] maxima per sweep: slot 0 = last head sweep, 1.. = the chunk's
  bool warmed = false;
  hipStream_t stream = nullptr, xstream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_x = nullptr, ev0 = nullptr, ev1 = nullptr;
  // captured launch sequences, keyed (kind, a, b, c): kind 0 = head (a = sweeps), kind 1 = tail chunk (a = ring size, b = first
  // buffer, c = sweeps), kind 2 = self-test.  Separate key components: no two sequences can share a key.
  std::map<std::array<long, 4>, hipGraphExec_t> graphs;
  bool use_graph = true;
  bool overlap = true;                       // exchange on its own stream beside the interior rows (else in line on the sweep's stream)
  int capture_exchange = 1;                  // sweeps that carry an RCCL exchange: 1 captured into device graphs, 0 enqueued eagerly,
                                             // -1 undecided -- the first run's self-test (captured vs eager sweeps, bit for bit, all ranks) decides
  int selftest = 0;                          // 0 not run, 1 passed (captured), 2 failed (eager)
  bool fused = false;                        // ONE launch for all rows of a sweep (part[0] = every row), exchange after it
  bool scatter = true;                       // the boundary SpMM stores its rows into the send buffer itself (no pack kernel)
  int32_t *dup_ptr = nullptr, *dup_pos = nullptr;   // [rows of part 0 + 1], [n_send]: send-buffer positions of every row
  void* st_ref = nullptr;                    // self-test: the eager result
  unsigned int* st_diff = nullptr;           // self-test: mismatch counter (device) -- made global with an all-reduce
  // GATHER form (GLX_DIST_FORM_GATHER; SURVEY 8e's fallback when the halo is about all rows): the state is nranks blocks of `cap`
  // records in rank order, this rank's rows in block `rank` (own_off = rank * cap); the exchange is ONE in-place ncclAllGather of the
  // blocks -- no send lists, no pack, no send buffer
  bool gather = false;
  bool gather_next = false;                  // stepwise form: glx_dist_sweep_boundary has written the NEXT iterate's block (what get_send hands out)
  int64_t cap = 0, own_off = 0;
  bool problem_set = false;
  int cur = 0;                               // ring index of the current iterate
  int64_t sweeps_run = 0, exchanges = 0;
  double thresh = 0.0;
};

static int64_t part_lo1(const glx_dist_sweep* s) { return s->fused ? s->n_own : s->nb; }   // first row of part 1
static size_t recb(const glx_dist_sweep* s, int64_t rows) { return std::max<size_t>((size_t)rows * s->L.ld * s->L.esize, 64); }
static char* rec_at(void* base, const glx_dist_sweep* s, int64_t row) { return (char*)base + (size_t)row * s->L.ld * s->L.esize; }

// the communicator is going away: finish the sweep's work, release its captured graphs, forget the communicator (later
// collective calls on the sweep fail with an error; fetch / destroy still work)
static void detach_sweep(glx_dist_sweep* s) {
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->xstream) hipStreamSynchronize(s->xstream);
  for (auto& kv : s->graphs) hipGraphExecDestroy(kv.second);
  s->graphs.clear();
  s->comm = nullptr;
}

extern "C" int glx_dist_sweep_destroy(glx_dist_sweep* s) {
  if (!s) return GLX_OK;
  if (s->comm) {
    std::lock_guard<std::mutex> lk(s->comm->mu);
    auto& v = s->comm->sweeps;
    v.erase(std::remove(v.begin(), v.end(), s), v.end());
  }
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->xstream) hipStreamSynchronize(s->xstream);
  for (auto& kv : s->graphs) hipGraphExecDestroy(kv.second);
  for (int q = 0; q < 2; ++q) {
    glx_graph_destroy(s->part[q]);
    hipFree(s->flags[q]);
  }
  for (void* b : s->ring) hipFree(b);
  hipFree(s->bias);
  hipFree(s->init_rec);
  hipFree(s->deg);
  hipFree(s->vinf);
  hipFree(s->dense);
  hipFree(s->send_idx);
  hipFree(s->sendbuf);
  hipFree(s->dup_ptr);
  hipFree(s->dup_pos);
  hipFree(s->st_ref);
  hipFree(s->st_diff);
  hipFree(s->err);
  if (s->h_err) hipHostFree(s->h_err);
  if (s->ev_pack) hipEventDestroy(s->ev_pack);
  if (s->ev_x) hipEventDestroy(s->ev_x);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->stream) hipStreamDestroy(s->stream);
  if (s->xstream) hipStreamDestroy(s->xstream);
  delete s;
  return GLX_OK;
}

// rowptr / col / val: this rank's rows of P (boundary rows first), columns renumbered [0, n_own) owned,
// [n_own, n_own + n_halo) halo in the order the peers' records arrive (grouped by owner rank, ascending).
extern "C" int glx_dist_sweep_create(glx_comm* comm, int64_t n_own, int64_t n_halo, int64_t n_boundary, const int32_t* rowptr,
                                     const int32_t* col, const double* val, int state_dtype, int C, const int64_t* send_counts,
                                     const int32_t* send_idx, const int64_t* recv_counts, int64_t n_global, int force_exchange,
                                     int flags, glx_dist_sweep** out) {
  GLX_CHECK(comm && out && rowptr, GLX_EINVAL, "glx_dist_sweep_create: null argument");
  *out = nullptr;
  GLX_CHECK(n_own >= 0 && n_halo >= 0 && n_boundary >= 0 && n_boundary <= n_own, GLX_EINVAL, "glx_dist_sweep_create: bad sizes");
  const int nr = comm->nranks;
  const bool gather = (flags & GLX_DIST_FORM_GATHER) != 0;
  int64_t gcap = n_own;
  if (gather) {
    // columns are numbered owner * cap + (row within the owner's block); n_halo = (nranks - 1) * cap; every row is a boundary row
    GLX_CHECK(nr == 1 ? n_halo == 0 : n_halo % (nr - 1) == 0, GLX_EINVAL, "glx_dist_sweep_create: gather form: n_halo must be (nranks - 1) * cap");
    gcap = nr == 1 ? n_own : n_halo / (nr - 1);
    GLX_CHECK(n_own <= gcap && n_boundary == n_own, GLX_EINVAL, "glx_dist_sweep_create: gather form: n_own <= cap and n_boundary = n_own");
  }
  static const int64_t zero_counts[1024] = {0};
  if (gather) {
    GLX_CHECK(nr <= 1024, GLX_EUNSUPPORTED, "glx_dist_sweep_create: more than 1024 ranks");
    send_counts = zero_counts;
    recv_counts = zero_counts;
    send_idx = nullptr;
  }
  GLX_CHECK(send_counts && recv_counts, GLX_EINVAL, "glx_dist_sweep_create: null exchange lists");
  int64_t ns = 0, nrcv = 0;
  for (int r = 0; r < nr; ++r) {
    GLX_CHECK(send_counts[r] >= 0 && recv_counts[r] >= 0, GLX_EINVAL, "glx_dist_sweep_create: negative count");
    ns += send_counts[r];
    nrcv += recv_counts[r];
  }
  GLX_CHECK(gather || nrcv == n_halo, GLX_EINVAL, "glx_dist_sweep_create: receive counts sum to %lld, halo is %lld", (long long)nrcv, (long long)n_halo);
  GLX_CHECK(ns == 0 || send_idx, GLX_EINVAL, "glx_dist_sweep_create: null send list");
  for (int64_t q = 0; q < ns; ++q)
    GLX_CHECK(send_idx[q] >= 0 && send_idx[q] < n_boundary, GLX_EINVAL, "glx_dist_sweep_create: send row %d is not a boundary row", send_idx[q]);
  GLX_HIP(hipSetDevice(comm->device));
  glx_dist_sweep* s = new glx_dist_sweep();
  s->comm = comm;
  s->device = comm->device;
  s->dtype = state_dtype;
  s->C = C;
  s->n_own = n_own;
  s->n_halo = n_halo;
  s->nb = n_boundary;
  s->n_loc = gather ? gcap * nr : n_own + n_halo;
  s->gather = gather;
  s->cap = gcap;
  s->own_off = gather ? gcap * comm->rank : 0;
  s->n_global = n_global;
  s->use_graph = (flags & GLX_DIST_CAPTURE) != 0;
  {
    // HIP runtimes before 7.2 (e.g. the 7.0 copy bundled with torch, which wins when torch is imported first) recurse
    // without end in hipStreamEndCapture when a second stream forks from and joins back into the capturing stream:
    // captured sweeps then keep the exchange in line on the one stream; eager sweeps overlap on any runtime.
    int rt = 0;
    hipRuntimeGetVersion(&rt);
    s->overlap = !(s->use_graph && rt < 70200000);
    const bool inline_exchange = (flags & GLX_DIST_INLINE) != 0;      // the exchange on the sweep's own stream, whatever the form
    if (inline_exchange) s->overlap = false;
    // Grouped ncclSend/ncclRecv inside a stream capture has been exercised on ONE rank only (self exchange, RCCL 2.26.6 and
    // 2.27.7); with real peers the first run decides by a self-test (three captured exchanging sweeps, replayed, against three
    // eager ones: bit for bit, on every rank, with a deadline); GLX_DIST_EXCHANGE_CAPTURED / _EAGER / _SELFTEST settle it up front.
    // Sweeps without an exchange (no halo anywhere) are captured either way.
    s->capture_exchange = comm->nranks == 1 ? 1 : -1;
    if (flags & GLX_DIST_EXCHANGE_CAPTURED) s->capture_exchange = 1;
    if (flags & GLX_DIST_EXCHANGE_EAGER) s->capture_exchange = 0;
    if (flags & GLX_DIST_EXCHANGE_SELFTEST) s->capture_exchange = -1;
    if (!s->use_graph) s->capture_exchange = 0;
    if (s->capture_exchange == 0 && !inline_exchange) s->overlap = true;   // eager: two streams are safe on every runtime
    if (flags & GLX_DIST_PACK_KERNEL) s->scatter = false;                  // the round-2 pack kernel between SpMM and transport
    // Form of a sweep.  SPLIT: [boundary rows | exchange on a second stream beside the interior rows] hides min(interior, exchange)
    // but pays for a second launch (~9 us at 70 000 rows: a short launch is a chain of dependent memory round trips) and for the
    // cross-stream edges of the captured graph (~14 us measured: 35.0 vs 29.4 us per sweep with the exchange in line).  FUSED: ONE
    // launch for all rows, the exchange in line behind it: 22.9 us per sweep for the same 10 000-record halo (profiles/r03_dist_probe.txt).
    // The split form wins only when BOTH the interior rows and the exchange take longer than those fixed costs (~23 us): estimated
    // from the interior's stored entries (13 ps per entry) and the largest per-peer message (8 us + bytes / 50 GB/s).
    // GLX_DIST_FORM_SPLIT / GLX_DIST_FORM_FUSED force a form.
    double interior_us = 0.0, exchange_us = 0.0;
    {
      const double nnz_int = (double)(rowptr[n_own] - rowptr[n_boundary]);
      interior_us = nnz_int * 13e-6;
      int64_t peer_max = 0;
      for (int r = 0; r < nr; ++r) peer_max = std::max(peer_max, std::max(send_counts[r], recv_counts[r]));
      RecLayout L0;
      if (glx_make_layout(C, state_dtype, true, &L0) == GLX_OK)
        exchange_us = peer_max > 0 ? 8.0 + (double)peer_max * L0.ld * L0.esize / 50e3 : 0.0;
    }
    s->fused = n_own > 0 && std::min(interior_us, exchange_us) < 23.0;
    if (flags & GLX_DIST_FORM_SPLIT) s->fused = false;
    if (flags & GLX_DIST_FORM_FUSED) s->fused = n_own > 0;
    if (gather) s->fused = n_own > 0;   // every row is gathered by the peers: one launch, then the all-gather
    if (s->fused) s->overlap = false;   // nothing to run beside the exchange
  }
  s->thresh = 1.0 / (double)n_global;   // `> 1/n`, ssl.py:667, n = ALL vertices
  int rc = glx_make_layout(C, state_dtype, true, &s->L);
  if (rc) { delete s; return rc; }
#define DS_FAIL(code) do { glx_dist_sweep_destroy(s); return (code); } while (0)
#define DS_UP(call) do { const int rc_up_ = (call); if (rc_up_) DS_FAIL(rc_up_); } while (0)
#define DS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { glx_set_error("%s -> %s", #call, hipGetErrorString(e_)); DS_FAIL(GLX_EHIP); } } while (0)
  // the two operators over the same local vector
  const int64_t lo[2] = {0, s->fused ? n_own : n_boundary}, hi[2] = {s->fused ? n_own : n_boundary, n_own};
  for (int q = 0; q < 2; ++q) {
    if (hi[q] <= lo[q]) continue;
    std::vector<int32_t> rp(hi[q] - lo[q] + 1);
    for (int64_t i = lo[q]; i <= hi[q]; ++i) rp[i - lo[q]] = rowptr[i] - rowptr[lo[q]];
    const int64_t nnz = rp.back();
    // (resident: the arrays go to the device as they are and the plan is filled there -- no host copy of the rank's 10^8 entries)
    rc = glx_graph_create_resident(hi[q] - lo[q], s->n_loc, nnz, rp.data(), col + rowptr[lo[q]], val + rowptr[lo[q]], state_dtype, comm->device, nullptr,
                                   &s->part[q]);
    if (rc) DS_FAIL(rc);
    glx_graph_set_order(s->part[q], nullptr);      // (the caller's order: rank-local operators are rectangular and never renumbered)
    rc = glx_graph_plan(s->part[q], s->L.G, &s->plan[q]);
    if (rc) DS_FAIL(rc);
    DS_HIP(hipMalloc(&s->flags[q], std::max<size_t>((size_t)s->plan[q]->nslices * s->plan[q]->R, 64)));
  }
  DS_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  DS_HIP(hipStreamCreateWithFlags(&s->xstream, hipStreamNonBlocking));
  DS_HIP(hipEventCreateWithFlags(&s->ev_pack, hipEventDisableTiming));
  DS_HIP(hipEventCreateWithFlags(&s->ev_x, hipEventDisableTiming));
  DS_HIP(hipEventCreate(&s->ev0));
  DS_HIP(hipEventCreate(&s->ev1));
  for (int b = 0; b < 2; ++b) {
    void* p = nullptr;
    DS_HIP(hipMalloc(&p, recb(s, s->n_loc)));
    s->ring.push_back(p);
    DS_HIP(hipMemsetAsync(p, 0, recb(s, s->n_loc), s->stream));
  }
  DS_HIP(hipMalloc(&s->bias, recb(s, n_own)));
  DS_HIP(hipMalloc(&s->init_rec, recb(s, n_own)));
  DS_HIP(hipMalloc(&s->deg, std::max<size_t>(n_own * 8, 64)));
  DS_HIP(hipMalloc(&s->vinf, std::max<size_t>(n_own * 8, 64)));
  DS_HIP(hipMalloc(&s->dense, std::max<size_t>((size_t)n_own * std::max(C * s->L.esize, 16), 64)));
  // exchange lists
  s->send_cnt.assign(send_counts, send_counts + nr);
  s->recv_cnt.assign(recv_counts, recv_counts + nr);
  s->send_off.assign(nr + 1, 0);
  s->recv_off.assign(nr + 1, 0);
  for (int r = 0; r < nr; ++r) {
    s->send_off[r + 1] = s->send_off[r] + send_counts[r];
    s->recv_off[r + 1] = s->recv_off[r] + recv_counts[r];
  }
  s->n_send = ns;
  s->exchange = force_exchange != 0 || ns > 0 || n_halo > 0;   // the planner passes force_exchange = "some rank has a halo"
  if (gather) s->scatter = false;                              // (nothing to scatter: the rows are where the all-gather reads them)
  DS_HIP(hipMalloc(&s->send_idx, std::max<size_t>((size_t)ns * 4, 64)));
  if (ns > 0) DS_UP(glx_upload_sync(s->send_idx, send_idx, (size_t)ns * 4, __func__));
  DS_HIP(hipMalloc(&s->sendbuf, recb(s, ns)));
  {
    // rows of part 0 -> their positions in the send buffer (a row needed by several peers has several)
    const int64_t nr0 = hi[0];
    std::vector<int32_t> dp(nr0 + 1, 0), dq(std::max<int64_t>(ns, 1));
    for (int64_t q = 0; q < ns; ++q) dp[send_idx[q] + 1]++;
    for (int64_t i = 0; i < nr0; ++i) dp[i + 1] += dp[i];
    std::vector<int32_t> fill(dp.begin(), dp.end() - 1);
    for (int64_t q = 0; q < ns; ++q) dq[fill[send_idx[q]]++] = (int32_t)q;
    DS_HIP(hipMalloc(&s->dup_ptr, (size_t)(nr0 + 1) * 4));
    DS_HIP(hipMalloc(&s->dup_pos, dq.size() * 4));
    DS_UP(glx_upload_sync(s->dup_ptr, dp.data(), (size_t)(nr0 + 1) * 4, __func__));
    DS_UP(glx_upload_sync(s->dup_pos, dq.data(), dq.size() * 4, __func__));
  }
  DS_HIP(hipMalloc(&s->err, (size_t)ERR_SLOTS * ERR_SHARDS * 8));
  DS_HIP(hipMemsetAsync(s->err, 0, (size_t)ERR_SLOTS * ERR_SHARDS * 8, s->stream));
  DS_HIP(hipHostMalloc((void**)&s->h_err, (size_t)ERR_SLOTS * ERR_SHARDS * 8, hipHostMallocDefault));
  DS_HIP(hipStreamSynchronize(s->stream));
#undef DS_HIP
#undef DS_FAIL
  {
    std::lock_guard<std::mutex> lk(comm->mu);
    comm->sweeps.push_back(s);
  }
  *out = s;
  return GLX_OK;
}

extern "C" int glx_dist_sweep_set_problem(glx_dist_sweep* s, const void* Db_own, const double* w0_own, const double* deg_own,
                                          const double* vinf_own) {
  GLX_CHECK(s && w0_own && deg_own && vinf_own, GLX_EINVAL, "glx_dist_sweep_set_problem: null argument");
  GLX_HIP(hipSetDevice(s->device));
  for (auto& kv : s->graphs) hipGraphExecDestroy(kv.second);   // captured sequences bake in whether a bias is read
  s->graphs.clear();
  const size_t es = s->L.esize;
  int rc;
  if (Db_own) {
    GLX_UP(glx_upload(s->dense, Db_own, (size_t)s->n_own * s->C * es, s->stream, __func__));
    rc = glx_pack_records(s->dense, s->bias, s->n_own, s->L, s->dtype, nullptr, s->stream);
  } else {
    rc = glx_pack_records(nullptr, s->bias, s->n_own, s->L, s->dtype, nullptr, s->stream);
  }
  if (rc) return rc;
  for (int q = 0; q < 2; ++q) {
    if (!s->part[q]) continue;
    const int64_t lo = q == 0 ? 0 : part_lo1(s);
    rc = glx_bias_flags_dev(s->part[q], s->C, 1, rec_at(s->bias, s, lo), s->flags[q], s->stream);
    if (rc) return rc;
  }
  GLX_HIP(hipStreamSynchronize(s->stream));   // `dense` is reused
  GLX_UP(glx_upload(s->dense, w0_own, s->n_own * 8, s->stream, __func__));
  rc = glx_pack_records(nullptr, s->init_rec, s->n_own, s->L, s->dtype, (const double*)s->dense, s->stream);   // u = 0, w = w0
  if (rc) return rc;
  GLX_UP(glx_upload(s->deg, deg_own, s->n_own * 8, s->stream, __func__));
  GLX_UP(glx_upload(s->vinf, vinf_own, s->n_own * 8, s->stream, __func__));
  GLX_HIP(hipStreamSynchronize(s->stream));
  s->problem_set = true;
  return GLX_OK;
}

// ---- pieces of one sweep ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int32_t* __restrict__ idx,
                                                          int64_t nrec, int rec_bytes) {
  // 16 bytes per thread, rec_bytes / 16 threads per record
  const int per = rec_bytes / 16;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nrec * per) return;
  const int64_t q = i / per;
  const int o = (int)(i % per) * 16;
  *(uint4*)(dst + (size_t)q * rec_bytes + o) = *(const uint4*)(src + (size_t)idx[q] * rec_bytes + o);
}

static int launch_part(glx_dist_sweep* s, int q, const void* xin, void* xout, unsigned long long* err_next) {
  if (!s->part[q]) return GLX_OK;
  const int64_t lo = q == 0 ? 0 : part_lo1(s);
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.plan = s->plan[q];
  a.L = s->L;
  a.dtype = s->dtype;
  a.xin = xin;
  a.xout = rec_at(xout, s, s->own_off + lo);
  a.bias = rec_at(s->bias, s, lo);
  a.slot_has_bias = s->flags[q];
  a.has_w = true;
  a.n_rows = s->part[q]->n_rows;
  a.deg = s->deg + lo;
  a.vinf = s->vinf + lo;
  a.err_next = err_next;
  a.thresh = s->thresh;
  if (q == 0 && s->scatter && s->n_send > 0) {   // boundary rows leave for the send buffer from the kernel's registers
    a.dup_ptr = s->dup_ptr;
    a.dup_pos = s->dup_pos;
    a.dup_out = s->sendbuf;
  }
  return glx_launch_spmm(a, s->stream);
}

static int enqueue_pack(glx_dist_sweep* s, const void* x) {
  if (s->n_send == 0) return GLX_OK;
  const int rb = s->L.ld * s->L.esize;
  const int64_t tot = s->n_send * (rb / 16);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (const char*)x, (char*)s->sendbuf,
                     (const int32_t*)s->send_idx, s->n_send, rb);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// boundary records of x -> the peers' halo regions of their x (this rank's x[n_own:] receives).  Runs on the exchange
// stream behind the pack; the caller joins with wait_exchange().
static int enqueue_exchange(glx_dist_sweep* s, void* x, bool packed = false) {
  if (!s->exchange) return GLX_OK;
  if (s->gather) {
    // every rank's block straight into every rank's state: in place, the send part is the rank's own block
    if (s->comm->comm) {
      RcclApi* a = rccl_api();
      const size_t bytes = (size_t)s->cap * s->L.ld * s->L.esize;
      GLX_NCCL(a->AllGather(rec_at(x, s, s->own_off), x, bytes, ncclUint8, s->comm->comm, s->stream));
    }
    s->exchanges++;
    return GLX_OK;
  }
  int rc = packed ? GLX_OK : enqueue_pack(s, x);   // packed: the boundary SpMM has already filled the send buffer
  if (rc) return rc;
  hipStream_t xs = s->overlap ? s->xstream : s->stream;
  if (s->overlap) {
    GLX_HIP(hipEventRecord(s->ev_pack, s->stream));
    GLX_HIP(hipStreamWaitEvent(s->xstream, s->ev_pack, 0));
  }
  const size_t rb = (size_t)s->L.ld * s->L.esize;
  if (s->comm->comm) {
    RcclApi* a = rccl_api();
    GLX_NCCL(a->GroupStart());
    for (int r = 0; r < s->comm->nranks; ++r) {
      if (s->send_cnt[r] > 0)
        GLX_NCCL(a->Send((const char*)s->sendbuf + s->send_off[r] * rb, (size_t)s->send_cnt[r] * rb, ncclUint8, r, s->comm->comm, xs));
      if (s->recv_cnt[r] > 0)
        GLX_NCCL(a->Recv(rec_at(x, s, s->n_own + s->recv_off[r]), (size_t)s->recv_cnt[r] * rb, ncclUint8, r, s->comm->comm, xs));
    }
    GLX_NCCL(a->GroupEnd());
  } else if (s->n_send > 0) {   // one rank without a communicator: what it "sends" to itself lands in its own halo
    GLX_HIP(hipMemcpyAsync(rec_at(x, s, s->n_own), s->sendbuf, (size_t)std::min(s->n_send, s->n_halo) * rb, hipMemcpyDeviceToDevice, xs));
  }
  if (s->overlap) GLX_HIP(hipEventRecord(s->ev_x, s->xstream));
  s->exchanges++;
  return GLX_OK;
}

static int wait_exchange(glx_dist_sweep* s) {
  if (!s->exchange || !s->overlap) return GLX_OK;
  GLX_HIP(hipStreamWaitEvent(s->stream, s->ev_x, 0));
  return GLX_OK;
}

// one sweep xin -> xout; err_next: 64 shards receiving the rank-local max |deg w - vinf| of the new iterate (or null)
static int enqueue_sweep(glx_dist_sweep* s, const void* xin, void* xout, unsigned long long* err_next) {
  int rc = launch_part(s, 0, xin, xout, err_next);
  if (rc) return rc;
  rc = enqueue_exchange(s, xout, s->scatter);
  if (rc) return rc;
  rc = launch_part(s, 1, xin, xout, err_next);
  if (rc) return rc;
  return wait_exchange(s);
}

// state <- initial records, halo filled by one exchange
static int enqueue_reset(glx_dist_sweep* s, void* x) {
  GLX_HIP(hipMemcpyAsync(rec_at(x, s, s->own_off), s->init_rec, (size_t)s->n_own * s->L.ld * s->L.esize, hipMemcpyDeviceToDevice, s->stream));
  int rc = enqueue_exchange(s, x);
  if (rc) return rc;
  return wait_exchange(s);
}

static int ensure_ring(glx_dist_sweep* s, int nbuf) {
  while ((int)s->ring.size() < nbuf) {
    void* p = nullptr;
    GLX_HIP(hipMalloc(&p, recb(s, s->n_loc)));
    GLX_HIP(hipMemsetAsync(p, 0, recb(s, s->n_loc), s->stream));
    s->ring.push_back(p);
  }
  return GLX_OK;
}

// run `fn` (a sequence of enqueues on s->stream / s->xstream) through a captured device graph keyed by `key`
template <typename F>
static int run_captured(glx_dist_sweep* s, const std::array<long, 4>& key, F fn) {
  if (!s->use_graph || (s->exchange && s->capture_exchange != 1)) return fn();
  auto it = s->graphs.find(key);
  if (it == s->graphs.end()) {
    hipGraph_t graph;
    GLX_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int rc = fn();
    hipError_t e = hipStreamEndCapture(s->stream, &graph);
    if (rc) return rc;
    GLX_HIP(e);
    hipGraphExec_t exec;
    GLX_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    GLX_HIP(hipGraphDestroy(graph));
    it = s->graphs.emplace(key, exec).first;
  }
  GLX_HIP(hipGraphLaunch(it->second, s->stream));
  return GLX_OK;
}


// ---- self-test of captured exchanging sweeps -------------------------------------------------------------------
// Grouped ncclSend / ncclRecv inside a stream capture is documented RCCL usage, but this library could only ever run it with
// ONE rank (1-GPU test boxes).  So with real peers nothing is assumed: the first run executes three exchanging sweeps
// eagerly, then the same three through a captured graph -- launched twice, a replay is what the measured loop does -- and
// compares the iterates bit for bit; the verdicts are combined over the ranks (ncclAllReduce MAX of the mismatch counts)
// and only a clean pass selects the captured path.  A capture that hangs is cut off by a deadline instead of by the job's
// time limit (the call then fails with GLX_ERCCL and the caller can fall back to another engine).
__global__ __launch_bounds__(256) void count_diff_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b,
                                                         int64_t nwords, unsigned int* __restrict__ out) {
  unsigned int d = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) d += a[i] != b[i];
  if (d) atomicAdd(out, d);
}

static int sync_with_deadline(glx_dist_sweep* s, double seconds, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipStreamQuery(s->stream);
    if (e == hipSuccess) return GLX_OK;
    if (e != hipErrorNotReady) { glx_set_error("%s: %s", what, hipGetErrorString(e)); return GLX_EHIP; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
      glx_set_error("%s: no completion within %.0f s (rank %d)", what, seconds, s->comm->rank);
      return GLX_ERCCL;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

static int exchange_selftest(glx_dist_sweep* s) {
  const int K = 3;
  double deadline = 30.0;
  if (const char* e = getenv("GLX_DIST_SELFTEST_TIMEOUT")) deadline = atof(e);
  const size_t bytes = (size_t)s->n_loc * s->L.ld * s->L.esize;
  if (!s->st_ref) GLX_HIP(hipMalloc(&s->st_ref, std::max<size_t>(bytes, 64)));
  if (!s->st_diff) GLX_HIP(hipMalloc((void**)&s->st_diff, 64));
  auto body = [&]() -> int {
    int r2 = enqueue_reset(s, s->ring[0]);
    for (int t = 0; t < K && !r2; ++t) r2 = enqueue_sweep(s, s->ring[t & 1], s->ring[(t & 1) ^ 1], nullptr);
    return r2;
  };
  int rc = body();                                      // eager: plain RCCL usage
  if (rc) return rc;
  GLX_HIP(hipMemcpyAsync(s->st_ref, s->ring[K & 1], bytes, hipMemcpyDeviceToDevice, s->stream));
  GLX_HIP(hipMemsetAsync(s->st_diff, 0, 64, s->stream));
  rc = sync_with_deadline(s, deadline, "self-test, eager sweeps");
  if (rc) return rc;
  const int saved = s->capture_exchange;
  s->capture_exchange = 1;
  unsigned int h_diff = 0;
  for (int pass = 0; pass < 2 && !rc; ++pass) {         // capture + launch, then a replay
    GLX_HIP(hipMemsetAsync(s->ring[K & 1], 0xff, bytes, s->stream));   // the captured sweeps must really write it
    rc = run_captured(s, {2L, (long)K, 0L, 0L}, body);
    if (rc) break;
    if (bytes > 0) {
      hipLaunchKernelGGL(count_diff_kernel, dim3((unsigned)std::min<int64_t>(1024, (int64_t)(bytes / 8 + 255) / 256)), dim3(256), 0, s->stream,
                         (const unsigned long long*)s->st_ref, (const unsigned long long*)s->ring[K & 1], (int64_t)(bytes / 8), s->st_diff);
      GLX_HIP(hipGetLastError());
    }
    rc = sync_with_deadline(s, deadline, "self-test, captured sweeps");
  }
  s->capture_exchange = saved;
  auto it = s->graphs.find({2L, (long)K, 0L, 0L});
  if (it != s->graphs.end()) { hipGraphExecDestroy(it->second); s->graphs.erase(it); }
  if (rc) return rc;
  if (s->comm->comm) {                                   // one verdict for all ranks
    RcclApi* a = rccl_api();
    GLX_NCCL(a->AllReduce(s->st_diff, s->st_diff, 1, ncclUint32, ncclMax, s->comm->comm, s->stream));
  }
  GLX_HIP(hipMemcpyAsync(&h_diff, s->st_diff, 4, hipMemcpyDeviceToHost, s->stream));
  rc = sync_with_deadline(s, deadline, "self-test, verdict");
  if (rc) return rc;
  s->selftest = h_diff == 0 ? 1 : 2;
  s->capture_exchange = h_diff == 0 ? 1 : 0;
  if (s->capture_exchange == 0) s->overlap = true;
  hipFree(s->st_ref);
  s->st_ref = nullptr;
  return GLX_OK;
}

static double shard_max(const unsigned long long* h, int64_t t) {
  unsigned long long m = 0;
  for (int k = 0; k < ERR_SHARDS; ++k) m = std::max(m, h[(size_t)t * ERR_SHARDS + k]);
  return __builtin_bit_cast(double, m);
}

// make err[t0 .. t0+cnt) global (max over ranks) and bring it to the host
static int global_err(glx_dist_sweep* s, int64_t t0, int64_t cnt) {
  unsigned long long* p = s->err + (size_t)t0 * ERR_SHARDS;
  if (s->comm->comm) {   // (also with one rank: the forced-collective tests run the reduction for real)
    RcclApi* a = rccl_api();
    // fp64 bit patterns of non-negative values (and of the NaN marker, which sorts above +inf) order like uint64
    GLX_NCCL(a->AllReduce(p, p, (size_t)cnt * ERR_SHARDS, ncclUint64, ncclMax, s->comm->comm, s->stream));
  }
  GLX_HIP(hipMemcpyAsync(s->h_err + (size_t)t0 * ERR_SHARDS, p, (size_t)cnt * ERR_SHARDS * 8, hipMemcpyDeviceToHost, s->stream));
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// All sweeps.  err0: max |v0 - vinf| over ALL vertices (read only when min_iter = 0).  check_every: sweeps per
// stop-test chunk past min_iter (1..64).
extern "C" int glx_poisson_sweep_dist(glx_dist_sweep* s, int min_iter, int max_iter, int check_every, double err0, int* T_out,
                                      float* device_ms_out) {
  GLX_CHECK(s && s->problem_set, GLX_EINVAL, "glx_poisson_sweep_dist: set the problem first");
  GLX_CHECK(s->comm, GLX_EINVAL, "glx_poisson_sweep_dist: the communicator of this sweep has been destroyed");
  GLX_CHECK(min_iter >= 0 && max_iter >= 0 && check_every >= 1 && check_every < ERR_SLOTS, GLX_EINVAL,
            "glx_poisson_sweep_dist: bad iteration bounds");
  GLX_CHECK(s->comm->comm || s->comm->nranks == 1, GLX_EINVAL, "glx_poisson_sweep_dist: this communicator has no transport (use the stepwise form)");
  GLX_HIP(hipSetDevice(s->device));
  int rc;
  if (!s->warmed) {
    // first use: one eager exchange and reduction, so that RCCL sets up its peer connections outside any capture
    if (s->exchange) {
      rc = enqueue_reset(s, s->ring[0]);
      if (rc) return rc;
    }
    rc = global_err(s, 0, 1);
    if (rc) return rc;
    if (s->capture_exchange < 0) {
      if (s->exchange && s->use_graph) {
        rc = exchange_selftest(s);
        if (rc) return rc;
      } else {
        s->capture_exchange = 1;                         // nothing to exchange: sweeps are plain launches
      }
    }
    s->warmed = true;
  }
  const int head = std::min(min_iter, max_iter);
  const bool head_err = head > 0 && head >= min_iter;   // the last head sweep produces v_min_iter: its error decides sweep min_iter + 1
  GLX_HIP(hipEventRecord(s->ev0, s->stream));
  // head: reset + the sweeps the stop test cannot cut short, ping-pong on ring[0] / ring[1]
  rc = run_captured(s, {0L, (long)head, 0L, 0L}, [&]() -> int {
    // (a kernel, not a memset node: captured and replayed -- glx_zero_async, glx_internal.h)
    { int rz = glx_zero_async(s->err, ERR_SHARDS * 8, s->stream); if (rz) return rz; }
    int r2 = enqueue_reset(s, s->ring[0]);
    for (int t = 0; t < head && !r2; ++t)
      r2 = enqueue_sweep(s, s->ring[t & 1], s->ring[(t & 1) ^ 1], (head_err && t + 1 == head) ? s->err : nullptr);
    return r2;
  });
  if (rc) return rc;
  s->cur = head & 1;
  s->sweeps_run += head;
  int T = head;
  double err_T = err0;
  if (head < max_iter) {
    if (head_err) {
      rc = global_err(s, 0, 1);
      if (rc) return rc;
      err_T = shard_max(s->h_err, 0);
    }
    const int R = check_every + 1;
    while (T < max_iter && err_T > s->thresh) {   // a NaN error ends the loop like `nan > 1/n` (ssl.py:667)
      rc = ensure_ring(s, R);
      if (rc) return rc;
      const int cnt = std::min(check_every, max_iter - T);
      const int cur0 = s->cur;
      rc = run_captured(s, {1L, (long)R, (long)cur0, (long)cnt}, [&]() -> int {   // the ring size is part of the buffers a chunk touches
        { int rz = glx_zero_async(s->err + ERR_SHARDS, (size_t)cnt * ERR_SHARDS * 8, s->stream); if (rz) return rz; }
        int r2 = GLX_OK;
        for (int j = 0; j < cnt && !r2; ++j)
          r2 = enqueue_sweep(s, s->ring[(cur0 + j) % R], s->ring[(cur0 + j + 1) % R], s->err + (size_t)(j + 1) * ERR_SHARDS);
        return r2;
      });
      if (rc) return rc;
      s->sweeps_run += cnt;
      rc = global_err(s, 1, cnt);
      if (rc) return rc;
      int q = 1;
      for (; q <= cnt; ++q) {
        err_T = shard_max(s->h_err, q);
        if (!(err_T > s->thresh)) break;
      }
      if (q > cnt) q = cnt;             // no stop inside the chunk: go on from its last iterate
      T += q;
      s->cur = (cur0 + q) % R;
    }
  }
  GLX_HIP(hipEventRecord(s->ev1, s->stream));
  GLX_HIP(hipStreamSynchronize(s->stream));
  if (device_ms_out) GLX_HIP(hipEventElapsedTime(device_ms_out, s->ev0, s->ev1));
  if (T_out) *T_out = T;
  return GLX_OK;
}

extern "C" int glx_dist_sweep_fetch(glx_dist_sweep* s, void* u_own_out) {
  GLX_CHECK(s && u_own_out, GLX_EINVAL, "glx_dist_sweep_fetch: null argument");
  GLX_HIP(hipSetDevice(s->device));
  int rc = glx_unpack_records(rec_at(s->ring[s->cur], s, s->own_off), s->dense, s->n_own, s->L, s->dtype, s->stream);
  if (rc) return rc;
  GLX_UP(glx_download(u_own_out, s->dense, (size_t)s->n_own * s->C * s->L.esize, s->stream, __func__));
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

extern "C" int glx_dist_sweep_stats(const glx_dist_sweep* s, int64_t out[4]) {
  GLX_CHECK(s && out, GLX_EINVAL, "glx_dist_sweep_stats: null argument");
  out[0] = s->sweeps_run;
  out[1] = s->exchanges;
  out[2] = (int64_t)s->graphs.size();
  out[3] = s->exchange ? 1 : 0;
  return GLX_OK;
}


// what the object decided: out[0] exchanging, [1] exchanging sweeps captured (1) / eager (0) / undecided (-1), [2] self-test
// (0 not run, 1 passed, 2 failed), [3] exchange beside the interior rows on a second stream, [4] one launch per sweep,
// [5] boundary rows scattered into the send buffer by the SpMM (no pack kernel), [6] records sent per sweep, [7] halo records
extern "C" int glx_dist_sweep_info(const glx_dist_sweep* s, int64_t out[8]) {
  GLX_CHECK(s && out, GLX_EINVAL, "glx_dist_sweep_info: null argument");
  out[0] = s->exchange ? 1 : 0;
  out[1] = s->capture_exchange;
  out[2] = s->selftest;
  out[3] = s->overlap ? 1 : 0;
  out[4] = s->fused ? 1 : 0;
  out[5] = s->scatter ? 1 : 0;
  out[6] = s->gather ? s->cap : s->n_send;
  out[7] = s->n_halo;
  if (s->gather) out[5] = 2;      // (2: no send buffer at all -- the all-gather reads the rank's own block)
  return GLX_OK;
}

// Device time of the rank-local pieces of one sweep, each timed alone over `reps` launches (HIP events on the sweep's stream,
// ping-pong between two state buffers, no exchange): us_out[0] boundary rows (incl. the scatter into the send buffer),
// [1] interior rows, [2] the pack kernel (the round-2 form of [0]'s scatter), [3] boundary + interior back to back.
// What scripts/scale_model.py builds the predicted scaling curve from.
extern "C" int glx_dist_sweep_time_parts(glx_dist_sweep* s, int reps, float us_out[4]) {
  GLX_CHECK(s && us_out && reps >= 1, GLX_EINVAL, "glx_dist_sweep_time_parts: bad argument");
  GLX_CHECK(s->problem_set, GLX_EINVAL, "glx_dist_sweep_time_parts: set the problem first");
  GLX_HIP(hipSetDevice(s->device));
  auto timed = [&](int which, float* out) -> int {
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 warms up
      GLX_HIP(hipEventRecord(s->ev0, s->stream));
      for (int j = 0; j < reps; ++j) {
        const void* xin = s->ring[j & 1];
        void* xout = s->ring[(j & 1) ^ 1];
        int rc = GLX_OK;
        if (which == 0 || which == 3) rc = launch_part(s, 0, xin, xout, nullptr);
        if (!rc && (which == 1 || which == 3)) rc = launch_part(s, 1, xin, xout, nullptr);
        if (!rc && which == 2) rc = enqueue_pack(s, xout);
        if (rc) return rc;
      }
      GLX_HIP(hipEventRecord(s->ev1, s->stream));
      GLX_HIP(hipStreamSynchronize(s->stream));
    }
    float ms = 0.f;
    GLX_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *out = ms * 1e3f / (float)reps;
    return GLX_OK;
  };
  for (int w = 0; w < 4; ++w) {
    int rc = timed(w, &us_out[w]);
    if (rc) return rc;
  }
  return GLX_OK;
}

// ---- stepwise form: the same pieces with the transport left to the caller -----------------------------------
// (multi-rank tests on ONE GPU exchange through a host-side backend: boundary + pack, caller moves the packed
// records to the peers' halos, interior.)  Eager, synchronous.
extern "C" int glx_dist_sweep_begin(glx_dist_sweep* s) {
  GLX_CHECK(s && s->problem_set, GLX_EINVAL, "glx_dist_sweep_begin: set the problem first");
  GLX_HIP(hipSetDevice(s->device));
  GLX_HIP(hipMemcpyAsync(rec_at(s->ring[0], s, s->own_off), s->init_rec, (size_t)s->n_own * s->L.ld * s->L.esize, hipMemcpyDeviceToDevice, s->stream));
  s->cur = 0;
  s->gather_next = false;
  int rc = s->gather ? GLX_OK : enqueue_pack(s, s->ring[0]);
  if (rc) return rc;
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// boundary rows of the next iterate + pack of its boundary records
extern "C" int glx_dist_sweep_boundary(glx_dist_sweep* s, int want_err) {
  GLX_CHECK(s && s->problem_set, GLX_EINVAL, "glx_dist_sweep_boundary: set the problem first");
  GLX_HIP(hipSetDevice(s->device));
  if (want_err) GLX_HIP(hipMemsetAsync(s->err, 0, ERR_SHARDS * 8, s->stream));
  void* xout = s->ring[s->cur ^ 1];
  int rc = launch_part(s, 0, s->ring[s->cur], xout, want_err ? s->err : nullptr);
  if (rc) return rc;
  if (!s->scatter && !s->gather) rc = enqueue_pack(s, xout);           // (scatter: the SpMM has filled the send buffer)
  s->gather_next = true;                                               // (gather form: get_send hands out the block just written)
  if (rc) return rc;
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

extern "C" int glx_dist_sweep_get_send(glx_dist_sweep* s, void* host_out) {
  GLX_CHECK(s && (host_out || s->n_send == 0), GLX_EINVAL, "glx_dist_sweep_get_send: null argument");
  GLX_HIP(hipSetDevice(s->device));
  if (s->gather) {       // the rank's block (cap records) of the iterate being exchanged: the newest one written
    GLX_CHECK(host_out, GLX_EINVAL, "glx_dist_sweep_get_send: null argument");
    GLX_UP(glx_download_sync(host_out, rec_at(s->ring[s->gather_next ? (s->cur ^ 1) : s->cur], s, s->own_off), (size_t)s->cap * s->L.ld * s->L.esize, __func__));
    return GLX_OK;
  }
  if (s->n_send > 0) GLX_UP(glx_download_sync(host_out, s->sendbuf, (size_t)s->n_send * s->L.ld * s->L.esize, __func__));
  return GLX_OK;
}

// records received from the peers -> halo region of the current iterate (next = 0) or of the one being computed (next = 1)
extern "C" int glx_dist_sweep_put_halo(glx_dist_sweep* s, const void* host_in, int next) {
  GLX_CHECK(s && (host_in || s->n_halo == 0), GLX_EINVAL, "glx_dist_sweep_put_halo: null argument");
  GLX_HIP(hipSetDevice(s->device));
  void* x = s->ring[next ? (s->cur ^ 1) : s->cur];
  if (s->gather) {       // host_in: the other ranks' blocks in rank order, cap records each
    const size_t bb = (size_t)s->cap * s->L.ld * s->L.esize;
    int k = 0;
    for (int r = 0; r < s->comm->nranks; ++r) {
      if (r == s->comm->rank) continue;
      GLX_UP(glx_upload_sync(rec_at(x, s, (int64_t)r * s->cap), (const char*)host_in + (size_t)k * bb, bb, __func__));
      ++k;
    }
    s->gather_next = false;
    return GLX_OK;
  }
  if (s->n_halo > 0) GLX_UP(glx_upload_sync(rec_at(x, s, s->n_own), host_in, (size_t)s->n_halo * s->L.ld * s->L.esize, __func__));
  return GLX_OK;
}

// interior rows, then the new iterate becomes the current one; err_local_out: rank-local max |deg w - vinf| (if asked)
extern "C" int glx_dist_sweep_interior(glx_dist_sweep* s, int want_err, double* err_local_out) {
  GLX_CHECK(s && s->problem_set, GLX_EINVAL, "glx_dist_sweep_interior: set the problem first");
  GLX_HIP(hipSetDevice(s->device));
  int rc = launch_part(s, 1, s->ring[s->cur], s->ring[s->cur ^ 1], want_err ? s->err : nullptr);
  if (rc) return rc;
  if (want_err) {
    GLX_HIP(hipMemcpyAsync(s->h_err, s->err, ERR_SHARDS * 8, hipMemcpyDeviceToHost, s->stream));
    GLX_HIP(hipStreamSynchronize(s->stream));
    if (err_local_out) *err_local_out = shard_max(s->h_err, 0);
  } else {
    GLX_HIP(hipStreamSynchronize(s->stream));
  }
  s->cur ^= 1;
  s->gather_next = false;
  s->sweeps_run++;
  return GLX_OK;
}
