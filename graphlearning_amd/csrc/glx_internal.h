// Internal declarations shared by the libglx translation units (gfx950 only).
#pragma once
#include <functional>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/glx_experimental.h"   // (includes glx.h)

// wavefronts (= slices) per workgroup of the SpMM kernel
#ifndef GLX_WPB
#define GLX_WPB 4
#endif

void glx_set_error(const char* fmt, ...);

#define GLX_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      glx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));  \
      return GLX_EHIP;                                                                    \
    }                                                                                     \
  } while (0)

#define GLX_CHECK(cond, code, ...)   \
  do {                               \
    if (!(cond)) {                   \
      glx_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

// Record layout of one vertex in a dense operand: `ld` elements of T per vertex,
// columns 0..C-1 first, zero padding up to nvec*4, then (optionally) one fp64 stop
// value at byte offset woff.  Rows are 32/64/128-byte aligned so one neighbour
// gather touches exactly one cache line for C <= 12 (fp64) / 28 (fp32).
struct RecLayout {
  int C;      // real columns
  int nvec;   // column vectors of 4 elements
  int ld;     // elements of T per vertex record
  int woff;   // byte offset of the fp64 stop value inside the record (-1: none)
  int G;      // lanes cooperating on one row (power of two, >= nvec + has_w)
  int esize;  // sizeof(T)
  int ngroups = 1;  // column groups (stacked training sets, groups.hip): C = ngroups * (columns per group)
  int nstop = 0;    // lanes behind the column vectors that hold fp64 stop values (one value per group, group b at byte woff + 8 b)
};
int glx_make_layout(int C, int dtype, bool has_w, RecLayout* L);
// `B` groups of `C` columns with one fp64 stop value each (B = 1: the layout of glx_make_layout(C, dtype, true))
int glx_make_layout_groups(int C, int B, int dtype, RecLayout* L);
// stop-test maxima of a grouped sweep: per iteration [ngroups][GLX_GRP_SHARDS] values (sharded atomic maxima, sweep.hip)
#define GLX_GRP_SHARDS 16

// Sliced-ELL view of a sparse operator.  A slice is the R = 64/G rows one wavefront
// processes; entries are stored in chunks of 64: chunk k of a slice holds, for row
// slot g and t in [0,G), the entry jj = k*G + t of that row at index g*G + t, so a
// wavefront fetches a whole chunk with one coalesced load and hands entry t to the
// G lanes of a row with a DPP quad broadcast.  In G = 4 plans a long row occupies S = 4 or
// 16 consecutive slots (its entries dealt to them 4 at a time, in order), see sweep.hip.
struct SliceHdr {
  int64_t ptr;      // entry offset (within the tail region) of the slice's chunk 1 (multiple of 64)
  int32_t nchunks;  // 64-entry chunks
  int32_t S;        // bits 0..7: segments per row: 1, or 4 / 16 for long rows (G = 4 plans); bits 8..: chunks without a short slot ("full")
};

struct SellPlan {
  int G = 0, R = 0;
  bool relaxed = false;        // built for kernels that add a row's entries in any fixed order (tolerance-mode CG): see glx_graph_plan
  int64_t nslices = 0;
  int64_t stored = 0;          // stored entries incl. padding
  int64_t head = 0;            // first `head` entries: chunk 0 of every slice (slice s at s*64), addressable
                               // without the slice header; chunks 1.. follow at head + hdr.ptr
  int32_t* d_slot_row = nullptr;   // [nslices*R] row id or -1
  int32_t* d_slot_len = nullptr;   // [nslices*R]
  SliceHdr* d_slice_hdr = nullptr; // [nslices]
  int32_t* d_col = nullptr;        // [stored]
  void* d_val = nullptr;           // [stored] of state dtype
};

struct glx_graph {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int dtype = GLX_F64;
  int device = 0;
  int max_row = 0;
  // host copy of the CSR (entry order preserved), used to build plans lazily
  std::vector<int32_t> h_rowptr, h_col;
  std::vector<double> h_val;
  std::vector<SellPlan> plans;   // one per G in use
  // locality renumbering (square operators): perm[new] = old, inv[old] = new; empty = identity
  bool order_ready = false, keep_order = false;
  std::vector<int32_t> h_perm, h_inv;
  int32_t* d_perm = nullptr;
  int32_t* d_inv = nullptr;
  // resident source (glx_graph_create_resident): the CSR arrays live on the DEVICE -- the host keeps the row pointers only (h_col /
  // h_val stay empty unless something asks for the pattern) -- and the operator's row i is row i of the source, optionally with
  // its entries in reverse order and scaled by row_scale[i] (glx_graph_set_row_transform: P = D^-1 W^T of a symmetric W)
  int32_t* d_src_rowptr = nullptr;
  int32_t* d_src_col = nullptr;
  double* d_src_val = nullptr;
  double* d_row_scale = nullptr;
  bool reverse_rows = false;
  void* cg_ws = nullptr;   // work buffers of the conjugate-gradient solves on this operator (cg.hip), reused between calls
  std::mutex solve_mu;     // one solve at a time per operator: the work buffers, their stream and the plans are shared (ctypes
                           // releases the GIL, so two Python threads can reach the same operator)
};
void glx_cg_ws_destroy(void* ws);

// Zero-fill by a KERNEL, for launch sequences that are captured and replayed: a memset NODE of a captured sequence is not safe on every
// runtime this library meets -- on the HIP runtime bundled with PyTorch 2.10 + rocm7.0 (7.0.51831: what the library runs on whenever torch was
// imported first, i.e. in every multi-GPU job) a replayed memset node fills with the value of the process's last eager hipMemset instead
// of its own (scripts/probes/graph_memset_probe.hip; found in round 6 when a poisoned pool turned a row of stop values into 1.4e306 on
// the second replay of the stacked trials' head graph).  `p` 8-byte aligned, `bytes` a multiple of 8.
int glx_zero_async(void* p, size_t bytes, hipStream_t st);
// device work-buffer pool (graph.hip): size-class free lists in front of hipMalloc / hipFree
int glx_pool_alloc(void** out, size_t bytes);
void glx_pool_free(void* p);
// small page-locked blocks (graph.hip): power-of-two classes in front of hipHostMalloc / hipHostFree (0.25 ms each)
int glx_pinned_alloc(void** out, size_t bytes);
void glx_pinned_free(void* p);

// a finished full search: its lists on the device (pooled blocks, [n][k]) and the cell order it worked out (empty: none)
struct glx_knn_result {
  int64_t n = 0;
  int k = 0, device = 0;
  int64_t* ind = nullptr;
  double* dist = nullptr;
  std::vector<int32_t> order;       // the cell order worked out on the host (cell-pruned search) ...
  int32_t* order_dev = nullptr;     // ... or left on the device (rows reordered by the cellrank kernels): a pooled block of n entries
};

// a non-blocking stream and four events, handed out from a per-device list of idle sets and returned to it (graph.hip):
// creating and destroying them costs milliseconds -- as much as the kNN search of 70 000 points itself
struct glx_work {
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;     // a second stream for work beside the main one (the cell-order by-product of the kNN search)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_side = nullptr;
  int device = 0;
  void* stage = nullptr;          // page-locked host staging area of the set (glx_work_stage), kept with it
  size_t stage_bytes = 0;
};
int glx_work_acquire(int device, glx_work** out);
void glx_work_release(glx_work* w);
// `bytes` of page-locked host memory that stays with the work set (grown on demand, contents not preserved across a growth): where
// device-to-host copies of counters and lists land that the host then reads.  A copy into FRESH pageable memory makes the runtime
// pin the destination on the fly -- 8 ms for a 280 KB std::vector the first time a graph of a new size is built.
int glx_work_stage(glx_work* w, size_t bytes, void** out);
// ---- host -> device uploads ------------------------------------------------------------------------------------------------------------
// Round 6 found the one parity failure the randomised soak ever produced: a copy between PAGEABLE host memory and the device -- hipMemcpyAsync
// from a numpy array, the runtime staging it internally -- that arrived with a run of 256 zero bytes 128-132 KB into the transfer, about once
// in 10 000 copies and only while a dozen processes shared the GPU (EXPERIMENTS.md round 6, section 2; scripts/replay_crumbs.sh reproduces
// it).  Copies from and to page-locked memory never showed it.  So every upload of 128 KB or more goes through page-locked staging memory of
// the library's own (two halves that take turns, filled by host threads), and is CHECKED: the host threads add the 64-bit words up while
// they copy them in, a kernel adds up what arrived, the sums are compared before the call returns (one stream synchronisation).  A
// difference is described on stderr, counted (glx_upload_stats) and the upload repeated through another part of the staging area, up to
// three times; GLX_EHIP if none arrives intact.  Below 128 KB: hipMemcpyAsync as before (no faulty run ever sat that early in a transfer).
// The staging area belongs to the calling thread and device; `st` may be any stream (the null stream included).
int glx_upload(void* dst, const void* src, size_t bytes, hipStream_t st, const char* what);
int glx_upload_sync(void* dst, const void* src, size_t bytes, const char* what);      // ... on the null stream, complete on return
// The other direction, for the same reason: a result of 128 KB or more that goes into memory which is NOT page-locked (a caller's plain numpy
// array) comes down through the staging area piece by piece and is checked the same way (a kernel's sum of the device words against the sum
// of what the host threads copied out); the call then returns with the copy COMPLETE.  Into page-locked memory (what the Python layer hands
// in for its results) and below 128 KB: hipMemcpyAsync as before, complete when the caller synchronises `st`.
int glx_download(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st, const char* what);
int glx_download_sync(void* dst_host, const void* src_dev, size_t bytes, const char* what);
#define GLX_UP(call) do { const int rc_up_ = (call); if (rc_up_) return rc_up_; } while (0)

int glx_graph_plan(glx_graph* g, int G, SellPlan** out, bool relaxed = false);
int glx_graph_ensure_order(glx_graph* g);

// ---- tolerance-mode CG (cg_fused.hip): two launches per iteration ----------------------------------------------------
// Device-resident scalars, counters and partial sums of one solve.  The iteration number lives on the device (it_a read by the
// SpMM kernel, it_b by the update kernel: neither is written by a kernel whose own late workgroups still read it), so one
// captured launch sequence serves every iteration.  Who reduces what (every sum in a fixed order, no float atomics):
//   update kernel (it-1)  writes part2[workgroup][ncols] = partial r.r;
//   SpMM kernel (it)      one extra workgroup closes iteration it-1 BESIDE the product: rsold = sum part2, err_hist[it-1];
//                         the others write part1[workgroup][3 ncols] (p.Ap, r.Ap, Ap.Ap); the last arriver of every group of
//                         `grp` workgroups adds its group's rows into part1g[group];
//   update kernel (it)    every workgroup adds the (few) rows of part1g and forms alpha, beta for itself.
// Nothing but the last group's sum sits between the end of a kernel's real work and the next launch.
struct CgDev {
  double* rsold;      // [ncols] r.r of the current residual
  double* err_hist;   // [max_iter + 2 + slack][stride]: per system, then the maximum over the systems still running; row 0 = 1 (utils.py:519)
  int stride, ngroups, Cg, C;
  int max_iter;
  int* it_a;          // iteration about to run (SpMM kernel reads, update kernel writes)
  int* it_b;          // the same, handed from the SpMM kernel to the update kernel
  int* closed;        // highest iteration whose err_hist row is complete
  const char* r;      // residual records (the SpMM kernel forms r.Ap beside p.Ap and Ap.Ap)
  double* part1;      // [nblocks][3 ncols]
  double* part1g;     // [ngrp][3 ncols]
  unsigned* tick1;    // [ngrp] arrival counters (back at zero after every launch that ran)
  int grp, ngrp;      // workgroups per group, groups
  double* part2;      // [nb2][ncols]
  int nb2;
  double* host_hist;  // page-locked mirror of err_hist the host polls (null: none)
};

// Cross-workgroup hand-off inside one launch (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed
// by another CU's stores): the payload leaves with agent-scope (write-through) stores, the writer drains them before it draws its
// ticket, the reader uses agent-scope loads -- the "8-byte agent atomics on both sides" form of the hardware guide.
__device__ __forceinline__ void glx_agent_store(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool AGENT> __device__ __forceinline__ double glx_load_part(const double* p) {
  if constexpr (AGENT)
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else
    return *p;
}
// every thread of the workgroup calls it after its agent-scope stores; true (in all threads) for the workgroup that arrives last
// of `count`.  The counter is back at zero when it returns true.  `s_flag`: any LDS word the caller can spare around the call.
__device__ __forceinline__ bool glx_arrive_last(unsigned* ticket, unsigned count, double* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == count - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *(volatile unsigned*)s_flag = last ? 1u : 0u;
  }
  __syncthreads();
  const bool last = *(volatile unsigned*)s_flag != 0;
  __syncthreads();   // s_flag is free again
  return last;
}
// column sums of part[nrows][nq] in a fixed order: the workgroup's threads split the rows into parts, eight independent
// accumulators per thread keep eight loads in flight, the parts are added in order.  consume(q, total) is called by exactly one
// thread per column.  s_tmp: blockDim.x doubles of LDS.  AGENT: the rows were written by other workgroups of THIS launch
// (glx_agent_store); plain loads serve rows a previous launch wrote.
template <bool AGENT, class F>
__device__ __forceinline__ void glx_reduce_rows(const double* part, int64_t nrows, int nq, double* s_tmp, F&& consume) {
#pragma clang fp contract(off)
  const int nt = (int)blockDim.x;
  for (int q0 = 0; q0 < nq; q0 += nt) {
    const int nqc = nq - q0 < nt ? nq - q0 : nt;
    const int parts = nt / nqc;
    const int q = (int)threadIdx.x % nqc, part_id = (int)threadIdx.x / nqc;
    double s = 0.0;
    if (part_id < parts) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int64_t r = part_id;
      for (; r + 7 * (int64_t)parts < nrows; r += 8 * (int64_t)parts) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += glx_load_part<AGENT>(part + (size_t)(r + j * (int64_t)parts) * nq + q0 + q);
      }
      for (int j = 0; r < nrows; r += parts, ++j) a[j] += glx_load_part<AGENT>(part + (size_t)r * nq + q0 + q);
      s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    s_tmp[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < nqc) {
      double tot = s_tmp[threadIdx.x];
      for (int pp = 1; pp < parts; ++pp) tot += s_tmp[pp * nqc + threadIdx.x];
      consume(q0 + (int)threadIdx.x, tot);
    }
    __syncthreads();
  }
}

// Close iteration `j = it - 1` (one workgroup of 256 threads; idempotent): rsold = r.r summed over part2 for the systems that ran it
// (utils.py:527, 530), err_j = np.sqrt(np.sum(rsnew)) per system (utils.py:528; np.sum over a contiguous 1-D float64 array is
// numpy's pairwise_sum: 8 strided accumulators below 128 elements, a plain loop below 8), then the maximum over the systems still
// running.  A NaN error stops its own system (`nan > tol` is false) and must not keep the others alive.
__device__ __forceinline__ void glx_cg_close_iteration(const CgDev& cg, int it, double tol, double* s_tmp /* [256] */) {
#pragma clang fp contract(off)
  const int j = it - 1;
  if (j < 0 || *cg.closed >= j) return;
  // system g ran iteration j iff the row before says so; "iteration 0" is the set-up: rsold = r.r of the right-hand side (utils.py:517)
  const double* ran = j >= 1 ? cg.err_hist + (size_t)(j - 1) * cg.stride : nullptr;
  const int ncols = ((cg.C + 3) / 4) * 4;
  glx_reduce_rows<false>(cg.part2, (int64_t)cg.nb2, ncols, s_tmp, [&](int q, double tot) {
    if (!ran || q >= cg.C || ran[q / cg.Cg] > tol) cg.rsold[q] = tot;
  });
  __threadfence_block();
  __syncthreads();
  if (j >= 1) {
    double mine = 0.0;
    const int g = threadIdx.x;
    double* row = cg.err_hist + (size_t)j * cg.stride;
    // the host's copy of the row (page-locked, mapped): per-system values first, the row's maximum LAST -- the host polls that slot
    // (it holds NaN until then; the maximum itself is never NaN) and needs no event or copy in the solve's stream
    double* hrow = cg.host_hist ? cg.host_hist + (size_t)j * cg.stride : nullptr;
    if (g < cg.ngroups) {
      if (ran[g] > tol) {
        const double* v = cg.rsold + (size_t)g * cg.Cg;
        const int C = cg.Cg;
        double e;
        if (C < 8) {
          e = 0.0;
          for (int q = 0; q < C; ++q) e = e + v[q];
        } else {
          double r8[8];
          for (int q = 0; q < 8; ++q) r8[q] = v[q];
          int i = 8;
          for (; i < C - (C % 8); i += 8)
            for (int q = 0; q < 8; ++q) r8[q] = r8[q] + v[i + q];
          e = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
          for (; i < C; ++i) e = e + v[i];
        }
        mine = sqrt(e);
      }
      row[g] = mine;        // 0 = stopped: every row of the history is written in full, so nothing needs clearing between solves
      if (hrow) __hip_atomic_store((unsigned long long*)&hrow[g], (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_tmp[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      double m = 0.0;
      for (int q = 0; q < cg.ngroups && q < 256; ++q)
        if (s_tmp[q] > m) m = s_tmp[q];
      row[cg.ngroups] = m;
      if (hrow) {
        __threadfence_system();        // (the per-system values above were stored before the barrier in front of this block)
        __hip_atomic_store((unsigned long long*)&hrow[cg.ngroups], (unsigned long long)__double_as_longlong(m), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (threadIdx.x == 0) {
    __threadfence();
    *cg.closed = j;
  }
  __syncthreads();
}

struct RecLayout;
int glx_cg_fused_update_blocks(int64_t n, int* rows_per_block);
int glx_cg_fused_update(int dtype, void* x, void* r, void* p, const void* ap, int64_t n, const RecLayout& L, const CgDev& cg,
                        double tol, hipStream_t st);
int glx_cg_fused_close(const CgDev& cg, double tol, hipStream_t st);

// kernels' launch wrappers (sweep.hip)
struct SweepArgs {
  const SellPlan* plan;
  RecLayout L;
  int dtype;
  const void* xin;
  void* xout;
  const void* bias;              // record layout or nullptr
  const uint8_t* slot_has_bias;  // per slot flag (nullptr -> dense bias if bias != nullptr)
  // stop column
  bool has_w;
  const double* deg;
  const double* vinf;
  unsigned long long* err_prev;  // [64] shards read at kernel start (nullptr: no early exit)
  unsigned long long* err_next;  // [64] shards written (nullptr: do not compute)
  double thresh;
  // fused column dots (CG): dot_out[block][C] partial sums of xin[row,c]*xout[row,c]
  double* dot_partial;
  const double* exit_err;        // optional early exit: skip when !(*exit_err > exit_tol)
  double exit_tol;
  double* prod_out;              // optional: elementwise xin*xout, row-major (nvec*4 columns), caller row order (CG)
  const double* act_row;         // optional (CG column groups): group g still runs iff act_row[g] > exit_tol
  int act_cg, act_c;             // columns per group / total columns
  int prod_sc;                   // prod_out column-block width
  const int32_t* perm;
  int64_t n_rows;
  // optional (vertex-partitioned sweep, boundary rows): row r is also stored at records dup_pos[dup_ptr[r] .. dup_ptr[r+1])
  // of dup_out (the halo exchange's send buffer)
  const int32_t* dup_ptr;
  const int32_t* dup_pos;
  void* dup_out;
  // CG: Dirichlet rows (bit g of rowmask[record]: A p held at zero there for system g); tolerance-mode state (cg_fused.hip)
  const unsigned* rowmask;
  const CgDev* cg;
  // column groups (stacked training sets): ngroups > 1 selects the grouped kernel; err_prev / err_next are then [ngroups][GLX_GRP_SHARDS]
  int ngroups;
  unsigned used_mask;
};
int glx_launch_spmm(const SweepArgs& a, hipStream_t stream);
int64_t glx_spmm_blocks(const SellPlan* plan);

// perm (device, may be null = identity): record `i` holds vertex perm[i] of the caller's arrays
int glx_pack_records(const void* dense, void* rec, int64_t n, const RecLayout& L, int dtype, const double* w, hipStream_t s,
                     const int32_t* perm = nullptr);
int glx_unpack_records(const void* rec, void* dense, int64_t n, const RecLayout& L, int dtype, hipStream_t s,
                       const int32_t* perm = nullptr);

// project.hip: ssl.predict / volume_label_projection on a device-resident (n, C) array of the state dtype
struct glx_projector;
int glx_project_device(glx_projector** pp, const void* dense_dev, int dtype, int64_t n, int C, const double* priors,
                       double* weights_inout, double* err_out, int* steps_out, int max_steps, int similarity, hipStream_t st,
                       const long long** d_labels_out, const std::function<int(bool)>* hook = nullptr);
int glx_onehot_device(const long long* d_labels, void* dense_dev, int dtype, int64_t n, int C, hipStream_t st);
int glx_project_scores(glx_projector** pp, int64_t n, int C, double** scores_out);
int glx_onehot_records(const long long* d_labels, void* rec, int dtype, int64_t n, const RecLayout& L, const int32_t* perm, hipStream_t st);
void glx_projector_destroy(glx_projector* p);

