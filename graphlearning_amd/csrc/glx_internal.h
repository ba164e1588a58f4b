// Internal declarations shared by the libglx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/glx.h"

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
};
int glx_make_layout(int C, int dtype, bool has_w, RecLayout* L);

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
  void* cg_ws = nullptr;   // work buffers of the conjugate-gradient solves on this operator (cg.hip), reused between calls
  std::mutex solve_mu;     // one solve at a time per operator: the work buffers, their stream and the plans are shared (ctypes
                           // releases the GIL, so two Python threads can reach the same operator)
};
void glx_cg_ws_destroy(void* ws);

// device work-buffer pool (graph.hip): size-class free lists in front of hipMalloc / hipFree
int glx_pool_alloc(void** out, size_t bytes);
void glx_pool_free(void* p);
// the neighbour indices the last full search left on the device after glx_knn_retain_next(1) ([n][k] int64, a pooled block):
// handed over to the caller (who frees it with glx_pool_free), or GLX_EINVAL when nothing of that shape is retained (knn.hip)
int glx_knn_take_retained(int64_t n, int k, int device, int64_t** ind_dev);

// a non-blocking stream and four events, handed out from a per-device list of idle sets and returned to it (graph.hip):
// creating and destroying them costs milliseconds -- as much as the kNN search of 70 000 points itself
struct glx_work {
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;     // a second stream for work beside the main one (the cell-order by-product of the kNN search)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_side = nullptr;
  int device = 0;
};
int glx_work_acquire(int device, glx_work** out);
void glx_work_release(glx_work* w);

int glx_graph_plan(glx_graph* g, int G, SellPlan** out);
int glx_graph_ensure_order(glx_graph* g);

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
                       const long long** d_labels_out);
int glx_onehot_device(const long long* d_labels, void* dense_dev, int dtype, int64_t n, int C, hipStream_t st);
void glx_projector_destroy(glx_projector* p);

