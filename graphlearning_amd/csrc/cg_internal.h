// Shared by cg.hip (reference-order conjugate gradient) and cg_fused.hip (tolerance mode).
#pragma once
#include "glx_internal.h"
#include <map>
#include <vector>
#include <algorithm>

struct PwPlan {          // device arrays; value index space: leaves [0, nleaves), internal nodes after them by height
  const int64_t* leaf_off;
  const int32_t* leaf_len;
  const int32_t* node_l;     // [ninternal] children of internal node q (value indices)
  const int32_t* node_r;
  const int32_t* level_start;   // [nlevels + 1] ranges of internal nodes (0-based among internals) per height
  double* vals;              // [nleaves + ninternal]
  int nleaves, ninternal, nlevels;
};


// Work buffers of a solve.  They live with the operator (glx_graph::cg_ws) and are reused by later solves on it:
// a dozen hipMalloc / hipFree pairs per call cost milliseconds -- as much as a whole tolerance-mode solve at 60k.
struct CgBufs {
  void *x = nullptr, *r = nullptr, *p = nullptr, *ap = nullptr, *dense = nullptr;
  double* prod = nullptr;
  int64_t* pw_off = nullptr;
  int32_t *pw_len = nullptr, *pw_l = nullptr, *pw_r = nullptr, *pw_ls = nullptr;
  double* pw_vals = nullptr;
  int32_t *mask_rows = nullptr, *mask_ptr = nullptr, *rhs_rows = nullptr;
  double* out_scale = nullptr;
  double *part_dot = nullptr, *part_rs = nullptr, *scal = nullptr, *err_hist = nullptr, *h_err = nullptr;
  // tolerance mode (cg_fused.hip): partial sums, counters, Dirichlet-row masks, staging, the captured launch sequences
  double *f_part1 = nullptr, *f_part1g = nullptr, *f_part2 = nullptr;
  unsigned *f_tick = nullptr, *f_rowmask = nullptr;
  int* f_it = nullptr;
  char *f_stage = nullptr, *h_stage = nullptr;          // one upload per solve: rows, Dirichlet rows, values, output scale
  double* h_hist = nullptr;                              // page-locked mirror of err_hist, written by the closing kernels, polled by the host
  int64_t h_hist_rows = 0, h_hist_dirty = 0;             // doubles laid out / rows a solve may have written (reset to 'not yet' before the next)
  double last_stop_margin = INFINITY;                    // of the last tolerance-mode solve (glx_cg_last_stop_margin)
  int h_hist_stride = 0;                                 // the row stride (systems + 1) the mirror's 'not yet' markers were last laid out for
  hipGraphExec_t f_exec[3] = {nullptr, nullptr, nullptr};        // captured chunks of 32, 16 and 4 iterations
  std::vector<unsigned long long> f_key;
  hipEvent_t f_ev[3] = {nullptr, nullptr, nullptr};
  hipStream_t stream = nullptr, side = nullptr;
  std::map<void**, size_t> cap;
  int64_t pw_n = -1;   // rows the pairwise-summation plan was built for
  PwPlan pw;
  unsigned pw_grid = 1;
  // device buffer of at least `bytes` (contents undefined after growth)
  int need(void** ptr, size_t bytes) {
    bytes = std::max<size_t>(bytes, 64);
    auto it = cap.find(ptr);
    if (it != cap.end() && it->second >= bytes && *ptr) return GLX_OK;
    hipFree(*ptr);
    *ptr = nullptr;
    cap[ptr] = 0;
    GLX_HIP(hipMalloc(ptr, bytes));
    cap[ptr] = bytes;
    return GLX_OK;
  }
  template <class P> int need_host(P** ptr, size_t bytes) {
    auto it = cap.find((void**)ptr);
    if (it != cap.end() && it->second >= bytes && *ptr) return GLX_OK;
    if (*ptr) hipHostFree(*ptr);
    *ptr = nullptr;
    cap[(void**)ptr] = 0;
    GLX_HIP(hipHostMalloc((void**)ptr, bytes, hipHostMallocDefault));
    cap[(void**)ptr] = bytes;
    return GLX_OK;
  }
  ~CgBufs() {
    hipFree(x); hipFree(r); hipFree(p); hipFree(ap); hipFree(dense); hipFree(part_dot); hipFree(part_rs);
    hipFree(scal); hipFree(err_hist); hipFree(prod);
    hipFree(pw_off); hipFree(pw_len); hipFree(pw_l); hipFree(pw_r); hipFree(pw_ls); hipFree(pw_vals); hipFree(mask_rows); hipFree(mask_ptr);
    hipFree(rhs_rows); hipFree(out_scale);
    hipFree(f_part1); hipFree(f_part1g); hipFree(f_part2); hipFree(f_tick); hipFree(f_rowmask); hipFree(f_it);
    hipFree(f_stage);
    if (h_stage) hipHostFree(h_stage);
    if (h_hist) hipHostFree(h_hist);
    for (int q = 0; q < 3; ++q) if (f_exec[q]) hipGraphExecDestroy(f_exec[q]);
    for (int q = 0; q < 3; ++q) if (f_ev[q]) hipEventDestroy(f_ev[q]);
    if (side) hipStreamDestroy(side);
    if (h_err) hipHostFree(h_err);
    if (stream) hipStreamDestroy(stream);
  }
};


struct CgRhsRows {            // optional forms of the right-hand side and of the result (glx_cg_groups_rows)
  int64_t nb = 0;
  const int32_t* rows = nullptr;
  const void* vals = nullptr;
  const double* out_scale = nullptr;
};


// the tolerance-mode solve (cg_fused.hip); arguments as cg.hip's cg_run
int glx_cg_run_fused(glx_graph* A, const void* B, void* X, int C, int Cg, double tol, int64_t max_iter, int* iters_out, double* err_out,
                     int flags, const int32_t* mask_rows, const int32_t* mask_ptr, const CgRhsRows& rr);
// records -> dense (n, C) in the caller's row order, every row times scale[row] (device pointers)
int glx_cg_unpack_scaled(int dtype, const void* rec, void* dense, int64_t n, const RecLayout& L, const int32_t* perm, const double* scale,
                         hipStream_t st);
#define CG_NEED(ptr, bytes) do { int rc_ = b.need((void**)&(ptr), (bytes)); if (rc_) return rc_; } while (0)
