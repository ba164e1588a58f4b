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
  // block form of the reference-order reductions (cg_seqsum.hip): block / chunk sums, block records, row-by-row flags, counters
  double *ss_bsum = nullptr, *ss_csum = nullptr;
  char* ss_rec = nullptr;
  unsigned long long* ss_mask = nullptr;
  unsigned long long* ss_stats = nullptr;   // [3 modes][4]: blocks taken plain / by record / row by row, per kind of reduction (64-bit: 1.6e7 per iteration at the largest sizes)
  unsigned long long* h_ss = nullptr;       // page-locked copies of them: one per chunk in flight
  int ss_last[4] = {-1, -1, -1, -1};   // of the last reference-order solve: blocks taken plain / by record / row by row, kinds of reduction
                                       // still in block form at its end (bit 0: p.Ap, bit 1: r.r); -1: chain form
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
  // reference-order mode (cg.hip): a chunk of CG_CHUNK iterations as ONE captured launch sequence per combination of reduction forms
  // (bit 0: p.Ap in block form, bit 1: r.r), valid for the solve parameters in e_key; events of the two chunks in flight
  hipGraphExec_t e_exec[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // two instances each: the chunks in flight never share one
  std::vector<unsigned long long> e_key;
  hipEvent_t e_ev[2] = {nullptr, nullptr};
  unsigned* e_rowmask = nullptr;                         // Dirichlet rows of every system as bits (the SpMM holds A p at zero there)
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
    hipFree(ss_bsum); hipFree(ss_csum); hipFree(ss_rec); hipFree(ss_mask); hipFree(ss_stats); hipFree(e_rowmask);
    for (int q = 0; q < 8; ++q) if (e_exec[q / 2][q % 2]) hipGraphExecDestroy(e_exec[q / 2][q % 2]);
    for (int q = 0; q < 2; ++q) if (e_ev[q]) hipEventDestroy(e_ev[q]);
    if (h_stage) hipHostFree(h_stage);
    if (h_hist) hipHostFree(h_hist);
    for (int q = 0; q < 3; ++q) if (f_exec[q]) hipGraphExecDestroy(f_exec[q]);
    for (int q = 0; q < 3; ++q) if (f_ev[q]) hipEventDestroy(f_ev[q]);
    if (side) hipStreamDestroy(side);
    if (h_err) hipHostFree(h_err);
    if (h_ss) hipHostFree(h_ss);
    if (stream) hipStreamDestroy(stream);
  }
};


// The C columns may hold several independent systems side by side ("groups" of Cg columns: the
// trials of ssl.ssl_trials, ssl.py:292-396, stacked on one operator).  Every group has its own
// residual norm, stop test and iteration count, exactly as if it had been solved alone; a group
// that has converged is frozen (no further updates of its columns) while the others run on.
struct CgScalars {
  double* rsold;     // [ncols]
  double* alpha;     // [ncols]
  double* beta;      // [ncols]
  double* err_hist;  // [max_hist+1][stride]: per group, then the maximum over the groups still running;
                     // row 0 = 1 (utils.py:519), unwritten rows = 0 (= stopped)
  int stride;        // ngroups + 1
  int ngroups;
  int Cg;            // columns per group
  int C;             // ngroups * Cg
};

// `while (err > tol)`, utils.py:521 (NaN stops the loop too): does iteration `it` run for ...
__device__ __forceinline__ bool cg_any_active(const CgScalars& sc, int it, double tol) {   // ... any group
  return sc.err_hist[(size_t)(it - 1) * sc.stride + sc.ngroups] > tol;
}
__device__ __forceinline__ bool cg_col_active(const CgScalars& sc, int it, double tol, int col) {   // ... this column's group
  return col < sc.C && sc.err_hist[(size_t)(it - 1) * sc.stride + col / sc.Cg] > tol;
}


struct CgRhsRows {            // optional forms of the right-hand side and of the result (glx_cg_groups_rows)
  int64_t nb = 0;
  const int32_t* rows = nullptr;
  const void* vals = nullptr;
  const double* out_scale = nullptr;
};


// the reference-order column reductions in block form (cg_seqsum.hip; arithmetic in seqsum_exact.h).  mode 0: alpha = rsold / sum p*Ap;
// 1: beta = sum r*r / rsold, rsold = sum r*r; 2: rsold = sum r*r -- the results of cg.hip's cg_seqsum_dpp_kernel, bit for bit.
struct SsWork {
  char* rec;
  double *bsum, *csum;
  unsigned long long* mask;
  unsigned long long* stats;  // [3 modes][4]: blocks taken as plain integers / through their record / row by row (may be null)
  int nchunks;
};
int glx_seqsum_chunks(int64_t n);
int glx_seqsum_prepare();
int glx_seqsum_max_chunks();
size_t glx_seqsum_rec_bytes(int ncols, int nchunks);
size_t glx_seqsum_sum_doubles(int ncols, int nchunks, int which);
int glx_seqsum_run(int mode, const double* prod, int64_t n, int ncols_all, int C, const CgScalars& sc, int it, double tol, const SsWork& w,
                   hipStream_t st);

// the tolerance-mode solve (cg_fused.hip); arguments as cg.hip's cg_run
int glx_cg_run_fused(glx_graph* A, const void* B, void* X, int C, int Cg, double tol, int64_t max_iter, int* iters_out, double* err_out,
                     int flags, const int32_t* mask_rows, const int32_t* mask_ptr, const CgRhsRows& rr);
// records -> dense (n, C) in the caller's row order, every row times scale[row] (device pointers)
int glx_cg_unpack_scaled(int dtype, const void* rec, void* dense, int64_t n, const RecLayout& L, const int32_t* perm, const double* scale,
                         hipStream_t st);
#define CG_NEED(ptr, bytes) do { int rc_ = b.need((void**)&(ptr), (bytes)); if (rc_) return rc_; } while (0)
