// Tolerance-mode conjugate gradient (reduce='tree', GLX_CG_TREE): utils.conjgrad of the reference
// (graphlearning/utils.py:483-532) in TWO launches per iteration.
//
//   launch 1 (sweep.hip, spmm_sell_kernel<T,G,false,true> with p.fused):  Ap = A p, Dirichlet rows held at zero, and the column
//            dots p.Ap, r.Ap, Ap.Ap as per-group partial sums; one extra workgroup closes the PREVIOUS iteration meanwhile
//            (rsold = r.r, err = sqrt(sum r.r) per system, utils.py:527-530);
//   launch 2 (here):  alpha = rsold / p.Ap (utils.py:524) and beta = ||r - alpha Ap||^2 / rsold, the norm taken from the identity
//            rsold - 2 alpha r.Ap + alpha^2 Ap.Ap; then x += alpha p, r -= alpha Ap, p = r + beta p in one pass (utils.py:525-529)
//            and the partial sums of the true r.r.
//
// beta is the only quantity that is not the reference's expression evaluated literally (the reference forms it from the true
// r.r, which would need a third launch behind the reduction); the two agree to rounding -- on the Laplace systems of the parity
// suite the iterates differ by < 1e-13 and the iteration counts are identical -- and the stop test and every alpha use the true
// r.r.  The exact mode (cg.hip) remains the default wherever bit-identical iterates are promised.
// All reductions are deterministic: per-workgroup partial sums combined in a fixed order (no float atomics), see CgDev.
#include "cg_internal.h"
#include <string.h>
#include <atomic>
#include <limits>
#include <thread>
#include <stdlib.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename T> struct V4Of;
template <> struct V4Of<float> { typedef f32x4 type; };
template <> struct V4Of<double> { typedef f64x4 type; };

template <typename T>
__global__ __launch_bounds__(256) void cg_fused_update_kernel(T* __restrict__ x, T* __restrict__ r, T* __restrict__ p,
                                                              const T* __restrict__ Ap, int64_t n, int ld, int nvec, const CgDev cg,
                                                              double tol, int rows_per_block) {
#pragma clang fp contract(off)
  typedef typename V4Of<T>::type V4;
  const int it = *cg.it_b;
  const double* prev = cg.err_hist + (size_t)(it - 1) * cg.stride;
  if (it > cg.max_iter || !(prev[cg.ngroups] > tol)) return;   // `while (err > tol)`, utils.py:521: this iteration does not run
  if (blockIdx.x == 0 && threadIdx.x == 0) *cg.it_a = it + 1;  // (this kernel reads it_b only)
  __shared__ double s_part[256 * 4];
  __shared__ double s_tot[3 * 256];
  __shared__ double s_ab[2 * 256];
  const int nvq = ld / 4;
  const int rows_pass = 256 / nvq;
  const int cv = threadIdx.x % nvq, rs = threadIdx.x / nvq;
  const int ncols = nvec * 4;
  const bool on = rs < rows_pass && cv < nvec;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  // the first two rows of every thread travel while the scalars are formed (the kernel is a latency chain, not a bandwidth
  // problem: 40 MB at config 3 in a single wave of workgroups)
  struct Rows { V4 r, p, ap, x; };
  auto load_rows = [&](int64_t row) -> Rows {
    Rows q;
    q.r = q.p = q.ap = q.x = V4{0, 0, 0, 0};
    if (on && row < r1) {
      const size_t o = (size_t)row * ld + cv * 4;
      q.r = *(const V4*)(r + o);
      q.p = *(const V4*)(p + o);
      q.ap = *(const V4*)(Ap + o);
      q.x = *(const V4*)(x + o);
    }
    return q;
  };
  const int64_t row_first = r0 + rs;
  Rows nA = load_rows(row_first), nB = load_rows(row_first + rows_pass);
  // totals of the three dots: the per-group sums of the SpMM kernel, added in group order by every workgroup for itself
  glx_reduce_rows<false>(cg.part1g, (int64_t)cg.ngrp, 3 * ncols, s_part, [&](int q, double tot) { s_tot[q] = tot; });
  if ((int)threadIdx.x < ncols) {
    const int col = threadIdx.x;
    double al = 0.0, be = 0.0;
    if (col < cg.C && prev[col / cg.Cg] > tol) {
      const double rsq = cg.rsold[col];
      al = rsq / s_tot[col];                                   // utils.py:524
      const double t1 = al * s_tot[ncols + col];
      const double t2 = (al * al) * s_tot[2 * ncols + col];
      be = ((rsq - (t1 + t1)) + t2) / rsq;                     // ||r - alpha Ap||^2 / rsold  (= rsnew / rsold, utils.py:529)
    }
    s_ab[col] = al;
    s_ab[256 + col] = be;
  }
  __syncthreads();
  bool act[4] = {false, false, false, false};
  V4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = cv * 4 + e;
      act[e] = col < cg.C && prev[col / cg.Cg] > tol;     // columns of a converged system keep their values
      a[e] = (T)s_ab[col];
      b[e] = (T)s_ab[256 + col];
    }
  }
  double acc[4] = {0, 0, 0, 0};
  auto step = [&](int64_t row, const Rows& q) {
    if (!(on && row < r1)) return;
    const size_t o = (size_t)row * ld + cv * 4;
    V4 rv = q.r, pv = q.p, xv = q.x;
    const V4 t1 = a * pv;
    const V4 xn = xv + t1;
    const V4 t2 = a * q.ap;
    const V4 rn = rv - t2;
    const V4 t3 = b * pv;
    const V4 pn = rn + t3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xv[e] = act[e] ? xn[e] : xv[e];
      rv[e] = act[e] ? rn[e] : rv[e];
      pv[e] = act[e] ? pn[e] : pv[e];
    }
    *(V4*)(x + o) = xv;
    *(V4*)(r + o) = rv;
    *(V4*)(p + o) = pv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double sq = (double)rv[e] * (double)rv[e];
      acc[e] = acc[e] + sq;
    }
  };
  for (int64_t row = row_first; row < r1; row += 2 * rows_pass) {
    const Rows cA = nA, cB = nB;
    if (row + 2 * rows_pass < r1) {       // (uniform per thread row; the next pair travels while this one is updated)
      nA = load_rows(row + 2 * rows_pass);
      nB = load_rows(row + 3 * rows_pass);
    }
    step(row, cA);
    step(row + rows_pass, cB);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) s_part[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if ((int)threadIdx.x < ncols) {
    const int c = threadIdx.x, ccv = c / 4, ce = c % 4;
    double s = 0.0;
    for (int q = 0; q < rows_pass; ++q) s += s_part[(q * nvq + ccv) * 4 + ce];
    cg.part2[(size_t)blockIdx.x * ncols + c] = s;       // read after the launch boundary (glx_cg_close_iteration)
  }
}

// the last iteration of a chunk is closed here (the SpMM kernel of the next chunk would do it as well: idempotent)
__global__ __launch_bounds__(256) void cg_fused_close_kernel(const CgDev cg, double tol) {
  __shared__ double s_tmp[256];
  glx_cg_close_iteration(cg, *cg.it_a, tol, s_tmp);
}

int glx_cg_fused_close(const CgDev& cg, double tol, hipStream_t st) {
  hipLaunchKernelGGL(cg_fused_close_kernel, dim3(1), dim3(256), 0, st, cg, tol);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

int glx_cg_fused_update_blocks(int64_t n, int* rows_per_block) {
  // one pass of a workgroup covers 256 / (ld / 4) rows; two passes per workgroup keep eight 32-byte loads per thread in flight
  // and the number of partial rows the last arriver adds up small
  int rpb = 128;
  while ((n + rpb - 1) / rpb > 2048) rpb *= 2;
  *rows_per_block = rpb;
  return (int)std::max<int64_t>((n + rpb - 1) / rpb, 1);
}

int glx_cg_fused_update(int dtype, void* x, void* r, void* p, const void* ap, int64_t n, const RecLayout& L, const CgDev& cg,
                        double tol, hipStream_t st) {
  int rpb = 0;
  const int nb = glx_cg_fused_update_blocks(n, &rpb);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(cg_fused_update_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (float*)x, (float*)r, (float*)p,
                       (const float*)ap, n, L.ld, L.nvec, cg, tol, rpb);
  else
    hipLaunchKernelGGL(cg_fused_update_kernel<double>, dim3((unsigned)nb), dim3(256), 0, st, (double*)x, (double*)r, (double*)p,
                       (const double*)ap, n, L.ld, L.nvec, cg, tol, rpb);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// ---- set-up kernels ----------------------------------------------------------------------------------------------------------
// right-hand side rows given one by one (all other rows were zeroed): r[rec(rows[q])][:] = vals[q][:]; and the Dirichlet rows of
// every system as bits of rowmask (zeroed before)
template <typename T>
__global__ __launch_bounds__(256) void cg_fused_scatter_kernel(T* __restrict__ rec, int ld, int C, const int32_t* __restrict__ rows,
                                                               const T* __restrict__ vals, int64_t nb, const int32_t* __restrict__ inv,
                                                               unsigned* __restrict__ rowmask, const int32_t* __restrict__ mrows,
                                                               const int32_t* __restrict__ mgroup, int64_t nmask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nb * C) {
    const int64_t q = i / C;
    const int c = (int)(i % C);
    const int32_t row = rows[q];
    rec[(size_t)(inv ? inv[row] : row) * ld + c] = vals[i];
  }
  if (i < nmask) {
    const int32_t row = mrows[i];
    atomicOr(&rowmask[inv ? inv[row] : row], 1u << mgroup[i]);
  }
}

// x = 0 (unless it holds x0), p = r (utils.py:516), the partial sums of r.r (utils.py:517; closed as "iteration 0" by the first
// SpMM launch), and the counters of the solve
template <typename T>
__global__ __launch_bounds__(256) void cg_fused_init_kernel(T* __restrict__ x, const T* __restrict__ r, T* __restrict__ p, int64_t n, int ld,
                                                            int nvec, const CgDev cg, int rows_per_block, int keep_x) {
#pragma clang fp contract(off)
  typedef typename V4Of<T>::type V4;
  __shared__ double s_part[256 * 4];
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      *cg.it_a = 1;
      *cg.it_b = 1;
      *cg.closed = -1;
    }
    for (int q = threadIdx.x; q <= cg.ngroups; q += 256) cg.err_hist[q] = 1.0;      // utils.py:519
    for (int q = threadIdx.x; q < cg.ngrp; q += 256) cg.tick1[q] = 0u;               // (a launch behind the last iteration of the previous solve may have left arrivals behind)
  }
  const int nvq = ld / 4;
  const int rows_pass = 256 / nvq;
  const int cv = threadIdx.x % nvq, rs = threadIdx.x / nvq;
  const int ncols = nvec * 4;
  double acc[4] = {0, 0, 0, 0};
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  if (rs < rows_pass && cv < nvec) {
    for (int64_t row = r0 + rs; row < r1; row += rows_pass) {
      const size_t o = (size_t)row * ld + cv * 4;
      const V4 rv = *(const V4*)(r + o);
      *(V4*)(p + o) = rv;
      if (!keep_x) *(V4*)(x + o) = V4{0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double q = (double)rv[e] * (double)rv[e];
        acc[e] = acc[e] + q;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) s_part[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if ((int)threadIdx.x < ncols) {
    const int c = threadIdx.x, ccv = c / 4, ce = c % 4;
    double s = 0.0;
    for (int q = 0; q < rows_pass; ++q) s += s_part[(q * nvq + ccv) * 4 + ce];
    cg.part2[(size_t)blockIdx.x * ncols + c] = s;
  }
}

// records -> dense (n, C) in the caller's row order, every row times its scale (ssl.laplace: `v = M*v`, ssl.py:1250)
template <typename T>
__global__ __launch_bounds__(256) void cg_unpack_scaled_kernel(const T* __restrict__ rec, T* __restrict__ dense, int64_t n, int C, int ld,
                                                               const int32_t* __restrict__ perm, const double* __restrict__ scale) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * C) return;
  const int64_t row = i / C;
  const int c = (int)(i % C);
  const int64_t orow = perm ? (int64_t)perm[row] : row;
  dense[orow * C + c] = (T)((T)scale[orow] * rec[row * ld + c]);
}

int glx_cg_unpack_scaled(int dtype, const void* rec, void* dense, int64_t n, const RecLayout& L, const int32_t* perm, const double* scale,
                         hipStream_t st) {
  const unsigned grid = (unsigned)std::max<int64_t>((n * L.C + 255) / 256, 1);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(cg_unpack_scaled_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)rec, (float*)dense, n, L.C, L.ld, perm, scale);
  else
    hipLaunchKernelGGL(cg_unpack_scaled_kernel<double>, dim3(grid), dim3(256), 0, st, (const double*)rec, (double*)dense, n, L.C, L.ld, perm, scale);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// ---- the solve ---------------------------------------------------------------------------------------------------------------
// Iterations per captured chunk.  Every hand-over from one launched graph to the next costs ~37 us of idle device (kernel trace:
// profiles/r04_cg_tree_chunks.txt) whether or not the next one was enqueued in time, an iteration past convergence ~9 us (its two
// kernels exit at once): long chunks while the end is far, short ones near it.
static const int CG_NLEN = 3;
static const int CG_LEN[CG_NLEN] = {32, 16, 4};
static const int CG_LONG = 32;                 // (the longest: sizes the history buffers)

int glx_cg_run_fused(glx_graph* A, const void* B, void* X, int C, int Cg, double tol, int64_t max_iter, int* iters_out, double* err_out,
                     int flags, const int32_t* mask_rows, const int32_t* mask_ptr, const CgRhsRows& rr) {
  const int ngroups = C / Cg;
  const int stride = ngroups + 1;
  const int64_t n = A->n_rows;
  const int dtype = A->dtype;
  RecLayout L;
  int rc = glx_make_layout(C, dtype, false, &L);
  if (rc) return rc;
  SellPlan* plan = nullptr;
  rc = glx_graph_plan(A, L.G, &plan, true);      // the relaxed image: a row's segments sum independently (graph.hip)
  if (rc) return rc;
  const size_t es = L.esize;
  const int ncols = L.nvec * 4;
  GLX_CHECK(ncols <= 256, GLX_EUNSUPPORTED, "glx_cg_multi: C=%d too wide for the column reducer", C);
  GLX_CHECK(256 / (L.ld / 4) >= 1, GLX_EUNSUPPORTED, "glx_cg_multi: record too wide");
  GLX_CHECK(max_iter < (1ll << 24), GLX_EUNSUPPORTED, "glx_cg_multi: max_iter %lld exceeds the supported 2^24-1", (long long)max_iter);
  const int64_t nmask = (mask_rows && mask_ptr) ? mask_ptr[ngroups] : 0;
  GLX_CHECK(nmask == 0 || ngroups <= 32, GLX_EUNSUPPORTED,
            "glx_cg_groups_masked: at most 32 systems with Dirichlet rows per tolerance-mode solve (got %d)", ngroups);
  for (int64_t q = 0; q < nmask; ++q)
    GLX_CHECK(mask_rows[q] >= 0 && mask_rows[q] < n, GLX_EINVAL, "glx_cg_groups_masked: row %d out of range", mask_rows[q]);
  const int64_t nb_spmm = std::max<int64_t>(glx_spmm_blocks(plan), 1);
  int rpb = 0;
  const int nb2 = glx_cg_fused_update_blocks(n, &rpb);
  // groups of SpMM workgroups: small enough that a group's last arriver adds its rows in one round of loads, few enough that
  // every workgroup of the update kernel can add the groups for itself
  const int grp = (int)std::max<int64_t>(32, (nb_spmm + 63) / 64);
  const int64_t ngrp = (nb_spmm + grp - 1) / grp;
  const int nq = 3 * ncols;
  const int64_t hist_cap = max_iter + 2 + 2 * CG_LONG;   // whole chunks of rows are copied, one chunk ahead

  if (!A->cg_ws) A->cg_ws = new CgBufs();
  CgBufs& b = *(CgBufs*)A->cg_ws;
  const size_t recb = std::max<size_t>((size_t)n * L.ld * es, 64);
  if (!b.stream) GLX_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  if (!b.side) GLX_HIP(hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking));
  for (int q = 0; q < 3; ++q)
    if (!b.f_ev[q]) GLX_HIP(hipEventCreateWithFlags(&b.f_ev[q], hipEventDisableTiming));
  hipStream_t st = b.stream;
  CG_NEED(b.x, recb);
  CG_NEED(b.r, recb);
  CG_NEED(b.p, recb);
  CG_NEED(b.ap, recb);
  CG_NEED(b.dense, (size_t)n * C * es);
  CG_NEED(b.scal, (size_t)3 * ncols * 8);
  CG_NEED(b.err_hist, (size_t)hist_cap * stride * 8);
  CG_NEED(b.f_part1, (size_t)nb_spmm * nq * 8);
  CG_NEED(b.f_part1g, (size_t)ngrp * nq * 8);
  CG_NEED(b.f_part2, (size_t)nb2 * ncols * 8);
  CG_NEED(b.f_tick, (size_t)(ngrp + 1) * 4);
  CG_NEED(b.f_it, 64);
  if (nmask) CG_NEED(b.f_rowmask, (size_t)n * 4);
  {
    // the host's mirror of the residual history: the maximum slot of every row a solve may write holds NaN = "not yet written"
    // A marker is only where the LAYOUT of the solve that armed it put it: another number of systems (ssl_trials' stacked solve
    // followed by a single fit on the same operator) or a longer history moves the marker slots onto words that hold an earlier
    // solve's residuals, which the poll below would take for rows already written.  So the markers are re-armed row by row only
    // while the layout stays the same; any change (or a new buffer) fills the whole mirror.
    const bool fresh = !b.h_hist || b.h_hist_rows < hist_cap * stride || b.h_hist_stride != stride;
    int rc_ = b.need_host(&b.h_hist, (size_t)hist_cap * stride * 8);
    if (rc_) return rc_;
    const double not_yet = std::numeric_limits<double>::quiet_NaN();
    if (fresh) {
      const int64_t words = (int64_t)(b.cap[(void**)&b.h_hist] / 8);
      for (int64_t j = 0; j < words; ++j) b.h_hist[j] = not_yet;
      b.h_hist_rows = words;
      b.h_hist_stride = stride;
    } else {
      for (int64_t j = 0; j <= std::min<int64_t>(b.h_hist_dirty, hist_cap - 1); ++j) b.h_hist[(size_t)j * stride + ngroups] = not_yet;
    }
    b.h_hist_dirty = 0;
  }

  // one upload: [right-hand side rows | Dirichlet rows | their systems | values | output scale]
  const size_t o_rows = 0, o_mrows = o_rows + (size_t)rr.nb * 4, o_mgrp = o_mrows + (size_t)nmask * 4;
  const size_t o_vals = (o_mgrp + (size_t)nmask * 4 + 15) / 16 * 16;
  const size_t o_scale = (o_vals + (rr.rows ? (size_t)rr.nb * C * es : 0) + 15) / 16 * 16;
  const size_t stage_bytes = o_scale + (rr.out_scale ? (size_t)n * 8 : 0);
  if (stage_bytes) {
    CG_NEED(b.f_stage, stage_bytes);
    { int rc_ = b.need_host(&b.h_stage, stage_bytes); if (rc_) return rc_; }
    if (rr.rows && rr.nb) {
      memcpy(b.h_stage + o_rows, rr.rows, (size_t)rr.nb * 4);
      memcpy(b.h_stage + o_vals, rr.vals, (size_t)rr.nb * C * es);
    }
    for (int g = 0; g < ngroups && nmask; ++g)
      for (int q = mask_ptr[g]; q < mask_ptr[g + 1]; ++q) {
        ((int32_t*)(b.h_stage + o_mrows))[q] = mask_rows[q];
        ((int32_t*)(b.h_stage + o_mgrp))[q] = g;
      }
    if (rr.out_scale) memcpy(b.h_stage + o_scale, rr.out_scale, (size_t)n * 8);
    GLX_HIP(hipMemcpyAsync(b.f_stage, b.h_stage, stage_bytes, hipMemcpyHostToDevice, st));
  }
  const bool keep_x = (flags & GLX_CG_X0) != 0;
  if (keep_x) {   // X holds x0 on entry; B is the caller's r0 = b - A@x0 (utils.py:510-514)
    GLX_UP(glx_upload(b.dense, X, (size_t)n * C * es, st, __func__));
    rc = glx_pack_records(b.dense, b.x, n, L, dtype, nullptr, st, A->d_perm);
    if (rc) return rc;
  }
  if (rr.rows) {
    GLX_HIP(hipMemsetAsync(b.r, 0, recb, st));
  } else {
    GLX_UP(glx_upload(b.dense, B, (size_t)n * C * es, st, __func__));
    rc = glx_pack_records(b.dense, b.r, n, L, dtype, nullptr, st, A->d_perm);   // r = b - A@0 = b (utils.py:514)
    if (rc) return rc;
  }
  if (nmask) GLX_HIP(hipMemsetAsync(b.f_rowmask, 0, (size_t)n * 4, st));
  if ((rr.rows && rr.nb) || nmask) {
    const int64_t work = std::max<int64_t>(rr.rows ? rr.nb * C : 0, nmask);
    const dim3 grid((unsigned)((work + 255) / 256));
    if (dtype == GLX_F32)
      hipLaunchKernelGGL(cg_fused_scatter_kernel<float>, grid, dim3(256), 0, st, (float*)b.r, L.ld, C, (const int32_t*)(b.f_stage + o_rows),
                         (const float*)(b.f_stage + o_vals), rr.rows ? rr.nb : 0, (const int32_t*)A->d_inv, b.f_rowmask,
                         (const int32_t*)(b.f_stage + o_mrows), (const int32_t*)(b.f_stage + o_mgrp), nmask);
    else
      hipLaunchKernelGGL(cg_fused_scatter_kernel<double>, grid, dim3(256), 0, st, (double*)b.r, L.ld, C, (const int32_t*)(b.f_stage + o_rows),
                         (const double*)(b.f_stage + o_vals), rr.rows ? rr.nb : 0, (const int32_t*)A->d_inv, b.f_rowmask,
                         (const int32_t*)(b.f_stage + o_mrows), (const int32_t*)(b.f_stage + o_mgrp), nmask);
    GLX_HIP(hipGetLastError());
  }
  CgDev cd;
  memset(&cd, 0, sizeof(cd));
  cd.rsold = b.scal;
  cd.err_hist = b.err_hist;
  cd.stride = stride;
  cd.ngroups = ngroups;
  cd.Cg = Cg;
  cd.C = C;
  cd.max_iter = (int)max_iter;
  cd.it_a = b.f_it;
  cd.it_b = b.f_it + 4;
  cd.closed = b.f_it + 8;
  cd.r = (const char*)b.r;
  cd.part1 = b.f_part1;
  cd.part1g = b.f_part1g;
  cd.tick1 = b.f_tick;
  cd.grp = grp;
  cd.ngrp = (int)ngrp;
  cd.part2 = b.f_part2;
  cd.nb2 = nb2;
  {
    void* view = nullptr;
    GLX_HIP(hipHostGetDevicePointer(&view, b.h_hist, 0));
    cd.host_hist = (double*)view;
  }
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(cg_fused_init_kernel<float>, dim3((unsigned)nb2), dim3(256), 0, st, (float*)b.x, (const float*)b.r, (float*)b.p, n, L.ld,
                       L.nvec, cd, rpb, keep_x ? 1 : 0);
  else
    hipLaunchKernelGGL(cg_fused_init_kernel<double>, dim3((unsigned)nb2), dim3(256), 0, st, (double*)b.x, (const double*)b.r, (double*)b.p, n,
                       L.ld, L.nvec, cd, rpb, keep_x ? 1 : 0);
  GLX_HIP(hipGetLastError());

  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.plan = plan;
  a.L = L;
  a.dtype = dtype;
  a.xin = b.p;
  a.xout = b.ap;
  a.n_rows = n;
  a.exit_tol = tol;
  a.act_cg = Cg;
  a.act_c = C;
  a.rowmask = nmask ? b.f_rowmask : nullptr;
  a.cg = &cd;
  auto enqueue_chunk = [&](int iters) -> int {
    for (int q = 0; q < iters; ++q) {
      int rc2 = glx_launch_spmm(a, st);                                                      // Ap = A@p, dots; closes the previous iteration
      if (rc2) return rc2;
      rc2 = glx_cg_fused_update(dtype, b.x, b.r, b.p, b.ap, n, L, cd, tol, st);              // alpha, beta, x, r, p, r.r
      if (rc2) return rc2;
    }
    return glx_cg_fused_close(cd, tol, st);                                                  // the chunk's last err
  };
  // everything the captured kernels were given: a replay is only valid for the same arguments
  std::vector<unsigned long long> key = {(unsigned long long)(uintptr_t)plan, (unsigned long long)(uintptr_t)plan->d_col,
                                         (unsigned long long)(uintptr_t)b.x, (unsigned long long)(uintptr_t)b.r,
                                         (unsigned long long)(uintptr_t)b.p, (unsigned long long)(uintptr_t)b.ap,
                                         (unsigned long long)(uintptr_t)b.scal, (unsigned long long)(uintptr_t)b.err_hist,
                                         (unsigned long long)(uintptr_t)b.f_part1, (unsigned long long)(uintptr_t)b.f_part1g,
                                         (unsigned long long)grp, (unsigned long long)(uintptr_t)b.f_part2,
                                         (unsigned long long)(uintptr_t)b.f_tick, (unsigned long long)(uintptr_t)b.f_it,
                                         (unsigned long long)(uintptr_t)a.rowmask, (unsigned long long)(uintptr_t)cd.host_hist, (unsigned long long)C, (unsigned long long)Cg,
                                         (unsigned long long)max_iter, (unsigned long long)n, (unsigned long long)dtype, 0ull};
  memcpy(&key.back(), &tol, 8);
  if (!b.f_exec[0] || b.f_key != key) {
    for (int v = 0; v < CG_NLEN; ++v) {
      if (b.f_exec[v]) { hipGraphExecDestroy(b.f_exec[v]); b.f_exec[v] = nullptr; }
      hipGraph_t graph;
      GLX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      rc = enqueue_chunk(CG_LEN[v]);
      hipError_t e = hipStreamEndCapture(st, &graph);
      if (rc) return rc;
      GLX_HIP(e);
      GLX_HIP(hipGraphInstantiate(&b.f_exec[v], graph, nullptr, nullptr, 0));
      GLX_HIP(hipGraphDestroy(graph));
    }
    b.f_key = key;
  }

  std::vector<int64_t> iters(ngroups, 0);       // iterations that ran, per system (utils.py:522 `i`)
  std::vector<double> err(ngroups, 1.0);        // utils.py:519
  std::vector<char> done(ngroups, !(1.0 > tol));
  int running = 0;
  for (int g = 0; g < ngroups; ++g) running += !done[g];
  double e_prev = 1.0, e_last = 1.0;            // the two latest maxima over the running systems the host has seen
  double margin_seen = INFINITY;                // min over every decision taken of |err - tol| / tol (glx_cg_last_stop_margin)
  auto read_history = [&](const double* h, int64_t it0, int64_t cnt) {
    for (int64_t q = 0; q < cnt && running > 0; ++q) {
      // iteration it0+q+1 ran for every system whose previous err was > tol; its err decides the next one
      for (int g = 0; g < ngroups; ++g) {
        if (done[g]) continue;
        iters[g] = it0 + q + 1;
        err[g] = h[(size_t)q * stride + g];
        // EVERY residual norm a running system shows is a stop decision (CG's residual norms are not monotone: one that came within
        // the band of tol on the far side iterations ago could have stopped the reference's order of additions there)
        if (tol > 0 && err[g] == err[g]) margin_seen = std::min(margin_seen, fabs(err[g] - tol) / tol);
        if (!(err[g] > tol)) { done[g] = 1; --running; }
      }
      e_prev = e_last;
      e_last = h[(size_t)q * stride + ngroups];
    }
  };
  // The GPU never waits for the host: the next chunk is launched before the history of the previous one is looked at (kernels of
  // iterations past convergence exit at once).  The first chunk has 16 iterations and the one launched blind behind it 4; from then
  // on the residuals the host has seen choose -- CG's error falls roughly geometrically, the last ratio predicts the iterations still
  // needed, and the longest chunk that does not overshoot them by more than two is launched.
  struct Flight { int slot; int64_t it0; int cnt; };
  Flight fl[2];
  int nfl = 0, slot = 0;
  int64_t launched = 0, looked = 0;
  auto launch_chunk = [&](int v) -> int {
    const int len = CG_LEN[v];
    GLX_HIP(hipGraphLaunch(b.f_exec[v], st));
    // (nothing else goes into the stream between two chunks: the closing kernels write the history into the host's page-locked
    // mirror themselves -- an event record + copy here cost a ~37 us bubble per chunk, profiles/r04_cg_tree_chunks.txt)
    fl[nfl].slot = slot;
    fl[nfl].it0 = launched;
    fl[nfl].cnt = len;
    ++nfl;
    launched += len;
    slot ^= 1;
    return GLX_OK;
  };
  auto next_kind = [&]() -> int {
    if (looked < 4) return CG_NLEN - 1;                           // nothing seen yet: the short chunk behind the first one
    if (!(e_last > tol) || !(e_last < e_prev)) return 1;
    const double rate = e_last / e_prev;
    const double need = log(tol / e_last) / log(rate);           // iterations still needed after `looked`
    const double open = need - (double)(launched - looked);      // ... of which not yet launched
    for (int v = 0; v < CG_NLEN - 1; ++v)
      if ((double)CG_LEN[v] <= open + 2.0) return v;
    return CG_NLEN - 1;
  };
  if (running > 0 && max_iter > 0) {
    rc = launch_chunk(1);
    if (rc) return rc;
    while (running > 0 && looked < max_iter) {
      if (launched < max_iter && nfl < 2) {
        rc = launch_chunk(next_kind());
        if (rc) return rc;
      }
      const int64_t cnt = std::min<int64_t>(fl[0].cnt, max_iter - fl[0].it0);
      for (int64_t q = 0; q < cnt && running > 0; ++q) {
        // row it0+q+1 is written by the kernels of the chunk in flight: iteration it0+q+1 runs because the row before said so
        const volatile double* slot_max = b.h_hist + (size_t)(fl[0].it0 + q + 1) * stride + ngroups;
        int spins = 0;
        bool idle_seen = false;
        while (*slot_max != *slot_max) {
          if (++spins < 64) continue;
          spins = 0;
          std::this_thread::yield();
          const hipError_t qe = hipStreamQuery(st);
          if (qe == hipSuccess) {                    // everything launched has finished: the row must be there by now
            if (idle_seen) { glx_set_error("glx_cg: residual history row %lld was never written", (long long)(fl[0].it0 + q + 1)); return GLX_EHIP; }
            idle_seen = true;
          } else if (qe != hipErrorNotReady) {
            GLX_HIP(qe);
          }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        read_history((const double*)b.h_hist + (size_t)(fl[0].it0 + q + 1) * stride, fl[0].it0 + q, 1);
      }
      looked = fl[0].it0 + fl[0].cnt;
      fl[0] = fl[1];
      --nfl;
    }
    b.h_hist_dirty = launched + 1;
  }
  if (rr.out_scale) {
    rc = glx_cg_unpack_scaled(dtype, b.x, b.dense, n, L, A->d_perm, (const double*)(b.f_stage + o_scale), st);
  } else {
    rc = glx_unpack_records(b.x, b.dense, n, L, dtype, st, A->d_perm);
  }
  if (rc) return rc;
  GLX_UP(glx_download(X, b.dense, (size_t)n * C * es, st, __func__));
  GLX_HIP(hipStreamSynchronize(st));
  // How close the stop decisions of this solve came to going the other way (glx_cg_last_stop_margin): the smallest relative distance
  // from `tol` of ANY residual norm a running system showed -- the one that passed the test, the last one that failed it, and every
  // earlier one (residual norms of CG are not monotone).  The reordered reductions of this mode move a residual norm by a relative
  // 1e-13 .. 1e-8 (more after hundreds of iterations); a decision with a margin below that could have gone the other way with the
  // reference's order of additions.
  b.last_stop_margin = margin_seen;
  for (int g = 0; g < ngroups; ++g) {
    if (iters_out) iters_out[g] = (int)iters[g];
    if (err_out) err_out[g] = err[g];
  }
  return GLX_OK;
}

// The smallest relative distance from `tol` of the residual norms the LAST tolerance-mode solve on this operator compared with it --
// every iteration of every system, not only the deciding ones (+inf: no such solve yet, or no iteration ran).  What ssl._solve(reduce='auto') reads: a solve whose
// stop hung on less than AUTO_STOP_BAND is handed back to the reference-order reductions.
extern "C" int glx_cg_last_stop_margin(glx_graph* A, double* margin_out) {
  GLX_CHECK(A && margin_out, GLX_EINVAL, "glx_cg_last_stop_margin: null argument");
  *margin_out = A->cg_ws ? ((CgBufs*)A->cg_ws)->last_stop_margin : INFINITY;
  return GLX_OK;
}
