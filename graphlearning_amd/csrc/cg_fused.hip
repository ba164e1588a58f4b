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
#include "glx_internal.h"
#include <algorithm>

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
