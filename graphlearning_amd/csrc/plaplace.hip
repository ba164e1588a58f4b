// Game-theoretic p-Laplace equation, Jacobi iteration of the upper and lower barrier functions:
// lp_iterate_main of the reference's C extension (c_code/lp_iterate.cpp:35-125), reached through
// graph.plaplace(..., fast=False) (graphlearning/graph.py:1262-1278).  One thread per vertex walks
// the vertex's stored entries in the caller's order (min / max / sequential sum of
// w_ij (u_j - u_i), separate multiply and add roundings) for both barriers at once -- a vertex
// record is the pair (uu_i, ul_i), one 16-byte gather per neighbour.  All iterations are enqueued
// from the host in chunks; an iteration that finds the stop condition of an earlier one
// (`err < tol && it > 10`, err = max(uu - ul) of the iterate that was read) raises a flag and it
// and all later ones return at once, so both iterates of the stopping step survive in the two
// buffers exactly as they do behind the reference's swapped pointers.
#include "glx_internal.h"
#include <algorithm>
#include <vector>

static const int LP_CHUNK = 64;

struct LpBufs {
  double2 *a = nullptr, *b = nullptr;
  int64_t* start = nullptr;
  int32_t *nbr = nullptr, *bdy = nullptr;
  double *w = nullptr, *invdeg = nullptr, *val = nullptr;
  unsigned long long* err = nullptr;
  int* stop = nullptr;
  unsigned long long* h_err = nullptr;
  hipStream_t stream = nullptr;
  ~LpBufs() {
    hipFree(a); hipFree(b); hipFree(start); hipFree(nbr); hipFree(bdy); hipFree(w); hipFree(invdeg); hipFree(val); hipFree(err);
    hipFree(stop);
    if (h_err) hipHostFree(h_err);
    if (stream) hipStreamDestroy(stream);
  }
};

__global__ __launch_bounds__(256) void lp_sweep_kernel(const double2* __restrict__ xin, double2* __restrict__ xout,
                                                       const int64_t* __restrict__ start, const int32_t* __restrict__ nbr,
                                                       const double* __restrict__ W, const double* __restrict__ invdeg,
                                                       const int32_t* __restrict__ bdy, const double* __restrict__ val, double dt,
                                                       double delta, int64_t n, int it, double tol,
                                                       unsigned long long* __restrict__ err, int* __restrict__ stop) {
#pragma clang fp contract(off)
  // did the loop of lp_iterate.cpp:74 end at an earlier iteration?  (uniform over the grid)
  if (*stop) return;
  if (it >= 1 && it - 1 > 10 && __longlong_as_double((long long)err[it - 1]) < tol) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *stop = 1;
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double e = 0.0;
  if (i < n) {
    const double2 me = xin[i];
    double minu = 0, maxu = 0, sumu = 0, minl = 0, maxl = 0, suml = 0;
    const int64_t j1 = start[i + 1];
    for (int64_t j = start[i]; j < j1; ++j) {     // lp_iterate.cpp:82-97
      const double2 x = xin[nbr[j]];
      const double w = W[j];
      const double du = x.x - me.x;
      const double tu = w * du;
      minu = (tu < minu) ? tu : minu;              // MIN / MAX of vector_operations.h: NaN leaves the bound alone
      maxu = (tu > maxu) ? tu : maxu;
      sumu = sumu + tu;
      const double dl = x.y - me.y;
      const double tl = w * dl;
      minl = (tl < minl) ? tl : minl;
      maxl = (tl > maxl) ? tl : maxl;
      suml = suml + tl;
    }
    const double id = invdeg[i];
    double2 out;
    {
      const double a1 = id * sumu, a2 = minu + maxu, a3 = delta * a2, a4 = a1 + a3, a5 = dt * a4;
      out.x = me.x + a5;
    }
    {
      const double a1 = id * suml, a2 = minl + maxl, a3 = delta * a2, a4 = a1 + a3, a5 = dt * a4;
      out.y = me.y + a5;
    }
    const int32_t bj = bdy[i];                     // Dirichlet values overwrite the update (:105-110)
    if (bj >= 0) { out.x = val[bj]; out.y = val[bj]; }
    xout[i] = out;
    const double gap = me.x - me.y;
    e = (gap > e) ? gap : e;                       // err = MAX(uu[i] - ul[i], err), err starts at 0 (:98)
  }
  __shared__ double s_e[256];
  s_e[threadIdx.x] = e;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off && s_e[threadIdx.x + off] > s_e[threadIdx.x]) s_e[threadIdx.x] = s_e[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && s_e[0] > 0.0) atomicMax(&err[it], (unsigned long long)__double_as_longlong(s_e[0]));
}

extern "C" int glx_lp_iterate(double* uu, double* ul, const int32_t* nbr, const int32_t* row, const double* W, const int32_t* ind,
                              const double* val, double p, int64_t T, double tol, int64_t n, int64_t M, int64_t m,
                              int64_t* iters_out, int device) {
  GLX_CHECK(uu && ul && (M == 0 || (nbr && row && W)) && (m == 0 || (ind && val)), GLX_EINVAL, "glx_lp_iterate: null argument");
  GLX_CHECK(n >= 1 && M >= 0 && m >= 0 && T >= 0, GLX_EINVAL, "glx_lp_iterate: bad sizes (n=%lld M=%lld m=%lld T=%lld)", (long long)n,
            (long long)M, (long long)m, (long long)T);
  GLX_CHECK(T <= (1ll << 24), GLX_EUNSUPPORTED, "glx_lp_iterate: T=%lld above the supported 2^24 iterations", (long long)T);
  GLX_HIP(hipSetDevice(device));
  // vertex blocks of the sorted entry list, inverse degrees, largest weight: lp_iterate.cpp:43-64
  const double alpha = 1 / p;
  const double delta = 1 - 2 / p;
  double dt = 0.9 / (alpha + 2 * delta);
  std::vector<int64_t> start(n + 1, 0);
  std::vector<double> invdeg(n, 0.0);
  int64_t j = 0;
  for (int64_t i = 0; i < n; ++i) {
    start[i] = j;
    double d = 0;
    while (j < M && row[j] == i) {
      GLX_CHECK(nbr[j] >= 0 && nbr[j] < n, GLX_EINVAL, "glx_lp_iterate: neighbour index %d out of range", nbr[j]);
      d += W[j];
      ++j;
    }
    invdeg[i] = alpha / d;
  }
  start[n] = j;     // entries past the last vertex's block (unsorted input) are never visited, as in the reference
  double maxw = 0;
  for (int64_t q = 0; q < M; ++q) maxw = (maxw > W[q]) ? maxw : W[q];
  dt = dt / maxw;
  std::vector<int32_t> bdy(n, -1);
  for (int64_t q = 0; q < m; ++q) {
    GLX_CHECK(ind[q] >= 0 && ind[q] < n, GLX_EINVAL, "glx_lp_iterate: boundary index %d out of range", ind[q]);
    bdy[ind[q]] = (int32_t)q;    // a vertex listed twice takes its last value, like the loop at :105-110
  }
  std::vector<double2> x0(n);
  for (int64_t i = 0; i < n; ++i) { x0[i].x = uu[i]; x0[i].y = ul[i]; }

  LpBufs b;
  GLX_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  hipStream_t st = b.stream;
  GLX_HIP(hipMalloc(&b.a, n * 16));
  GLX_HIP(hipMalloc(&b.b, n * 16));
  GLX_HIP(hipMalloc(&b.start, (n + 1) * 8));
  GLX_HIP(hipMalloc(&b.nbr, std::max<int64_t>(M, 1) * 4));
  GLX_HIP(hipMalloc(&b.w, std::max<int64_t>(M, 1) * 8));
  GLX_HIP(hipMalloc(&b.invdeg, n * 8));
  GLX_HIP(hipMalloc(&b.bdy, n * 4));
  GLX_HIP(hipMalloc(&b.val, std::max<int64_t>(m, 1) * 8));
  GLX_HIP(hipMalloc(&b.err, (T + 1) * 8));
  GLX_HIP(hipMalloc(&b.stop, 4));
  GLX_HIP(hipHostMalloc((void**)&b.h_err, LP_CHUNK * 8, hipHostMallocDefault));
  GLX_UP(glx_upload(b.a, x0.data(), n * 16, st, __func__));
  GLX_HIP(hipMemsetAsync(b.b, 0, n * 16, st));
  GLX_UP(glx_upload(b.start, start.data(), (n + 1) * 8, st, __func__));
  if (M > 0) {
    GLX_UP(glx_upload(b.nbr, nbr, M * 4, st, __func__));
    GLX_UP(glx_upload(b.w, W, M * 8, st, __func__));
  }
  GLX_UP(glx_upload(b.invdeg, invdeg.data(), n * 8, st, __func__));
  GLX_UP(glx_upload(b.bdy, bdy.data(), n * 4, st, __func__));
  if (m > 0) GLX_UP(glx_upload(b.val, val, m * 8, st, __func__));
  GLX_HIP(hipMemsetAsync(b.err, 0, (T + 1) * 8, st));
  GLX_HIP(hipMemsetAsync(b.stop, 0, 4, st));

  const unsigned grid = (unsigned)((n + 255) / 256);
  int64_t it = 0, stopped_at = -1;
  while (it < T && stopped_at < 0) {
    const int64_t it0 = it, end = std::min<int64_t>(T, it + LP_CHUNK);
    for (; it < end; ++it) {
      const double2* xin = (it & 1) ? b.b : b.a;
      double2* xout = (it & 1) ? b.a : b.b;
      hipLaunchKernelGGL(lp_sweep_kernel, dim3(grid), dim3(256), 0, st, xin, xout, (const int64_t*)b.start, (const int32_t*)b.nbr,
                         (const double*)b.w, (const double*)b.invdeg, (const int32_t*)b.bdy, (const double*)b.val, dt, delta, n, (int)it,
                         tol, b.err, b.stop);
      GLX_HIP(hipGetLastError());
    }
    GLX_HIP(hipMemcpyAsync(b.h_err, b.err + it0, (size_t)(end - it0) * 8, hipMemcpyDeviceToHost, st));
    GLX_HIP(hipStreamSynchronize(st));
    for (int64_t q = it0; q < end; ++q) {
      const double e = __builtin_bit_cast(double, b.h_err[q - it0]);
      if (e < tol && q > 10) { stopped_at = q; break; }     // lp_iterate.cpp:113
    }
  }
  // the caller's arrays are buffer `a`: whatever iterate last lived there (see the file header)
  GLX_UP(glx_download(x0.data(), b.a, n * 16, st, __func__));
  GLX_HIP(hipStreamSynchronize(st));
  for (int64_t i = 0; i < n; ++i) { uu[i] = x0[i].x; ul[i] = x0[i].y; }
  if (iters_out) *iters_out = stopped_at >= 0 ? stopped_at : T;
  return GLX_OK;
}
