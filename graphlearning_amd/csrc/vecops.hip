// Dense vector kernels of the multi right-hand-side conjugate gradient on DEVICE records (reference graphlearning/utils.py:
// 483-532: `np.sum(p*Ap, axis=0)`, `x += alpha*p; r -= alpha*Ap`, `p = r + (rsnew/rsold)*p`), device-pointer entry points for
// the vertex-partitioned (multi-GPU) solve of dist.py: every rank runs them on its own rows, the column sums are then added
// over the ranks by one all-reduce each.  Tolerance mode by construction (the summation order depends on the partition);
// inside a rank the order is fixed: 1024-row blocks, a fixed tree inside a block, block partials added in block order.
#include "glx_internal.h"

static const int DOT_ROWS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void rec_dots_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n, int ld, int C, int ncp,
                                                       double* __restrict__ partial) {
  // thread = (row lane, column): ncp = power of two >= C columns, 256 / ncp rows in flight
  const int c = threadIdx.x % ncp, rl = threadIdx.x / ncp, rstep = 256 / ncp;
  const int64_t r0 = (int64_t)blockIdx.x * DOT_ROWS, r1 = min(n, r0 + DOT_ROWS);
  double s = 0.0;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += rstep) s += (double)a[r * ld + c] * (double)b[r * ld + c];
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    double t = sm[c];
    for (int q = 1; q < rstep; ++q) t += sm[q * ncp + c];     // fixed order
    partial[(size_t)blockIdx.x * ncp + c] = t;
  }
}

// block partials -> out[c]: thread = (column, one of 256 / ncp runs of blocks), eight partial sums per thread whose loads do not wait
// for one another, the runs combined in order (a fixed summation order for a given n and C; one dependent load + add per block
// was 100 us at 10^6 rows)
__global__ __launch_bounds__(256) void rec_dots_finish_kernel(const double* __restrict__ partial, int64_t nblk, int ncp, int C, double* __restrict__ out) {
  __shared__ double sm[256];
  const int c = threadIdx.x % ncp, part = threadIdx.x / ncp, nparts = 256 / ncp;
  double s = 0.0;
  if (c < C) {
    const int64_t per = (nblk + nparts - 1) / nparts;
    const int64_t b0 = part * per, b1 = min(nblk, b0 + per);
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += partial[(b + q) * ncp + c];
    }
    for (int q = 0; b < b1; ++b, ++q) a[q] += partial[b * ncp + c];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C) {
    double t = 0.0;
    for (int q = 0; q < nparts; ++q) t += sm[q * ncp + threadIdx.x];
    out[threadIdx.x] = t;
  }
}

static int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p *= 2;
  return p;
}

// out[c] = sum over the n records of a[i, c] * b[i, c] (fp64 products and sums).  partial: device scratch of at least
// glx_rec_dots_scratch(n, C) doubles.
extern "C" int64_t glx_rec_dots_scratch(int64_t n, int C) {
  return ((n + DOT_ROWS - 1) / DOT_ROWS + 1) * (int64_t)pow2_at_least(C > 0 ? C : 1);
}

extern "C" int glx_rec_dots_dev(const void* a, const void* b, int64_t n, int C, int dtype, int has_w, double* partial, double* out,
                                void* stream) {
  GLX_CHECK(a && b && partial && out, GLX_EINVAL, "glx_rec_dots_dev: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  GLX_CHECK(C <= 256, GLX_EUNSUPPORTED, "glx_rec_dots_dev: C=%d above 256", C);
  const int ncp = pow2_at_least(C);
  const int64_t nblk = (n + DOT_ROWS - 1) / DOT_ROWS;
  hipStream_t st = (hipStream_t)stream;
  if (nblk > 0) {
    if (dtype == GLX_F32)
      hipLaunchKernelGGL(rec_dots_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)a, (const float*)b, n, L.ld, C, ncp, partial);
    else
      hipLaunchKernelGGL(rec_dots_kernel<double>, dim3((unsigned)nblk), dim3(256), 0, st, (const double*)a, (const double*)b, n, L.ld, C, ncp, partial);
    GLX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(rec_dots_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, nblk, ncp, C, out);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void rec_axpy2_kernel(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p, const T* __restrict__ Ap,
                                                        const double* __restrict__ alpha, int64_t n, int ld, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ld) return;
  const int c = (int)(i % ld);
  if (c >= C) return;
  const T al = (T)alpha[c];            // numpy: alpha has the arrays' dtype (`rsold / np.sum(p*Ap, axis=0)`)
  x[i] = x[i] + al * p[i];
  r[i] = r[i] - al * Ap[i];
}

// x += alpha * p ; r -= alpha * Ap, column-wise alpha (device, fp64[C])
extern "C" int glx_rec_axpy2_dev(void* x, void* r, const void* p, const void* Ap, const double* alpha, int64_t n, int C, int dtype, int has_w,
                                 void* stream) {
  GLX_CHECK(x && r && p && Ap && alpha, GLX_EINVAL, "glx_rec_axpy2_dev: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  const int64_t tot = n * L.ld;
  if (tot == 0) return GLX_OK;
  const unsigned grid = (unsigned)((tot + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(rec_axpy2_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)x, (float*)r, (const float*)p, (const float*)Ap, alpha, n, L.ld, C);
  else
    hipLaunchKernelGGL(rec_axpy2_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (double*)x, (double*)r, (const double*)p, (const double*)Ap, alpha, n, L.ld, C);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void rec_xpby_kernel(T* __restrict__ p, const T* __restrict__ r, const double* __restrict__ beta, int64_t n, int ld,
                                                       int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ld) return;
  const int c = (int)(i % ld);
  if (c >= C) return;
  p[i] = r[i] + (T)beta[c] * p[i];
}

// p = r + beta * p, column-wise beta (device, fp64[C]); p may have more records than n (a halo region behind the owned rows)
extern "C" int glx_rec_xpby_dev(void* p, const void* r, const double* beta, int64_t n, int C, int dtype, int has_w, void* stream) {
  GLX_CHECK(p && r && beta, GLX_EINVAL, "glx_rec_xpby_dev: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  const int64_t tot = n * L.ld;
  if (tot == 0) return GLX_OK;
  const unsigned grid = (unsigned)((tot + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(rec_xpby_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)p, (const float*)r, beta, n, L.ld, C);
  else
    hipLaunchKernelGGL(rec_xpby_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (double*)p, (const double*)r, beta, n, L.ld, C);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// ---- exp_cr on an array (what the Gaussian weights of assemble.hip are made of; the tests check it against a 60-digit exp) ----
#include "exp_cr.h"
__global__ void exp_cr_kernel(const double* __restrict__ x, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = exp_cr(x[i]);
}
extern "C" int glx_exp_cr(const double* x, double* out, int64_t n, int device) {
  GLX_CHECK((x && out) || n == 0, GLX_EINVAL, "glx_exp_cr: null argument");
  if (n <= 0) return GLX_OK;
  GLX_HIP(hipSetDevice(device));
  double *dx = nullptr, *dy = nullptr;
  int rc = glx_pool_alloc((void**)&dx, (size_t)n * 8);
  if (!rc) rc = glx_pool_alloc((void**)&dy, (size_t)n * 8);
  if (!rc) {
    rc = glx_upload_sync(dx, x, (size_t)n * 8, "glx_exp_cr");
    hipError_t e = hipSuccess;
    if (!rc) {
      hipLaunchKernelGGL(exp_cr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const double*)dx, dy, n);
      e = hipGetLastError();
      if (e == hipSuccess) rc = glx_download_sync(out, dy, (size_t)n * 8, "glx_exp_cr");
    }
    if (!rc && e != hipSuccess) { glx_set_error("glx_exp_cr: %s", hipGetErrorString(e)); rc = GLX_EHIP; }
  }
  glx_pool_free(dx);
  glx_pool_free(dy);
  return rc;
}

__global__ __launch_bounds__(256) void glx_zero_kernel(unsigned long long* __restrict__ p, int64_t nwords) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) p[i] = 0ull;
}
int glx_zero_async(void* p, size_t bytes, hipStream_t st) {
  GLX_CHECK(((uintptr_t)p % 8 == 0) && bytes % 8 == 0, GLX_EINVAL, "glx_zero_async: 8-byte words only");
  if (bytes == 0) return GLX_OK;
  const int64_t nw = (int64_t)(bytes / 8);
  hipLaunchKernelGGL(glx_zero_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, (nw + 255) / 256))), dim3(256), 0, st,
                     (unsigned long long*)p, nw);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}
