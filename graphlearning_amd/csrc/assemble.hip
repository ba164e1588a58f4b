// kNN data -> symmetric sparse weight matrix on the device: weightmatrix.knn of the reference
// given knn_data (graphlearning/weightmatrix.py:134-187): kernel weights (:140-164), COO->CSR
// with duplicates summed (:171-175), symmetrisation (:177-183), zero diagonal + drop zeros
// (:185-186).  No global sort: every row of A has exactly k entries, so A^T is built by
// counting reverse neighbours (atomics) + a host scan of n counters, then one wavefront per
// row merges its forward and reverse lists with a bitonic sort in LDS and applies the
// symmetrisation rule per column.  Output: canonical CSR (sorted, no duplicates, no zeros).
#include "glx_internal.h"
#define GLX_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#include <algorithm>
#include <vector>
#include <utility>

static const int ROW_CAP = 1024;   // forward + reverse entries one wavefront can merge in LDS

enum { K_GIVEN = 0, K_UNIFORM = 1, K_GAUSSIAN = 2, K_SYMGAUSSIAN = 3, K_DISTANCE = 4, K_SINGULAR = 5 };
enum { SYM_NONE = 0, SYM_MEAN = 1, SYM_MAX = 2, SYM_SYMGAUSS = 3 };

// weights exactly as numpy forms them, operation by operation (weightmatrix.py:140-156)
__global__ void knn_weights_kernel(const int64_t* __restrict__ ind, const double* __restrict__ dist, int64_t n, int kk, int k,
                                   int kernel, const double* __restrict__ given, double* __restrict__ w) {
#pragma clang fp contract(off)
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int t = (int)(e % k);
  const double d = dist ? dist[i * kk + t] : 0.0;
  double v = 1.0;
  if (kernel == K_GIVEN) {
    v = given[i * k + t];
  } else if (kernel == K_GAUSSIAN) {
    const double dk = dist[i * kk + k - 1];
    const double D = d * d, eps = dk * dk;
    const double a = -4.0 * D;
    v = exp(a / eps);
  } else if (kernel == K_SYMGAUSSIAN) {
    const double ei = dist[i * kk + k - 1];
    const double ej = dist[ind[i * kk + t] * kk + k - 1];
    const double a = -4.0 * d;
    const double b = a * d;
    const double c = b / ei;
    v = exp(c / ej);
  } else if (kernel == K_DISTANCE) {
    v = d;
  } else if (kernel == K_SINGULAR) {
    v = 1.0 / (d == 0.0 ? 1.0 : d);
  }
  w[e] = v;
}

__global__ void count_reverse_kernel(const int64_t* __restrict__ ind, int64_t n, int kk, int k, int* __restrict__ rcnt,
                                     int* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int64_t j = ind[(e / k) * kk + (e % k)];
  if (j < 0 || j >= n) { *bad = 1; return; }
  atomicAdd(&rcnt[j], 1);
}

__global__ void fill_reverse_kernel(const int64_t* __restrict__ ind, const double* __restrict__ w, int64_t n, int kk, int k,
                                    const int64_t* __restrict__ roff, int* __restrict__ cursor, int* __restrict__ rsrc,
                                    double* __restrict__ rw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int64_t j = ind[i * kk + (e % k)];
  const int64_t pos = roff[j] + atomicAdd(&cursor[j], 1);
  rsrc[pos] = (int)i;
  rw[pos] = w[e];
}

// one wavefront per row: merge forward (tag 0) and reverse (tag 1) entries, combine per column
// mode 0: count kept entries -> rowcnt[i];  mode 1: write them at rowptr[i]
__global__ __launch_bounds__(256) void merge_rows_kernel(const int64_t* __restrict__ ind, const double* __restrict__ w, int64_t n, int kk,
                                                         int k, const int64_t* __restrict__ roff, const int* __restrict__ rsrc,
                                                         const double* __restrict__ rw, int sym, int mode, int* __restrict__ rowcnt,
                                                         const int64_t* __restrict__ rowptr, int* __restrict__ col_out,
                                                         double* __restrict__ val_out, int* __restrict__ overflow) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* key = (unsigned long long*)sm + (size_t)wave * ROW_CAP;                 // (col << 32) | (tag << 16) | seq
  double* val = (double*)(sm + (size_t)4 * ROW_CAP * 8) + (size_t)wave * ROW_CAP;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const int rc = sym == SYM_NONE ? 0 : (int)(roff[i + 1] - roff[i]);
  const int M = k + rc;
  if (M > ROW_CAP) {   // hub vertex: merged on the host (rare), see glx_knn_to_csr
    if (lane == 0 && !mode) { rowcnt[i] = -1; *overflow = 1; }
    return;
  }
  int P = 64;
  while (P < M) P <<= 1;
  for (int e = lane; e < P; e += 64) {
    unsigned long long kx = ~0ull;
    double v = 0.0;
    if (e < k) {
      kx = ((unsigned long long)(unsigned)ind[i * kk + e] << 32) | (0ull << 16) | (unsigned)e;
      v = w[i * k + e];
    } else if (e < M) {
      const int64_t p = roff[i] + (e - k);
      kx = ((unsigned long long)(unsigned)rsrc[p] << 32) | (1ull << 16) | (unsigned)(e - k);
      v = rw[p];
    }
    key[e] = kx;
    val[e] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < P / 2; t += 64) {
        const int lo = (t / stride) * stride * 2 + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long kl = key[lo], kh = key[hi];
        if (up ? (kh < kl) : (kl < kh)) {
          key[lo] = kh;
          key[hi] = kl;
          const double vl = val[lo];
          val[lo] = val[hi];
          val[hi] = vl;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // segment heads: first entry of each distinct column
  int kept_before = 0;   // running count of kept entries (uniform across the wave)
  const int64_t obase = mode ? rowptr[i] : 0;
  for (int e0 = 0; e0 < M; e0 += 64) {
    const int e = e0 + lane;
    bool keep = false;
    int c = 0;
    double v = 0.0;
    if (e < M) {
      c = (int)(key[e] >> 32);
      const bool head = e == 0 || (int)(key[e - 1] >> 32) != c;
      if (head) {
        double a = 0.0, b = 0.0;   // a = A[i,c] (forward, duplicates summed), b = A[c,i] (reverse)
        for (int q = e; q < M && (int)(key[q] >> 32) == c; ++q) {
          if ((key[q] >> 16) & 1) b = b + val[q]; else a = a + val[q];
        }
        if (sym == SYM_NONE) {
          v = a;
        } else if (sym == SYM_MEAN) {            // (W + W^T)/2, weightmatrix.py:183
          v = (a + b) / 2.0;
        } else if (sym == SYM_MAX) {             // utils.sparse_max(W, W^T), utils.py:263-286
          v = (b > a) ? b : (((a + b) > 0.0) ? a : 0.0);
        } else {                                 // W + W^T*(W^T>W) - W*(W^T>W), weightmatrix.py:181
          if (b > a) { const double s = a + b; v = s - a; } else v = a;
        }
        keep = (c != (int)i) && (v != 0.0);      // setdiag(0); eliminate_zeros(), weightmatrix.py:185-186
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (keep && mode) {
      const int pos = kept_before + __popcll(mask & ((1ull << lane) - 1ull));
      col_out[obase + pos] = c;
      val_out[obase + pos] = v;
    }
    kept_before += __popcll(mask);
  }
  if (!mode && lane == 0) rowcnt[i] = kept_before;
}

// the per-column combination rule, shared by the device kernel (above, inline) and the host path
static double combine_host(double a, double b, int sym) {
  if (sym == SYM_NONE) return a;
  if (sym == SYM_MEAN) return (a + b) / 2.0;
  if (sym == SYM_MAX) return (b > a) ? b : (((a + b) > 0.0) ? a : 0.0);
  if (b > a) { const double s = a + b; return s - a; }
  return a;
}

struct HostRow {
  std::vector<int32_t> col;
  std::vector<double> val;
};

struct AsmBufs {
  int64_t *ind = nullptr, *roff = nullptr, *rowptr = nullptr;
  double *dist = nullptr, *given = nullptr, *w = nullptr, *rw = nullptr, *val = nullptr;
  int *rcnt = nullptr, *cursor = nullptr, *rsrc = nullptr, *rowcnt = nullptr, *col = nullptr, *flag = nullptr;
  hipStream_t stream = nullptr;
  ~AsmBufs() {
    if (stream) hipStreamSynchronize(stream);   // pooled blocks are reused at once
    glx_pool_free(ind); glx_pool_free(roff); glx_pool_free(rowptr); glx_pool_free(dist); glx_pool_free(given); glx_pool_free(w); glx_pool_free(rw); glx_pool_free(val);
    glx_pool_free(rcnt); glx_pool_free(cursor); glx_pool_free(rsrc); glx_pool_free(rowcnt); glx_pool_free(col); glx_pool_free(flag);
    if (stream) hipStreamDestroy(stream);
  }
};

extern "C" int glx_knn_to_csr(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k, int kernel,
                              int sym, int32_t** rowptr_out, int32_t** col_out, double** val_out, int64_t* nnz_out, int device) {
  GLX_CHECK(ind && rowptr_out && col_out && val_out && nnz_out, GLX_EINVAL, "glx_knn_to_csr: null argument");
  GLX_CHECK(n >= 1 && k >= 1 && kk >= k, GLX_EINVAL, "glx_knn_to_csr: need n >= 1 and 1 <= k <= columns (n=%lld k=%d columns=%d)", (long long)n, k, kk);
  GLX_CHECK(kernel >= K_GIVEN && kernel <= K_SINGULAR, GLX_EINVAL, "glx_knn_to_csr: bad kernel id %d", kernel);
  GLX_CHECK(sym >= SYM_NONE && sym <= SYM_SYMGAUSS, GLX_EINVAL, "glx_knn_to_csr: bad symmetrisation id %d", sym);
  GLX_CHECK(kernel == K_GIVEN ? weights != nullptr : (kernel == K_UNIFORM || dist != nullptr), GLX_EINVAL, "glx_knn_to_csr: missing weights / distances");
  GLX_CHECK(n < (1ll << 31) && n * k < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_to_csr: n*k must fit int32");
  *rowptr_out = nullptr; *col_out = nullptr; *val_out = nullptr; *nnz_out = 0;
  GLX_HIP(hipSetDevice(device));
  AsmBufs b;
  GLX_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  hipStream_t st = b.stream;
  const int64_t ne = n * k;
  GLX_POOL(glx_pool_alloc((void**)&b.ind, (size_t)n * kk * 8));
  GLX_HIP(hipMemcpyAsync(b.ind, ind, (size_t)n * kk * 8, hipMemcpyHostToDevice, st));
  if (dist) {
    GLX_POOL(glx_pool_alloc((void**)&b.dist, (size_t)n * kk * 8));
    GLX_HIP(hipMemcpyAsync(b.dist, dist, (size_t)n * kk * 8, hipMemcpyHostToDevice, st));
  }
  if (kernel == K_GIVEN) {
    GLX_POOL(glx_pool_alloc((void**)&b.given, ne * 8));
    GLX_HIP(hipMemcpyAsync(b.given, weights, ne * 8, hipMemcpyHostToDevice, st));
  }
  GLX_POOL(glx_pool_alloc((void**)&b.w, ne * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rcnt, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.cursor, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.roff, (n + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowptr, (n + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowcnt, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.rsrc, ne * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.rw, ne * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.flag, 8));
  GLX_HIP(hipMemsetAsync(b.rcnt, 0, (n + 1) * 4, st));
  GLX_HIP(hipMemsetAsync(b.cursor, 0, (n + 1) * 4, st));
  GLX_HIP(hipMemsetAsync(b.flag, 0, 8, st));
  const unsigned ge = (unsigned)((ne + 255) / 256);
  hipLaunchKernelGGL(knn_weights_kernel, dim3(ge), dim3(256), 0, st, (const int64_t*)b.ind, (const double*)b.dist, n, kk, k, kernel,
                     (const double*)b.given, b.w);
  GLX_HIP(hipGetLastError());
  hipLaunchKernelGGL(count_reverse_kernel, dim3(ge), dim3(256), 0, st, (const int64_t*)b.ind, n, kk, k, b.rcnt, b.flag);
  GLX_HIP(hipGetLastError());
  std::vector<int> rcnt(n);
  int flags[2] = {0, 0};
  GLX_HIP(hipMemcpyAsync(rcnt.data(), b.rcnt, n * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  GLX_CHECK(!flags[0], GLX_EINVAL, "glx_knn_to_csr: neighbour index out of range");
  std::vector<int64_t> roff(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) roff[i + 1] = roff[i] + rcnt[i];
  GLX_HIP(hipMemcpyAsync(b.roff, roff.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(fill_reverse_kernel, dim3(ge), dim3(256), 0, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                     (const int64_t*)b.roff, b.cursor, b.rsrc, b.rw);
  GLX_HIP(hipGetLastError());
  const size_t shm = (size_t)4 * ROW_CAP * 16;
  GLX_HIP(hipFuncSetAttribute((const void*)merge_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const unsigned gr = (unsigned)((n + 3) / 4);
  hipLaunchKernelGGL(merge_rows_kernel, dim3(gr), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                     (const int64_t*)b.roff, (const int*)b.rsrc, (const double*)b.rw, sym, 0, b.rowcnt, (const int64_t*)nullptr,
                     (int*)nullptr, (double*)nullptr, b.flag + 1);
  GLX_HIP(hipGetLastError());
  std::vector<int> rowcnt(n);
  GLX_HIP(hipMemcpyAsync(rowcnt.data(), b.rowcnt, n * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  // hub vertices (more than ROW_CAP forward+reverse neighbours; high-dimensional data has them):
  // merged here on the host with the same rule, written into the CSR after the device pass
  std::vector<std::pair<int64_t, HostRow>> hubs;
  if (flags[1]) {
    std::vector<double> wrow(k);
    std::vector<int> rs;
    std::vector<double> rwv;
    for (int64_t i = 0; i < n; ++i) {
      if (rowcnt[i] >= 0) continue;
      const int rc = (int)(roff[i + 1] - roff[i]);
      rs.resize(rc);
      rwv.resize(rc);
      GLX_HIP(hipMemcpy(wrow.data(), b.w + i * k, (size_t)k * 8, hipMemcpyDeviceToHost));
      GLX_HIP(hipMemcpy(rs.data(), b.rsrc + roff[i], (size_t)rc * 4, hipMemcpyDeviceToHost));
      GLX_HIP(hipMemcpy(rwv.data(), b.rw + roff[i], (size_t)rc * 8, hipMemcpyDeviceToHost));
      struct Ent { int32_t col; int tag; int seq; double v; };
      std::vector<Ent> ents;
      ents.reserve(k + rc);
      for (int t = 0; t < k; ++t) ents.push_back({(int32_t)ind[i * kk + t], 0, t, wrow[t]});
      for (int q = 0; q < rc; ++q) ents.push_back({rs[q], 1, q, rwv[q]});
      std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) {
        return x.col != y.col ? x.col < y.col : (x.tag != y.tag ? x.tag < y.tag : x.seq < y.seq);
      });
      HostRow hr;
      for (size_t e = 0; e < ents.size();) {
        const int32_t c = ents[e].col;
        double a = 0.0, bb = 0.0;
        for (; e < ents.size() && ents[e].col == c; ++e) {
          if (ents[e].tag) bb = bb + ents[e].v; else a = a + ents[e].v;
        }
        const double v = combine_host(a, bb, sym);
        if (c != (int32_t)i && v != 0.0) { hr.col.push_back(c); hr.val.push_back(v); }
      }
      rowcnt[i] = (int)hr.col.size();
      hubs.emplace_back(i, std::move(hr));
    }
  }
  std::vector<int64_t> rp(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) rp[i + 1] = rp[i] + rowcnt[i];
  const int64_t nnz = rp[n];
  GLX_CHECK(nnz < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_to_csr: nnz %lld does not fit the int32 CSR of the reference", (long long)nnz);
  GLX_HIP(hipMemcpyAsync(b.rowptr, rp.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
  GLX_POOL(glx_pool_alloc((void**)&b.col, std::max<size_t>(nnz * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&b.val, std::max<size_t>(nnz * 8, 8)));
  hipLaunchKernelGGL(merge_rows_kernel, dim3(gr), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                     (const int64_t*)b.roff, (const int*)b.rsrc, (const double*)b.rw, sym, 1, b.rowcnt, (const int64_t*)b.rowptr,
                     b.col, b.val, b.flag + 1);
  GLX_HIP(hipGetLastError());
  int32_t* h_rp = (int32_t*)malloc((n + 1) * 4);
  int32_t* h_col = (int32_t*)malloc(std::max<size_t>(nnz * 4, 4));
  double* h_val = (double*)malloc(std::max<size_t>(nnz * 8, 8));
  if (!h_rp || !h_col || !h_val) {
    free(h_rp); free(h_col); free(h_val);
    glx_set_error("glx_knn_to_csr: host allocation failed");
    return GLX_ENOMEM;
  }
  for (int64_t i = 0; i <= n; ++i) h_rp[i] = (int32_t)rp[i];
  hipError_t e1 = hipMemcpyAsync(h_col, b.col, nnz * 4, hipMemcpyDeviceToHost, st);
  hipError_t e2 = hipMemcpyAsync(h_val, b.val, nnz * 8, hipMemcpyDeviceToHost, st);
  hipError_t e3 = hipStreamSynchronize(st);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    free(h_rp); free(h_col); free(h_val);
    glx_set_error("glx_knn_to_csr: download failed");
    return GLX_EHIP;
  }
  for (auto& hb : hubs) {
    std::copy(hb.second.col.begin(), hb.second.col.end(), h_col + rp[hb.first]);
    std::copy(hb.second.val.begin(), hb.second.val.end(), h_val + rp[hb.first]);
  }
  *rowptr_out = h_rp;
  *col_out = h_col;
  *val_out = h_val;
  *nnz_out = nnz;
  return GLX_OK;
}
