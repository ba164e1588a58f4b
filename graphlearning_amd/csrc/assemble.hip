// kNN data -> symmetric sparse weight matrix on the device: weightmatrix.knn of the reference
// given knn_data (graphlearning/weightmatrix.py:134-187): kernel weights (:140-164), COO->CSR
// with duplicates summed (:171-175), symmetrisation (:177-183), zero diagonal + drop zeros
// (:185-186).  No global sort: every row of A has exactly k entries, so A^T is built by
// counting reverse neighbours (atomics) + a host scan of n counters, then one wavefront per
// row merges its forward and reverse lists with a bitonic sort in LDS and applies the
// symmetrisation rule per column; a hub vertex (more entries than one wavefront's LDS share holds)
// gets a whole workgroup that sorts in a global scratch.  Output: canonical CSR (sorted, no duplicates, no zeros).
#include "glx_internal.h"
#include "exp_cr.h"
#define GLX_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#include <algorithm>
#include <vector>
#include <utility>

static const int ROW_CAP = 1024;   // forward + reverse entries one wavefront can merge in LDS

enum { K_GIVEN = 0, K_UNIFORM = 1, K_GAUSSIAN = 2, K_SYMGAUSSIAN = 3, K_DISTANCE = 4, K_SINGULAR = 5 };
enum { SYM_NONE = 0, SYM_MEAN = 1, SYM_MAX = 2, SYM_SYMGAUSS = 3 };

// weights exactly as numpy forms them, operation by operation (weightmatrix.py:140-156); the exponential is the correctly
// rounded one of exp_cr.h (numpy's is the host libm's or its own SIMD kernel: within an ulp of this, host by host)
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
    v = exp_cr(a / eps);
  } else if (kernel == K_SYMGAUSSIAN) {
    const double ei = dist[i * kk + k - 1];
    // (an index outside [0, n) is reported by count_reverse_kernel, which runs behind this kernel: here it must not become an
    // out-of-bounds read -- user-supplied knn_data reaches this kernel unchecked)
    const int64_t jn = ind[i * kk + t];
    const double ej = dist[(jn >= 0 && jn < n ? jn : i) * kk + k - 1];
    const double a = -4.0 * d;
    const double b = a * d;
    const double c = b / ei;
    v = exp_cr(c / ej);
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
                                    unsigned short* __restrict__ rpos, double* __restrict__ rw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int64_t j = ind[i * kk + (e % k)];
  const int64_t pos = roff[j] + atomicAdd(&cursor[j], 1);
  rsrc[pos] = (int)i;
  rpos[pos] = (unsigned short)(e % k);   // duplicates of one source row are summed in the order the row lists them
  rw[pos] = w[e];
}

__device__ __forceinline__ double combine_entry(double a, double b, int sym) {
#pragma clang fp contract(off)
  if (sym == SYM_NONE) return a;
  if (sym == SYM_MEAN) return (a + b) / 2.0;                                // (W + W^T)/2, weightmatrix.py:183
  if (sym == SYM_MAX) return (b > a) ? b : (((a + b) > 0.0) ? a : 0.0);     // utils.sparse_max(W, W^T), utils.py:263-286
  if (b > a) { const double s = a + b; return s - a; }                      // W + W^T*(W^T>W) - W*(W^T>W), weightmatrix.py:181
  return a;
}

// one wavefront per row: merge forward (tag 0) and reverse (tag 1) entries, combine per column
// mode 0: count kept entries -> rowcnt[i];  mode 1: write them at rowptr[i]
__global__ __launch_bounds__(256) void merge_rows_kernel(const int64_t* __restrict__ ind, const double* __restrict__ w, int64_t n, int kk,
                                                         int k, const int64_t* __restrict__ roff, const int* __restrict__ rsrc,
                                                         const unsigned short* __restrict__ rpos, const double* __restrict__ rw, int sym,
                                                         int mode, int* __restrict__ rowcnt,
                                                         const int64_t* __restrict__ rowptr, int* __restrict__ col_out,
                                                         double* __restrict__ val_out, int* __restrict__ overflow, int64_t row_base = 0,
                                                         int m_lo = 0, const int* __restrict__ rowlist = nullptr, int nlist = 0) {
  // m_lo: rows of at most m_lo entries are merge_rows_small_kernel's (mode 2 only); rowlist: only these rows (the others are)
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long* key = (unsigned long long*)sm + (size_t)wave * ROW_CAP;                 // (col << 32) | (tag << 16) | seq
  double* val = (double*)(sm + (size_t)4 * ROW_CAP * 8) + (size_t)wave * ROW_CAP;
  int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (rowlist) {
    if (i >= nlist) return;
    i = rowlist[i];
  }
  if (i >= n) return;
  const int rc = sym == SYM_NONE ? 0 : (int)(roff[i + 1] - roff[i]);
  const int M = k + rc;
  if (M <= m_lo) return;
  if (M > ROW_CAP) {   // hub vertex: merge_hub_kernel's
    if (lane == 0 && mode != 1) { rowcnt[i] = -1; *overflow = 1; }
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
      kx = ((unsigned long long)(unsigned)rsrc[p] << 32) | (1ull << 16) | (unsigned)rpos[p];
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
  // mode 2: ONE pass -- the kept entries go to the row's slot in a scratch image (room for all M candidates: offset i k + roff[i],
  // known without a scan) and compact_rows_kernel moves them to their place once the row pointers exist; the second sort of
  // every row that modes 0 + 1 cost was a fifth of weightmatrix.knn's assembly
  const int64_t obase = mode == 1 ? rowptr[i] : (mode == 2 ? i * (int64_t)k + (sym == SYM_NONE ? 0 : roff[i]) : 0);
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
        v = combine_entry(a, b, sym);
        keep = (c != (int)(i + row_base)) && (v != 0.0);      // setdiag(0); eliminate_zeros(), weightmatrix.py:185-186
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
  if (mode != 1 && lane == 0) rowcnt[i] = kept_before;
}

// The same merge for rows of at most 64 entries (forward + reverse) -- all of them unless the graph has popular vertices --, mode 2
// only: one entry per lane, sorted by a bitonic network over the wavefront in REGISTERS (keys only: the source lane rides in the
// key's low bits and the value is fetched from it afterwards), 4 KB of LDS per workgroup for the segment pass instead of the 64 KB
// the general kernel sizes for ROW_CAP entries (two workgroups per CU, 21 LDS passes with a barrier each: 0.24 ms of the 1.2 ms
// assembly at config 2).  Rows above 64 entries are left to merge_rows_kernel (m_lo = 64).
__global__ __launch_bounds__(256) void merge_rows_small_kernel(const int64_t* __restrict__ ind, const double* __restrict__ w, int64_t n, int kk,
                                                               int k, const int64_t* __restrict__ roff, const int* __restrict__ rsrc,
                                                               const unsigned short* __restrict__ rpos, const double* __restrict__ rw, int sym,
                                                               int* __restrict__ rowcnt, int* __restrict__ col_out,
                                                               double* __restrict__ val_out, int64_t row_base) {
#pragma clang fp contract(off)
  __shared__ unsigned long long s_key[4][64];
  __shared__ double s_val[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= n) return;
  const int rc = sym == SYM_NONE ? 0 : (int)(roff[i + 1] - roff[i]);
  const int M = k + rc;
  if (M > 64) return;
  // key: column | tag (0 forward, 1 reverse) | position in the source row | lane -- (column, tag, position) is unique, so the
  // order is that of merge_rows_kernel's (col << 32) | (tag << 16) | seq
  unsigned long long kx = ~0ull;
  double v = 0.0;
  if (lane < k) {
    kx = ((unsigned long long)(unsigned)ind[i * kk + lane] << 32) | (0ull << 22) | ((unsigned long long)(unsigned)lane << 6) | (unsigned)lane;
    v = w[i * k + lane];
  } else if (lane < M) {
    const int64_t p = roff[i] + (lane - k);
    kx = ((unsigned long long)(unsigned)rsrc[p] << 32) | (1ull << 22) | ((unsigned long long)rpos[p] << 6) | (unsigned)lane;
    v = rw[p];
  }
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)(kx & 0xffffffffull), stride), hi = __shfl_xor((unsigned)(kx >> 32), stride);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      const bool up = (lane & size) == 0, lower = (lane & stride) == 0;
      const bool take_min = lower == up;
      kx = (take_min == (o < kx)) ? o : kx;
    }
  }
  {
    const int src = (int)(kx & 63);                   // (padding keys point at lane 63: its value is 0 and never read)
    const int vlo = __shfl(__double2loint(v), src), vhi = __shfl(__double2hiint(v), src);
    s_key[wave][lane] = kx;
    s_val[wave][lane] = __hiloint2double(vhi, vlo);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long* key = s_key[wave];
  const double* val = s_val[wave];
  const int64_t obase = i * (int64_t)k + (sym == SYM_NONE ? 0 : roff[i]);
  bool keep = false;
  int c = 0;
  double vv = 0.0;
  if (lane < M) {
    c = (int)(key[lane] >> 32);
    const bool head = lane == 0 || (int)(key[lane - 1] >> 32) != c;
    if (head) {
      double a = 0.0, b = 0.0;   // a = A[i,c] (forward, duplicates summed), b = A[c,i] (reverse)
      for (int q = lane; q < M && (int)(key[q] >> 32) == c; ++q) {
        if ((key[q] >> 22) & 1) b = b + val[q]; else a = a + val[q];
      }
      vv = combine_entry(a, b, sym);
      keep = (c != (int)(i + row_base)) && (vv != 0.0);      // setdiag(0); eliminate_zeros(), weightmatrix.py:185-186
    }
  }
  const unsigned long long mask = __ballot(keep);
  if (keep) {
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
    col_out[obase + pos] = c;
    val_out[obase + pos] = vv;
  }
  if (lane == 0) rowcnt[i] = __popcll(mask);
}

// rows of the scratch image of merge_rows_kernel's mode 2 to their place in the CSR arrays (hub rows, rowcnt < 0, are written by
// merge_hub_kernel); four rows per workgroup
__global__ __launch_bounds__(256) void compact_rows_kernel(const int* __restrict__ rowcnt, const int64_t* __restrict__ roff, int64_t n, int k, int sym,
                                                           const int64_t* __restrict__ rowptr, const int* __restrict__ tcol,
                                                           const double* __restrict__ tval, int* __restrict__ col_out, double* __restrict__ val_out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int cnt = rowcnt[i];
  if (cnt <= 0) return;
  const int64_t src = i * (int64_t)k + (sym == SYM_NONE ? 0 : roff[i]), dst = rowptr[i];
  for (int e = lane; e < cnt; e += 64) {
    col_out[dst + e] = tcol[src + e];
    val_out[dst + e] = tval[src + e];
  }
}

// hub vertices (more than ROW_CAP forward + reverse entries; high-dimensional data has them): one workgroup per hub,
// the same keys sorted by the same bitonic network in a global scratch of P entries (P = power of two >= M), then the
// same segment rule.  mode 0 sorts and counts, mode 1 writes from the scratch mode 0 left sorted.
__global__ __launch_bounds__(1024) void merge_hub_kernel(const int64_t* __restrict__ ind, const double* __restrict__ w, int kk, int k,
                                                         const int64_t* __restrict__ roff, const int* __restrict__ rsrc,
                                                         const unsigned short* __restrict__ rpos, const double* __restrict__ rw, int sym,
                                                         int mode, const int64_t* __restrict__ hub_row, const int64_t* __restrict__ hub_off,
                                                         unsigned long long* __restrict__ skey, double* __restrict__ sval,
                                                         int* __restrict__ hub_cnt, const int64_t* __restrict__ rowptr,
                                                         int* __restrict__ col_out, double* __restrict__ val_out, int64_t row_base = 0) {
#pragma clang fp contract(off)
  __shared__ int wcnt[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t i = hub_row[blockIdx.x];
  const int64_t P = hub_off[blockIdx.x + 1] - hub_off[blockIdx.x];
  unsigned long long* key = skey + hub_off[blockIdx.x];      // (col << 32) | (tag << 31) | seq
  double* val = sval + hub_off[blockIdx.x];
  const int64_t rc = roff[i + 1] - roff[i];
  const int64_t M = k + rc;
  if (!mode) {
    for (int64_t e = tid; e < P; e += 1024) {
      unsigned long long kx = ~0ull;
      double v = 0.0;
      if (e < k) {
        kx = ((unsigned long long)(unsigned)ind[i * kk + e] << 32) | (unsigned)e;
        v = w[i * k + e];
      } else if (e < M) {
        const int64_t p = roff[i] + (e - k);
        kx = ((unsigned long long)(unsigned)rsrc[p] << 32) | (1ull << 31) | (unsigned)rpos[p];
        v = rw[p];
      }
      key[e] = kx;
      val[e] = v;
    }
    __syncthreads();
    for (int64_t size = 2; size <= P; size <<= 1) {
      for (int64_t stride = size >> 1; stride > 0; stride >>= 1) {
        for (int64_t t = tid; t < P / 2; t += 1024) {
          const int64_t lo = (t / stride) * stride * 2 + (t % stride);
          const int64_t hi = lo + stride;
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
        __syncthreads();
      }
    }
  }
  int kept_before = 0;   // uniform across the workgroup
  const int64_t obase = mode ? rowptr[i] : 0;
  for (int64_t e0 = 0; e0 < M; e0 += 1024) {
    const int64_t e = e0 + tid;
    bool keep = false;
    int c = 0;
    double v = 0.0;
    if (e < M) {
      c = (int)(key[e] >> 32);
      if (e == 0 || (int)(key[e - 1] >> 32) != c) {          // first entry of a column
        double a = 0.0, b = 0.0;   // a = A[i,c] (forward, duplicates summed), b = A[c,i] (reverse)
        for (int64_t q = e; q < M && (int)(key[q] >> 32) == c; ++q) {
          if ((key[q] >> 31) & 1) b = b + val[q]; else a = a + val[q];
        }
        v = combine_entry(a, b, sym);
        keep = (c != (int)(i + row_base)) && (v != 0.0);      // setdiag(0); eliminate_zeros(), weightmatrix.py:185-186
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) wcnt[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int q = 0; q < 16; ++q) {
      if (q < wave) before += wcnt[q];
      total += wcnt[q];
    }
    if (keep && mode) {
      const int64_t pos = obase + kept_before + before + __popcll(mask & ((1ull << lane) - 1ull));
      col_out[pos] = c;
      val_out[pos] = v;
    }
    kept_before += total;
    __syncthreads();
  }
  if (!mode && tid == 0) hub_cnt[blockIdx.x] = kept_before;
}

struct AsmBufs {
  int64_t *ind = nullptr, *roff = nullptr, *rowptr = nullptr, *hub_row = nullptr, *hub_off = nullptr;
  unsigned long long* skey = nullptr;
  unsigned short* rpos = nullptr;
  double *dist = nullptr, *given = nullptr, *w = nullptr, *rw = nullptr, *val = nullptr, *sval = nullptr;
  int *rcnt = nullptr, *cursor = nullptr, *rsrc = nullptr, *rowcnt = nullptr, *col = nullptr, *flag = nullptr, *hub_cnt = nullptr;
  int* biglist = nullptr;        // rows above 64 entries (merge_rows_kernel's share when merge_rows_small_kernel runs)
  int* tcol = nullptr;           // scratch image of the one-pass merge (merge_rows_kernel mode 2)
  double* tval = nullptr;
  glx_work* work = nullptr;      // the device's cached stream
  hipStream_t stream = nullptr;
  bool borrowed = false;         // ind / dist belong to a glx_knn_result
  ~AsmBufs() {
    if (stream) hipStreamSynchronize(stream);   // pooled blocks are reused at once
    if (borrowed) { ind = nullptr; dist = nullptr; }
    glx_pool_free(ind); glx_pool_free(roff); glx_pool_free(rowptr); glx_pool_free(dist); glx_pool_free(given); glx_pool_free(w); glx_pool_free(rw); glx_pool_free(val);
    glx_pool_free(rcnt); glx_pool_free(cursor); glx_pool_free(rsrc); glx_pool_free(rowcnt); glx_pool_free(col); glx_pool_free(flag);
    glx_pool_free(hub_row); glx_pool_free(hub_off); glx_pool_free(skey); glx_pool_free(sval); glx_pool_free(hub_cnt); glx_pool_free(rpos);
    glx_pool_free(tcol); glx_pool_free(tval); glx_pool_free(biglist);
    glx_work_release(work);
  }
};

// cap < 0: the CSR arrays are allocated here (malloc; glx_free releases them); cap >= 0: *rowptr_out / *col_out / *val_out
// are the caller's buffers with room for n + 1 row pointers and cap entries
static int knn_to_csr_impl(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k, int kernel,
                           int sym, int64_t cap, int32_t** rowptr_out, int32_t** col_out, double** val_out, int64_t* nnz_out, int device,
                           const glx_knn_result* res = nullptr) {
  // res: the lists are the device-resident ones of a search result (borrowed: glx_knn_result_to_csr)
  GLX_CHECK(rowptr_out && col_out && val_out && nnz_out, GLX_EINVAL, "glx_knn_to_csr: null argument");
  GLX_CHECK(ind || res, GLX_EINVAL, "glx_knn_to_csr: null neighbour indices");
  GLX_CHECK(n >= 1 && k >= 1 && kk >= k, GLX_EINVAL, "glx_knn_to_csr: need n >= 1 and 1 <= k <= columns (n=%lld k=%d columns=%d)", (long long)n, k, kk);
  GLX_CHECK(kernel >= K_GIVEN && kernel <= K_SINGULAR, GLX_EINVAL, "glx_knn_to_csr: bad kernel id %d", kernel);
  GLX_CHECK(sym >= SYM_NONE && sym <= SYM_SYMGAUSS, GLX_EINVAL, "glx_knn_to_csr: bad symmetrisation id %d", sym);
  GLX_CHECK(kernel == K_GIVEN ? weights != nullptr : (kernel == K_UNIFORM || dist != nullptr || res), GLX_EINVAL, "glx_knn_to_csr: missing weights / distances");
  GLX_CHECK(n < (1ll << 31) && n * k < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_to_csr: n*k must fit int32");
  GLX_CHECK(k <= 65535, GLX_EUNSUPPORTED, "glx_knn_to_csr: at most 65535 neighbours per row (k=%d)", k);
  if (cap < 0) { *rowptr_out = nullptr; *col_out = nullptr; *val_out = nullptr; }
  *nnz_out = 0;
  GLX_HIP(hipSetDevice(device));
  AsmBufs b;
  {
    int rcw = glx_work_acquire(device, &b.work);
    if (rcw) return rcw;
  }
  b.stream = b.work->stream;
  hipStream_t st = b.stream;
  const int64_t ne = n * k;
  if (res) {
    b.ind = res->ind;
    b.dist = res->dist;
    b.borrowed = true;
  } else {
    GLX_POOL(glx_pool_alloc((void**)&b.ind, (size_t)n * kk * 8));
    GLX_UP(glx_upload(b.ind, ind, (size_t)n * kk * 8, st, __func__));
  }
  if (dist && !res) {
    GLX_POOL(glx_pool_alloc((void**)&b.dist, (size_t)n * kk * 8));
    GLX_UP(glx_upload(b.dist, dist, (size_t)n * kk * 8, st, __func__));
  }
  if (kernel == K_GIVEN) {
    GLX_POOL(glx_pool_alloc((void**)&b.given, ne * 8));
    GLX_UP(glx_upload(b.given, weights, ne * 8, st, __func__));
  }
  GLX_POOL(glx_pool_alloc((void**)&b.w, ne * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rcnt, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.cursor, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.roff, (n + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowptr, (n + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowcnt, (n + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.rsrc, ne * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.rw, ne * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rpos, ne * 2));
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
  // the per-row counts the host scans land in the work set's page-locked staging area: [n] reverse counts | [n] kept entries
  int* stage = nullptr;
  GLX_POOL(glx_work_stage(b.work, (size_t)n * 8 + 64, (void**)&stage));
  int* rcnt = stage;
  int* rowcnt = stage + n;
  int flags[2] = {0, 0};
  GLX_HIP(hipMemcpyAsync(rcnt, b.rcnt, n * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  GLX_CHECK(!flags[0], GLX_EINVAL, "glx_knn_to_csr: neighbour index out of range");
  std::vector<int64_t> roff(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) roff[i + 1] = roff[i] + rcnt[i];
  GLX_UP(glx_upload(b.roff, roff.data(), (n + 1) * 8, st, __func__));
  hipLaunchKernelGGL(fill_reverse_kernel, dim3(ge), dim3(256), 0, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                     (const int64_t*)b.roff, b.cursor, b.rsrc, b.rpos, b.rw);
  GLX_HIP(hipGetLastError());
  const size_t shm = (size_t)4 * ROW_CAP * 16;
  GLX_HIP(hipFuncSetAttribute((const void*)merge_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const unsigned gr = (unsigned)((n + 3) / 4);
  GLX_POOL(glx_pool_alloc((void**)&b.tcol, (size_t)(2 * ne) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.tval, (size_t)(2 * ne) * 8));
  {
    // rows of at most 64 entries: the register kernel; the general one only when some row has more (the host has the counts)
    const bool small_rows = k <= 64;
    std::vector<int> big;          // rows above 64 entries (popular vertices): the general kernel's, one wavefront each
    if (small_rows && sym != SYM_NONE)
      for (int64_t i = 0; i < n; ++i)
        if (k + rcnt[i] > 64) big.push_back((int)i);
    if (small_rows)
      hipLaunchKernelGGL(merge_rows_small_kernel, dim3(gr), dim3(256), 0, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, b.rowcnt,
                         b.tcol, b.tval, (int64_t)0);
    if (!small_rows) {
      hipLaunchKernelGGL(merge_rows_kernel, dim3(gr), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 2, b.rowcnt, (const int64_t*)nullptr,
                         b.tcol, b.tval, b.flag + 1, (int64_t)0);
    } else if (!big.empty()) {
      GLX_POOL(glx_pool_alloc((void**)&b.biglist, big.size() * 4));
      GLX_UP(glx_upload(b.biglist, big.data(), big.size() * 4, st, __func__));
      hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)((big.size() + 3) / 4)), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, n, kk, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 2, b.rowcnt, (const int64_t*)nullptr,
                         b.tcol, b.tval, b.flag + 1, (int64_t)0, 64, (const int*)b.biglist, (int)big.size());
    }
  }
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipMemcpyAsync(rowcnt, b.rowcnt, n * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  // hub vertices: a workgroup each, sorted in a global scratch (merge_hub_kernel)
  int64_t nh = 0;
  if (flags[1]) {
    std::vector<int64_t> hrow, hoff(1, 0);
    for (int64_t i = 0; i < n; ++i) {
      if (rowcnt[i] >= 0) continue;
      const int64_t M = k + (roff[i + 1] - roff[i]);
      int64_t P = 2048;
      while (P < M) P <<= 1;
      hrow.push_back(i);
      hoff.push_back(hoff.back() + P);
    }
    nh = (int64_t)hrow.size();
    GLX_POOL(glx_pool_alloc((void**)&b.hub_row, nh * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.hub_off, (nh + 1) * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.hub_cnt, nh * 4));
    GLX_POOL(glx_pool_alloc((void**)&b.skey, (size_t)hoff.back() * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.sval, (size_t)hoff.back() * 8));
    GLX_UP(glx_upload(b.hub_row, hrow.data(), nh * 8, st, __func__));
    GLX_UP(glx_upload(b.hub_off, hoff.data(), (nh + 1) * 8, st, __func__));
    hipLaunchKernelGGL(merge_hub_kernel, dim3((unsigned)nh), dim3(1024), 0, st, (const int64_t*)b.ind, (const double*)b.w, kk, k,
                       (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 0,
                       (const int64_t*)b.hub_row, (const int64_t*)b.hub_off, b.skey, b.sval, b.hub_cnt, (const int64_t*)nullptr,
                       (int*)nullptr, (double*)nullptr);
    GLX_HIP(hipGetLastError());
    std::vector<int> hcnt(nh);
    GLX_HIP(hipMemcpyAsync(hcnt.data(), b.hub_cnt, nh * 4, hipMemcpyDeviceToHost, st));
    GLX_HIP(hipStreamSynchronize(st));
    for (int64_t h = 0; h < nh; ++h) rowcnt[hrow[h]] = hcnt[h];
  }
  std::vector<int64_t> rp(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) rp[i + 1] = rp[i] + rowcnt[i];
  const int64_t nnz = rp[n];
  GLX_CHECK(nnz < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_to_csr: nnz %lld does not fit the int32 CSR of the reference", (long long)nnz);
  GLX_UP(glx_upload(b.rowptr, rp.data(), (n + 1) * 8, st, __func__));
  const bool own = cap < 0;
  GLX_CHECK(own || nnz <= cap, GLX_EINVAL, "glx_knn_to_csr_into: %lld entries, room for %lld", (long long)nnz, (long long)cap);
  int32_t* h_rp = own ? (int32_t*)malloc((n + 1) * 4) : *rowptr_out;
  int32_t* h_col = own ? (int32_t*)malloc(std::max<size_t>(nnz * 4, 4)) : *col_out;
  double* h_val = own ? (double*)malloc(std::max<size_t>(nnz * 8, 8)) : *val_out;
  if (!h_rp || !h_col || !h_val) {
    if (own) { free(h_rp); free(h_col); free(h_val); }
    glx_set_error("glx_knn_to_csr: host allocation failed");
    return GLX_ENOMEM;
  }
  // Page-locked destinations (the arrays _hip hands in) are written by the compaction kernels themselves through their device
  // view: no device copy of the result, no copy-engine transfer behind it (whose first use after fresh allocations cost 8 ms).
  void *v_col = nullptr, *v_val = nullptr;
  const bool direct = !own && nnz > 0 && hipHostGetDevicePointer(&v_col, h_col, 0) == hipSuccess && v_col &&
                      hipHostGetDevicePointer(&v_val, h_val, 0) == hipSuccess && v_val;
  (void)hipGetLastError();
  int32_t* out_col = (int32_t*)v_col;
  double* out_val = (double*)v_val;
  if (!direct) {
    GLX_POOL(glx_pool_alloc((void**)&b.col, std::max<size_t>(nnz * 4, 4)));
    GLX_POOL(glx_pool_alloc((void**)&b.val, std::max<size_t>(nnz * 8, 8)));
    out_col = b.col;
    out_val = b.val;
  }
  hipLaunchKernelGGL(compact_rows_kernel, dim3(gr), dim3(256), 0, st, (const int*)b.rowcnt, (const int64_t*)b.roff, n, k, sym, (const int64_t*)b.rowptr,
                     (const int*)b.tcol, (const double*)b.tval, out_col, out_val);
  hipError_t e0 = hipGetLastError();
  if (e0 == hipSuccess && nh) {
    hipLaunchKernelGGL(merge_hub_kernel, dim3((unsigned)nh), dim3(1024), 0, st, (const int64_t*)b.ind, (const double*)b.w, kk, k,
                       (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 1,
                       (const int64_t*)b.hub_row, (const int64_t*)b.hub_off, b.skey, b.sval, b.hub_cnt, (const int64_t*)b.rowptr,
                       out_col, out_val);
    e0 = hipGetLastError();
  }
  for (int64_t i = 0; i <= n; ++i) h_rp[i] = (int32_t)rp[i];
  hipError_t e1 = hipSuccess, e2 = hipSuccess;
  if (!direct) {       // (into memory that may be ordinary: through the checked download)
    if (glx_download(h_col, b.col, nnz * 4, st, "glx_knn_to_csr")) e1 = hipErrorUnknown;
    if (e1 == hipSuccess && glx_download(h_val, b.val, nnz * 8, st, "glx_knn_to_csr")) e2 = hipErrorUnknown;
  }
  hipError_t e3 = hipStreamSynchronize(st);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    if (own) { free(h_rp); free(h_col); free(h_val); }
    glx_set_error("glx_knn_to_csr: download failed");
    return GLX_EHIP;
  }
  *rowptr_out = h_rp;
  *col_out = h_col;
  *val_out = h_val;
  *nnz_out = nnz;
  return GLX_OK;
}

// ---- a BLOCK of rows of the weight matrix (sharded build, SURVEY.md 8e "symmetrisation by owner rank") ----------------------
// Rows [row_base, row_base + m) of weightmatrix.knn's result from the block's own lists (ind_own / w_own: m x k, weights given)
// and the reverse entries the other rows' owners sent: (rev_row = j in the block, rev_src = i, rev_pos = position of j in row
// i's list, rev_w = w_ij).  The same merge as glx_knn_to_csr -- forward and reverse entries of a row sorted by (column, kind,
// position), duplicates summed in list order, the symmetrisation rule per column, diagonal and zeros dropped -- so the rows are
// bit-identical to the rows of the whole matrix.  Columns are global ids (< n_cols).
__global__ void count_rev_rows_kernel(const int64_t* __restrict__ rev_row, int64_t n_rev, int64_t row_base, int64_t m, int* __restrict__ rcnt,
                                      int* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rev) return;
  const int64_t j = rev_row[e] - row_base;
  if (j < 0 || j >= m) { *bad = 1; return; }
  atomicAdd(&rcnt[j], 1);
}

__global__ void fill_rev_rows_kernel(const int64_t* __restrict__ rev_row, const int64_t* __restrict__ rev_src, const int64_t* __restrict__ rev_pos,
                                     const double* __restrict__ rev_w, int64_t n_rev, int64_t row_base, const int64_t* __restrict__ roff,
                                     int* __restrict__ cursor, int* __restrict__ rsrc, unsigned short* __restrict__ rpos, double* __restrict__ rw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rev) return;
  const int64_t j = rev_row[e] - row_base;
  const int64_t pos = roff[j] + atomicAdd(&cursor[j], 1);
  rsrc[pos] = (int)rev_src[e];
  rpos[pos] = (unsigned short)rev_pos[e];
  rw[pos] = rev_w[e];
}

__global__ void check_cols_kernel(const int64_t* __restrict__ ind, int64_t total, int64_t n_cols, int* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total && (ind[e] < 0 || ind[e] >= n_cols)) *bad = 1;
}

extern "C" int glx_knn_rows_to_csr(const int64_t* ind_own, const double* w_own, int64_t m, int k, int64_t n_cols, int64_t row_base,
                                   const int64_t* rev_row, const int64_t* rev_src, const int64_t* rev_pos, const double* rev_w, int64_t n_rev,
                                   int sym, int64_t cap, int32_t* rowptr_out, int32_t* col_out, double* val_out, int64_t* nnz_out, int device) {
  GLX_CHECK(ind_own && w_own && rowptr_out && col_out && val_out && nnz_out, GLX_EINVAL, "glx_knn_rows_to_csr: null argument");
  GLX_CHECK(m >= 0 && k >= 1 && n_cols >= 1 && row_base >= 0 && row_base + m <= n_cols, GLX_EINVAL, "glx_knn_rows_to_csr: bad sizes");
  GLX_CHECK(n_rev >= 0 && (n_rev == 0 || (rev_row && rev_src && rev_pos && rev_w)), GLX_EINVAL, "glx_knn_rows_to_csr: null reverse arrays");
  GLX_CHECK(sym == SYM_NONE || sym == SYM_MEAN || sym == SYM_MAX, GLX_EINVAL, "glx_knn_rows_to_csr: symmetrisation %d needs other rows' data", sym);
  GLX_CHECK(n_cols < (1ll << 31) && m * k < (1ll << 31) && n_rev < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_rows_to_csr: sizes must fit int32");
  GLX_CHECK(k <= 65535, GLX_EUNSUPPORTED, "glx_knn_rows_to_csr: at most 65535 neighbours per row (k=%d)", k);
  *nnz_out = 0;
  rowptr_out[0] = 0;
  if (m == 0) return GLX_OK;
  GLX_HIP(hipSetDevice(device));
  AsmBufs b;
  {
    int rcw = glx_work_acquire(device, &b.work);
    if (rcw) return rcw;
  }
  b.stream = b.work->stream;
  hipStream_t st = b.stream;
  const int64_t ne = m * k, nr = sym == SYM_NONE ? 0 : n_rev;
  // (AsmBufs' generic slots: dist <- rev_w staging, given <- rev_pos staging, hub_row / hub_off reused below)
  int64_t *d_rrow = nullptr, *d_rsrc64 = nullptr, *d_rpos64 = nullptr;
  struct Extra { int64_t *a = nullptr, *b = nullptr, *c = nullptr; ~Extra() { glx_pool_free(a); glx_pool_free(b); glx_pool_free(c); } } ex;
  GLX_POOL(glx_pool_alloc((void**)&b.ind, (size_t)ne * 8));
  GLX_UP(glx_upload(b.ind, ind_own, (size_t)ne * 8, st, __func__));
  GLX_POOL(glx_pool_alloc((void**)&b.w, (size_t)ne * 8));
  GLX_UP(glx_upload(b.w, w_own, (size_t)ne * 8, st, __func__));
  GLX_POOL(glx_pool_alloc((void**)&b.rcnt, (m + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.cursor, (m + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.roff, (m + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowptr, (m + 1) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rowcnt, (m + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.rsrc, std::max<size_t>(nr * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&b.rw, std::max<size_t>(nr * 8, 8)));
  GLX_POOL(glx_pool_alloc((void**)&b.rpos, std::max<size_t>(nr * 2, 2)));
  GLX_POOL(glx_pool_alloc((void**)&b.flag, 8));
  GLX_HIP(hipMemsetAsync(b.rcnt, 0, (m + 1) * 4, st));
  GLX_HIP(hipMemsetAsync(b.cursor, 0, (m + 1) * 4, st));
  GLX_HIP(hipMemsetAsync(b.flag, 0, 8, st));
  hipLaunchKernelGGL(check_cols_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, (const int64_t*)b.ind, ne, n_cols, b.flag);
  if (nr > 0) {
    GLX_POOL(glx_pool_alloc((void**)&ex.a, (size_t)nr * 8));
    GLX_POOL(glx_pool_alloc((void**)&ex.b, (size_t)nr * 8));
    GLX_POOL(glx_pool_alloc((void**)&ex.c, (size_t)nr * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.dist, (size_t)nr * 8));
    d_rrow = ex.a; d_rsrc64 = ex.b; d_rpos64 = ex.c;
    GLX_UP(glx_upload(d_rrow, rev_row, (size_t)nr * 8, st, __func__));
    GLX_UP(glx_upload(d_rsrc64, rev_src, (size_t)nr * 8, st, __func__));
    GLX_UP(glx_upload(d_rpos64, rev_pos, (size_t)nr * 8, st, __func__));
    GLX_UP(glx_upload(b.dist, rev_w, (size_t)nr * 8, st, __func__));
    hipLaunchKernelGGL(count_rev_rows_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, (const int64_t*)d_rrow, nr, row_base, m, b.rcnt, b.flag);
    hipLaunchKernelGGL(check_cols_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, (const int64_t*)d_rsrc64, nr, n_cols, b.flag);
  }
  GLX_HIP(hipGetLastError());
  int* stage = nullptr;
  GLX_POOL(glx_work_stage(b.work, (size_t)m * 8 + 64, (void**)&stage));
  int* rcnt = stage;
  int* rowcnt = stage + m;
  int flags[2] = {0, 0};
  GLX_HIP(hipMemcpyAsync(rcnt, b.rcnt, m * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  GLX_CHECK(!flags[0], GLX_EINVAL, "glx_knn_rows_to_csr: index out of range (a neighbour id, or a reverse entry outside the block)");
  std::vector<int64_t> roff(m + 1, 0);
  for (int64_t i = 0; i < m; ++i) roff[i + 1] = roff[i] + rcnt[i];
  GLX_UP(glx_upload(b.roff, roff.data(), (m + 1) * 8, st, __func__));
  if (nr > 0) {
    hipLaunchKernelGGL(fill_rev_rows_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, (const int64_t*)d_rrow, (const int64_t*)d_rsrc64,
                       (const int64_t*)d_rpos64, (const double*)b.dist, nr, row_base, (const int64_t*)b.roff, b.cursor, b.rsrc, b.rpos, b.rw);
    GLX_HIP(hipGetLastError());
  }
  const size_t shm = (size_t)4 * ROW_CAP * 16;
  GLX_HIP(hipFuncSetAttribute((const void*)merge_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const unsigned gr = (unsigned)((m + 3) / 4);
  // (one pass into a scratch image, compacted below: see merge_rows_kernel mode 2)
  GLX_POOL(glx_pool_alloc((void**)&b.tcol, std::max<size_t>((size_t)(m * k + n_rev) * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&b.tval, std::max<size_t>((size_t)(m * k + n_rev) * 8, 8)));
  std::vector<int> big;          // rows above 64 entries (popular vertices): the general kernel's, one wavefront each (outlives the asynchronous copy)
  {
    const bool small_rows = k <= 64;
    if (small_rows && sym != SYM_NONE)
      for (int64_t i = 0; i < m; ++i)
        if (k + rcnt[i] > 64) big.push_back((int)i);
    if (small_rows)
      hipLaunchKernelGGL(merge_rows_small_kernel, dim3(gr), dim3(256), 0, st, (const int64_t*)b.ind, (const double*)b.w, m, k, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, b.rowcnt,
                         b.tcol, b.tval, row_base);
    if (!small_rows) {
      hipLaunchKernelGGL(merge_rows_kernel, dim3(gr), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, m, k, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 2, b.rowcnt, (const int64_t*)nullptr,
                         b.tcol, b.tval, b.flag + 1, row_base);
    } else if (!big.empty()) {
      GLX_POOL(glx_pool_alloc((void**)&b.biglist, big.size() * 4));
      GLX_UP(glx_upload(b.biglist, big.data(), big.size() * 4, st, __func__));
      hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)((big.size() + 3) / 4)), dim3(256), shm, st, (const int64_t*)b.ind, (const double*)b.w, m, k, k,
                         (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 2, b.rowcnt, (const int64_t*)nullptr,
                         b.tcol, b.tval, b.flag + 1, row_base, 64, (const int*)b.biglist, (int)big.size());
    }
  }
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipMemcpyAsync(rowcnt, b.rowcnt, m * 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(flags, b.flag, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  int64_t nh = 0;
  if (flags[1]) {      // hub rows: a workgroup each (merge_hub_kernel)
    std::vector<int64_t> hrow, hoff(1, 0);
    for (int64_t i = 0; i < m; ++i) {
      if (rowcnt[i] >= 0) continue;
      const int64_t M = k + (roff[i + 1] - roff[i]);
      int64_t P = 2048;
      while (P < M) P <<= 1;
      hrow.push_back(i);
      hoff.push_back(hoff.back() + P);
    }
    nh = (int64_t)hrow.size();
    GLX_POOL(glx_pool_alloc((void**)&b.hub_row, nh * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.hub_off, (nh + 1) * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.hub_cnt, nh * 4));
    GLX_POOL(glx_pool_alloc((void**)&b.skey, (size_t)hoff.back() * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.sval, (size_t)hoff.back() * 8));
    GLX_UP(glx_upload(b.hub_row, hrow.data(), nh * 8, st, __func__));
    GLX_UP(glx_upload(b.hub_off, hoff.data(), (nh + 1) * 8, st, __func__));
    hipLaunchKernelGGL(merge_hub_kernel, dim3((unsigned)nh), dim3(1024), 0, st, (const int64_t*)b.ind, (const double*)b.w, k, k,
                       (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 0,
                       (const int64_t*)b.hub_row, (const int64_t*)b.hub_off, b.skey, b.sval, b.hub_cnt, (const int64_t*)nullptr,
                       (int*)nullptr, (double*)nullptr, row_base);
    GLX_HIP(hipGetLastError());
    std::vector<int> hcnt(nh);
    GLX_HIP(hipMemcpyAsync(hcnt.data(), b.hub_cnt, nh * 4, hipMemcpyDeviceToHost, st));
    GLX_HIP(hipStreamSynchronize(st));
    for (int64_t h = 0; h < nh; ++h) rowcnt[hrow[h]] = hcnt[h];
  }
  std::vector<int64_t> rp(m + 1, 0);
  for (int64_t i = 0; i < m; ++i) rp[i + 1] = rp[i] + rowcnt[i];
  const int64_t nnz = rp[m];
  GLX_CHECK(nnz < (1ll << 31), GLX_EUNSUPPORTED, "glx_knn_rows_to_csr: nnz %lld does not fit an int32 CSR", (long long)nnz);
  GLX_CHECK(nnz <= cap, GLX_EINVAL, "glx_knn_rows_to_csr: %lld entries, room for %lld", (long long)nnz, (long long)cap);
  GLX_UP(glx_upload(b.rowptr, rp.data(), (m + 1) * 8, st, __func__));
  GLX_POOL(glx_pool_alloc((void**)&b.col, std::max<size_t>(nnz * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&b.val, std::max<size_t>(nnz * 8, 8)));
  hipLaunchKernelGGL(compact_rows_kernel, dim3(gr), dim3(256), 0, st, (const int*)b.rowcnt, (const int64_t*)b.roff, m, k, sym, (const int64_t*)b.rowptr,
                     (const int*)b.tcol, (const double*)b.tval, b.col, b.val);
  GLX_HIP(hipGetLastError());
  if (nh) {
    hipLaunchKernelGGL(merge_hub_kernel, dim3((unsigned)nh), dim3(1024), 0, st, (const int64_t*)b.ind, (const double*)b.w, k, k,
                       (const int64_t*)b.roff, (const int*)b.rsrc, (const unsigned short*)b.rpos, (const double*)b.rw, sym, 1,
                       (const int64_t*)b.hub_row, (const int64_t*)b.hub_off, b.skey, b.sval, b.hub_cnt, (const int64_t*)b.rowptr,
                       b.col, b.val, row_base);
    GLX_HIP(hipGetLastError());
  }
  for (int64_t i = 0; i <= m; ++i) rowptr_out[i] = (int32_t)rp[i];
  GLX_UP(glx_download(col_out, b.col, nnz * 4, st, __func__));
  GLX_UP(glx_download(val_out, b.val, nnz * 8, st, __func__));
  GLX_HIP(hipStreamSynchronize(st));
  *nnz_out = nnz;
  return GLX_OK;
}

extern "C" int glx_knn_to_csr(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k, int kernel,
                              int sym, int32_t** rowptr_out, int32_t** col_out, double** val_out, int64_t* nnz_out, int device) {
  return knn_to_csr_impl(ind, dist, weights, n, kk, k, kernel, sym, -1, rowptr_out, col_out, val_out, nnz_out, device);
}

// the same with the CSR written into the caller's arrays (rowptr: n + 1, col / val: cap entries; 2 n k always suffices,
// n k without symmetrisation): page-locked ones make the copy back run at PCIe speed and save the copy out of
// library-owned memory
extern "C" int glx_knn_to_csr_into(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k, int kernel,
                                   int sym, int64_t cap, int32_t* rowptr, int32_t* col, double* val, int64_t* nnz_out, int device) {
  GLX_CHECK(cap >= 0 && rowptr && col && val, GLX_EINVAL, "glx_knn_to_csr_into: null buffer or negative capacity");
  return knn_to_csr_impl(ind, dist, weights, n, kk, k, kernel, sym, cap, &rowptr, &col, &val, nnz_out, device);
}

// the weight matrix of a search result (glx_knn_search): the lists never leave the device.  k <= the result's columns; weights
// (n, k) only for kernel = given.  Caller's buffers as in glx_knn_to_csr_into.
extern "C" int glx_knn_result_to_csr(const glx_knn_result* res, int k, int kernel, int sym, const double* weights, int64_t cap,
                                     int32_t* rowptr, int32_t* col, double* val, int64_t* nnz_out) {
  GLX_CHECK(res && res->ind && res->dist, GLX_EINVAL, "glx_knn_result_to_csr: empty result");
  GLX_CHECK(rowptr && col && val && cap >= 0, GLX_EINVAL, "glx_knn_result_to_csr: null buffer");
  int32_t* rp = rowptr;
  int32_t* ci = col;
  double* va = val;
  return knn_to_csr_impl(nullptr, nullptr, weights, res->n, res->k, k, kernel, sym, cap, &rp, &ci, &va, nnz_out, res->device, res);
}
