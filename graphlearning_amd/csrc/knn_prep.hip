// Exact kNN search, set-up kernels: centring (column means, the largest centred norm), the operand images of the two candidate
// filters, and the thresholds the seeding pre-pass hands to the search proper.  See knn.hip for the stages of the search
// (reference graphlearning/weightmatrix.py:297-429).
#include "knn_internal.h"

// ---- stage 0: centred fp32 images with the norms folded in ---------------------------------
// Rf[i] = [x_0..x_{d-1}, 0.., |x|^2, 1]   Qf[i] = [-2x_0..-2x_{d-1}, 0.., 1, |x|^2]   (dpa floats)
__global__ void knn_prep_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d, int dpa,
                                float* __restrict__ Rf, float* __restrict__ Qf, float* __restrict__ qnorm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float nrm = 0.f;
  for (int f = 0; f < d; ++f) {
    const float x = (float)(X[i * d + f] - mean[f]);
    Rf[i * dpa + f] = x;
    Qf[i * dpa + f] = -2.f * x;
    nrm = fmaf(x, x, nrm);
  }
  for (int f = d; f < dpa - 2; ++f) { Rf[i * dpa + f] = 0.f; Qf[i * dpa + f] = 0.f; }
  Rf[i * dpa + dpa - 2] = nrm;
  Rf[i * dpa + dpa - 1] = 1.f;
  Qf[i * dpa + dpa - 2] = 1.f;
  Qf[i * dpa + dpa - 1] = nrm;
  qnorm[i] = sqrtf(nrm);
}

// ---- split-bf16 operands (the bf16 filter, knn_tile_bf16.h) ---------------------------------------------------------------
__device__ __forceinline__ unsigned short f32_to_bf16_rn(float x) {
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// Xb[i] = [hi_0 .. hi_{kpad-1} | lo_0 .. lo_{kpad-1}] (bf16), nrm[i] = |x32|^2 (fp32), qnorm[i] = |x32|
__global__ void knn_prep_bf16_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d, int kpad,
                                     unsigned short* __restrict__ Xb, float* __restrict__ nrm, float* __restrict__ qnorm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + KNN_PAD_ROWS) return;
  // (eight features at a time: their hi and lo halves leave in one 16-byte store each -- two-byte stores made this kernel 2.3 ms at
  // 10^6 x 64)
  uint4* row_hi = (uint4*)(Xb + i * 2 * kpad);
  uint4* row_lo = (uint4*)(Xb + i * 2 * kpad + kpad);
  if (i >= n) {                       // spare rows behind the data: zero features, infinitely far
    const uint4 z = {0u, 0u, 0u, 0u};
    for (int u = 0; u < kpad / 8; ++u) { row_hi[u] = z; row_lo[u] = z; }
    nrm[i] = 1e30f;
    return;
  }
  float s = 0.f;
  for (int u = 0; u < kpad / 8; ++u) {
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = u * 8 + e;
      hi[e] = 0; lo[e] = 0;
      if (f < d) {
        const float x = (float)(X[i * d + f] - mean[f]);
        hi[e] = f32_to_bf16_rn(x);
        lo[e] = f32_to_bf16_rn(x - bf16_to_f32(hi[e]));
        s = fmaf(x, x, s);
      }
    }
    uint4 vh, vl;
    vh.x = hi[0] | ((unsigned)hi[1] << 16); vh.y = hi[2] | ((unsigned)hi[3] << 16); vh.z = hi[4] | ((unsigned)hi[5] << 16); vh.w = hi[6] | ((unsigned)hi[7] << 16);
    vl.x = lo[0] | ((unsigned)lo[1] << 16); vl.y = lo[2] | ((unsigned)lo[3] << 16); vl.z = lo[4] | ((unsigned)lo[5] << 16); vl.w = lo[6] | ((unsigned)lo[7] << 16);
    row_hi[u] = vh;
    row_lo[u] = vl;
  }
  nrm[i] = s;
  qnorm[i] = sqrtf(s);
}

// Concatenated split operands for d <= 21 (round 3): hi.hi + hi.lo + lo.hi is ONE contraction of length 3 d <= 63 when the
// ref image is [rh | rh | rl] and the query image [qh | ql | qh] -- 64 bf16 per row, the size of the [hi(32) | lo(32)] rows the
// NKB = 2 kernel stages, so four MFMAs of K = 16 do the work of the six the block form needs (hi and lo blocks padded to 32).
// fold (d <= 20: the slots 20, 41, 62 of the three segments are free): the ref image holds -2 x (exact) and, in the free slots,
// |x|^2 as three bf16 pieces against ones in the query image -- the contraction then IS the selection value |r|^2 - 2 q.r and the
// tile kernel needs neither the norms of the tile nor an fma per pair (measured by ablation: 11 % of the config-2 tile kernel)
__global__ void knn_prep_bf16_cat_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d,
                                         unsigned short* __restrict__ Xa, unsigned short* __restrict__ Xq, float* __restrict__ nrm,
                                         float* __restrict__ qnorm, int fold) {
  // (a row of each image is put together in registers and leaves in eight 16-byte stores: 128 two-byte stores per row and image
  // took 78 us at 70 000 rows)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + KNN_PAD_ROWS) return;
  unsigned short ra[64], rq[64];
#pragma unroll
  for (int f = 0; f < 64; ++f) { ra[f] = 0; rq[f] = 0; }
  if (i >= n) {                       // spare rows behind the data: zero features, infinitely far
    nrm[i] = 1e30f;
    if (fold) { ra[KNN_CAT_SEG - 1] = f32_to_bf16_rn(1e30f); rq[KNN_CAT_SEG - 1] = 0x3f80; }
  } else {
    float s = 0.f;
    const float sc = fold ? -2.f : 1.f;
#pragma unroll
    for (int f = 0; f < KNN_CAT_SEG; ++f) {
      if (f < d) {
        const float x = (float)(X[i * d + f] - mean[f]);
        const unsigned short hi = f32_to_bf16_rn(x);
        const unsigned short lo = f32_to_bf16_rn(x - bf16_to_f32(hi));
        const unsigned short shi = f32_to_bf16_rn(sc * bf16_to_f32(hi)), slo = f32_to_bf16_rn(sc * bf16_to_f32(lo));   // (exact: a power of two)
        ra[f] = shi; ra[KNN_CAT_SEG + f] = shi; ra[2 * KNN_CAT_SEG + f] = slo;
        rq[f] = hi; rq[KNN_CAT_SEG + f] = lo; rq[2 * KNN_CAT_SEG + f] = hi;
        s = fmaf(x, x, s);
      }
    }
    nrm[i] = s;
    qnorm[i] = sqrtf(s);
    if (fold) {
      const unsigned short n1 = f32_to_bf16_rn(s);
      const float r1 = s - bf16_to_f32(n1);
      const unsigned short n2 = f32_to_bf16_rn(r1);
      const unsigned short n3 = f32_to_bf16_rn(r1 - bf16_to_f32(n2));
      ra[KNN_CAT_SEG - 1] = n1; ra[2 * KNN_CAT_SEG - 1] = n2; ra[3 * KNN_CAT_SEG - 1] = n3;
      rq[KNN_CAT_SEG - 1] = 0x3f80; rq[2 * KNN_CAT_SEG - 1] = 0x3f80; rq[3 * KNN_CAT_SEG - 1] = 0x3f80;
    }
  }
  uint4* oa = (uint4*)(Xa + i * 64);
  uint4* oq = (uint4*)(Xq + i * 64);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    uint4 va, vq;
    va.x = ra[8 * u + 0] | ((unsigned)ra[8 * u + 1] << 16); va.y = ra[8 * u + 2] | ((unsigned)ra[8 * u + 3] << 16);
    va.z = ra[8 * u + 4] | ((unsigned)ra[8 * u + 5] << 16); va.w = ra[8 * u + 6] | ((unsigned)ra[8 * u + 7] << 16);
    vq.x = rq[8 * u + 0] | ((unsigned)rq[8 * u + 1] << 16); vq.y = rq[8 * u + 2] | ((unsigned)rq[8 * u + 3] << 16);
    vq.z = rq[8 * u + 4] | ((unsigned)rq[8 * u + 5] << 16); vq.w = rq[8 * u + 6] | ((unsigned)rq[8 * u + 7] << 16);
    oa[u] = va;
    oq[u] = vq;
  }
}

// ---- centring on the device: column means and the largest centred norm without a host round trip.
// Round 2's column-sum kernel gave each of d threads a 1024-long strided chain (20 of 256 threads active at d = 20: 0.39 ms
// for 11 MB at config 2, a quarter of the tile kernel); now the 256 threads of a workgroup tile its CENTRE_ROWS x d slab as
// (rows in flight) x (columns side by side), every thread sums its column over its rows in a register, LDS combines the row
// lanes in a fixed order, and one more small kernel folds the block partials -- in block order -- into the mean.  The largest centred norm is
// reduced on the device too; the re-rank kernel reads it from memory, the host looks at it (is the input finite?) together
// with the acceptance flags at the end.  Any FIXED summation order serves: the mean only centres the filter's operands,
// distances come from the uncentred fp64 data.
__global__ __launch_bounds__(256) void knn_colsum_kernel(const double* __restrict__ X, int64_t n, int d, int dt, double* __restrict__ part) {
  // thread = (row lane, column): dt = power of two >= min(d, 256) columns side by side, 256 / dt rows in flight; the lanes of a
  // row read consecutive elements, consecutive row lanes the next rows of the contiguous slab
  const int f0 = threadIdx.x % dt, rl = threadIdx.x / dt, rt = 256 / dt;
  const int64_t r0 = (int64_t)blockIdx.x * CENTRE_ROWS, r1 = min(n, r0 + CENTRE_ROWS);
  __shared__ double sm[256];
  for (int fb = 0; fb < d; fb += dt) {             // (one pass unless d > 256)
    const int f = fb + f0;
    double s = 0.0;
    if (f < d)
      for (int64_t i = r0 + rl; i < r1; i += rt) s += X[i * d + f];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && f < d) {
      double t = sm[f0];
      for (int q = 1; q < rt; ++q) t += sm[q * dt + f0];     // fixed order
      part[(size_t)blockIdx.x * d + f] = t;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void knn_mean_kernel(const double* __restrict__ part, int64_t nblk, int d, int64_t n, double* __restrict__ mean) {
  // thread = (column, one of 256 / dt runs of blocks), eight partial sums per thread whose loads do not wait for one another, the
  // runs combined in order: a fixed summation order (one dependent load + add per block was 35 us at 70 000 x 20)
  __shared__ double sm[256];
  int dt = 1;
  while (dt < d && dt < 256) dt *= 2;
  const int c = threadIdx.x % dt, part_id = threadIdx.x / dt, nparts = 256 / dt;
  for (int f0 = 0; f0 < d; f0 += dt) {
    const int f = f0 + c;
    double s = 0.0;
    if (f < d) {
      const int64_t per = (nblk + nparts - 1) / nparts;
      const int64_t b0 = part_id * per, b1 = min(nblk, b0 + per);
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int64_t b = b0;
      for (; b + 8 <= b1; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += part[(size_t)(b + q) * d + f];
      }
      for (int q = 0; b < b1; ++b, ++q) a[q] += part[(size_t)b * d + f];
      s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (part_id == 0 && f < d) {
      double t = 0.0;
      for (int q = 0; q < nparts; ++q) t += sm[q * dt + c];
      mean[f] = t / (double)n;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void knn_maxnorm_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d,
                                                          double* __restrict__ part) {
  // 256 rows per workgroup, sixteen lanes on a row (consecutive lanes on consecutive features: a thread walking its own row reads
  // one value per 64 cache lines and made this pass 0.94 ms at 10^6 x 64); the lanes' partial sums meet in lane 0 of the sixteen
  const int l16 = threadIdx.x & 15;
  double s = 0.0;
  for (int r = threadIdx.x >> 4; r < 256; r += 16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + r;
    double t = 0.0;
    if (i < n)
      for (int f = l16; f < d; f += 16) { const double c = X[i * d + f] - mean[f]; t += c * c; }
    t += __shfl_xor(t, 1, 16);
    t += __shfl_xor(t, 2, 16);
    t += __shfl_xor(t, 4, 16);
    t += __shfl_xor(t, 8, 16);
    if (!(t == t)) t = INFINITY;   // NaN input: reported as non-finite
    s = t > s ? t : s;
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off && sm[threadIdx.x + off] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
// rmax_out[0] = sqrt(max) * (1 + 1e-6) as a float (what the re-rank's acceptance bound uses), [1] = 1 if the input is finite
__global__ __launch_bounds__(256) void knn_rmax_kernel(const double* __restrict__ part, int64_t nblk, float* __restrict__ rmax_out) {
  double m = 0.0;
  for (int64_t b = threadIdx.x; b < nblk; b += 256) m = part[b] > m ? part[b] : m;
  __shared__ double sm[256];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off && sm[threadIdx.x + off] > sm[threadIdx.x]) sm[threadIdx.x] = sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double r2 = sm[0];
    rmax_out[0] = (float)(sqrt(r2) * (1.0 + 1e-6));
    rmax_out[1] = (r2 == r2 && r2 < INFINITY) ? 1.0f : 0.0f;
  }
}

// ---- seeding: a threshold for every query BEFORE the search proper -----------------------------
// The pre-pass ran the tile kernel over a sample of the refs (every 8th tile, say; one range: two lists per query).  Any k
// distinct refs bound the k-th neighbour from above: with v_k = the k-th smallest filter value among the sample's candidates,
// true dist^2 of those k refs <= v_k + eps, hence the exact k-th distance^2 dk2 <= v_k + eps.  The search proper starts every
// list's threshold at seed = v_k + 4 eps (instead of +inf): a ref it rejects has filter value >= seed, i.e. true dist^2 >=
// v_k + 3 eps > dk2 -- never one of the k nearest (nor tied with the k-th).  What it buys: the lists only ever see refs within a few
// percent of the k-th distance (in d dimensions a sample of 1/8 is (8)^(1/d) further out), a tenth of the appends and merges of
// lists that start empty; list maintenance was 41-62 % of the tile kernel.  Values here carry the query's norm (cand_d does).
template <int M>       // M = 2 KP candidates per query (16 / 32 / 64): they wait in registers (read from memory inside the double loop the
                       // kernel took 4.8 ms at 10^6 queries)
__global__ __launch_bounds__(256) void knn_seed_kernel(const float* __restrict__ pre_d, const int* __restrict__ pre_i, int64_t nq, int64_t q_begin,
                                                       int k, const float* __restrict__ qnorm, const float* __restrict__ nrm,
                                                       const float* __restrict__ rmax_p, double cerr, int* __restrict__ gtau,
                                                       double* __restrict__ ub2) {
  const int64_t ql = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ql >= nq) return;
  float v[M];
  {
    const float4* pv = (const float4*)(pre_d + ql * M);
    const int4* pi = (const int4*)(pre_i + ql * M);
#pragma unroll
    for (int u = 0; u < M / 4; ++u) {
      const float4 a = pv[u];
      const int4 b = pi[u];
      v[4 * u + 0] = b.x >= 0 ? a.x : INFINITY;    // (an empty slot counts as +inf: never among the k smallest)
      v[4 * u + 1] = b.y >= 0 ? a.y : INFINITY;
      v[4 * u + 2] = b.z >= 0 ? a.z : INFINITY;
      v[4 * u + 3] = b.w >= 0 ? a.w : INFINITY;
    }
  }
  float vk = INFINITY;
#pragma unroll
  for (int a = 0; a < M; ++a) {                 // the k-th smallest of M values: the one with exactly k - 1 in front of it
    const float x = v[a];
    int before = 0;
#pragma unroll
    for (int c = 0; c < M; ++c) before += (v[c] < x || (v[c] == x && c < a)) ? 1 : 0;
    if (x < INFINITY && before == k - 1) vk = x;
  }
  int key = 0x7f800000;                          // +inf: fewer than k candidates in the sample
  if (vk < INFINITY) {
    const double rq = (double)qnorm[q_begin + ql] + (double)rmax_p[0];
    const double eps = cerr * rq * rq;
    // back to the kernel's form (without the query's norm: nrm holds |q|^2 as the tile kernel adds it), rounded up
    double x = (double)vk + 4.0 * eps;
    x += 1e-6 * fabs(x);
    float seed = (float)(x - (double)nrm[q_begin + ql]);
    seed = nextafterf(seed, INFINITY);
    key = __float_as_int(seed);
    key ^= (key >> 31) & 0x7fffffff;
    if (ub2) ub2[ql] = (double)vk + eps;        // exact k-th distance^2 <= this
  } else if (ub2) {
    ub2[ql] = INFINITY;
  }
  gtau[ql] = key;
}

int knn_launch_seed(int KP, const KnnBufs& b, int64_t nq, int64_t q0, int k, double cerr, hipStream_t st) {
  const dim3 sg((unsigned)((nq + 255) / 256));
#define GLX_SEED(MM)                                                                                                                  \
  hipLaunchKernelGGL(knn_seed_kernel<MM>, sg, dim3(256), 0, st, (const float*)b.pre_d, (const int*)b.pre_i, nq, q0, k, (const float*)b.qnorm, \
                     (const float*)b.nrm, (const float*)b.rmax, cerr, b.gtau, b.ub2)
  if (KP == 8) GLX_SEED(16);
  else if (KP == 16) GLX_SEED(32);
  else GLX_SEED(64);
#undef GLX_SEED
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}
