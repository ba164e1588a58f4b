// Exact k-nearest-neighbour search on the MI355X: weightmatrix.knnsearch of the reference
// (graphlearning/weightmatrix.py:297-429; kdtree branch :349-352 is the exact answer we
// reproduce).  Three stages:
//   1. candidate filter -- brute-force tiled pairwise squared distances as an (n x d) @ (d x n) contraction on the matrix
//      cores, refs staged through LDS: split-bf16 operands (knn_tile_bf16.h; d <= 128) or fp32 operands with the norms folded
//      in as two extra features (knn_tile_f32.h); every lane owns one query column and keeps the KP best of its half of the
//      refs of its range in a list (unsorted, maximum tracked);
//   2. exact re-rank (knn_rerank.hip) -- fp64 direct-difference distances (the accumulation pattern of scipy cKDTree's
//      sqeuclidean_distance_double) of the candidates, sorted by (distance, index); a row is accepted only if every candidate
//      list's threshold exceeds the exact k-th distance by twice a bound on the filter's error;
//   3. fallback -- rows that fail the check are redone by an exact fp64 scan.
// This file: one pass of the search (knn_pass), the escalation to long lists, the C-ABI entry points.
#include "knn_internal.h"
#include <chrono>
#include <stdlib.h>
#include <unistd.h>
#include <vector>

// statistics of the calling thread's last search (glx_knn_stats)
static thread_local double g_knn_stats[16];
extern "C" int glx_knn_stats(double stats[16]) {
  GLX_CHECK(stats, GLX_EINVAL, "glx_knn_stats: null output");
  for (int i = 0; i < 16; ++i) stats[i] = g_knn_stats[i];
  return GLX_OK;
}

// the calling thread's plan overrides (glx_knn_set_options; all zero / -1 = the library decides)
static thread_local glx_knn_options g_knn_opt = {0, 0, 0, -1};
extern "C" int glx_knn_set_options(const glx_knn_options* opt) {
  if (!opt) { g_knn_opt = {0, 0, 0, -1}; return GLX_OK; }
  GLX_CHECK(opt->filter >= 0 && opt->filter <= 2 && opt->lists >= 0 && opt->lists <= 2 && opt->nsplit >= 0 && opt->nsplit <= 8 &&
            opt->concat >= -1 && opt->concat <= 2, GLX_EINVAL, "glx_knn_set_options: value out of range");
  g_knn_opt = *opt;
  return GLX_OK;
}

// ---- debugging aid: is the device copy of X the caller's X? ------------------------------------------------------------------------
// glx_debug_set(flags): bit 0 (1) = after the upload of a search's features the device copy is read back TWICE -- by the copy engine, and through a
// kernel (i.e. through the L2s) -- and compared with the caller's array; differences are counted (glx_debug_counters) and described
// on stderr.  Round 6: the one parity failure of the randomised soak that left evidence was a search whose device copy of ONE row of X
// was not the caller's (EXPERIMENTS.md round 6, section 2).
static int g_debug_flags = 0;
static unsigned long long g_debug_counts[4] = {0, 0, 0, 0};   // uploads checked, uploads whose engine read-back differed, whose kernel read-back differed, bytes differing
extern "C" int glx_debug_set(int flags) { g_debug_flags = flags; return GLX_OK; }
extern "C" int glx_debug_counters(unsigned long long out[4]) {
  GLX_CHECK(out, GLX_EINVAL, "glx_debug_counters: null output");
  for (int q = 0; q < 4; ++q) out[q] = g_debug_counts[q];
  return GLX_OK;
}
__global__ __launch_bounds__(256) void knn_copy_u64_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
static int knn_verify_upload(const double* X_host, const double* X_dev, int64_t n, int d, hipStream_t st, const char* what) {
  const size_t bytes = (size_t)n * d * 8;
  GLX_HIP(hipStreamSynchronize(st));
  // (read back INTO PAGE-LOCKED MEMORY: a copy into pageable memory can show the very holes this check looks for)
  unsigned long long* back = nullptr;
  GLX_HIP(hipHostMalloc((void**)&back, std::max<size_t>(bytes, 64), hipHostMallocDefault));
  struct Free { unsigned long long* p; ~Free() { hipHostFree(p); } } free_back{back};
  ++g_debug_counts[0];
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) {
      GLX_HIP(hipMemcpy(back, X_dev, bytes, hipMemcpyDeviceToHost));
    } else {
      void* tmp = nullptr;
      GLX_HIP(hipMalloc(&tmp, bytes));                  // (a fresh allocation, not a pooled block)
      hipLaunchKernelGGL(knn_copy_u64_kernel, dim3(1024), dim3(256), 0, st, (const unsigned long long*)X_dev, (unsigned long long*)tmp, (int64_t)(bytes / 8));
      hipError_t e = hipStreamSynchronize(st);
      if (e == hipSuccess) e = hipMemcpy(back, tmp, bytes, hipMemcpyDeviceToHost);
      hipFree(tmp);
      GLX_HIP(e);
    }
    const unsigned long long* src = (const unsigned long long*)X_host;
    size_t nbad = 0, first = 0, last = 0;
    for (size_t i = 0; i < bytes / 8; ++i)
      if (back[i] != src[i]) { if (!nbad) first = i; last = i; ++nbad; }
    if (nbad) {
      ++g_debug_counts[1 + pass];
      g_debug_counts[3] += nbad * 8;
      fprintf(stderr, "[glx] knn DEBUG (%s, pid %d): the device copy of X read back by %s differs from the caller's array in %zu of %zu words: words %zu .. %zu "
                      "(rows %zu .. %zu of %lld, d = %d; byte offsets %zu .. %zu; device address %p; source address %p)\n", what, (int)getpid(),
              pass == 0 ? "the copy engine" : "a kernel (through the L2s)", nbad, bytes / 8, first, last, first / d, last / d, (long long)n, d, first * 8, last * 8 + 7,
              (const void*)X_dev, (const void*)X_host);
      if (pass == 0) {
        // what the wrong words hold: zeros, words of the SAME array from another place (a shifted or repeated piece), or nothing of it
        size_t zeros = 0, shown = 0;
        for (size_t i = first; i <= last; ++i) {
          if (back[i] == src[i]) continue;
          if (back[i] == 0) { ++zeros; continue; }
          if (shown < 6) {
            long long at = -1;
            for (size_t j = 0; j < bytes / 8; ++j)
              if (src[j] == back[i]) { at = (long long)j; break; }
            fprintf(stderr, "[glx] knn DEBUG   word %zu: got %016llx (as a double %.6g), expected %016llx (%.6g); the value got %s%lld\n", i, back[i],
                    __builtin_bit_cast(double, back[i]), src[i], __builtin_bit_cast(double, src[i]),
                    at >= 0 ? "is word " : "occurs nowhere in the caller's array ", at);
            ++shown;
          }
        }
        fprintf(stderr, "[glx] knn DEBUG   %zu of the wrong words are zero\n", zeros);
      }
    }
  }
  return GLX_OK;
}

static const int KNN_ESCALATE = 1;    // knn_pass: too many rows failed the acceptance test of the short lists -- search again with long ones

// One pass of the search.  long_lists = false: the default (short lists where they apply); if then so many query rows fail
// the acceptance test that repairing them row by row -- each streams the whole data set -- would take longer than
// searching again, KNN_ESCALATE is returned: the caller repeats the search with the long lists (one list
// holds all k neighbours of a query, whatever their arrangement in the data).  It takes data whose k nearest neighbours
// sit in the same 16 of 32 consecutive points to get there (tight groups stored one after another); interleaving the ref tiles
// over the ranges already spreads anything coarser.
// capture (full searches: glx_knn_search): the lists stay on the device with this result object, together with the cell order
// the pass worked out (if it did), instead of being copied out.
static int knn_pass(const double* X, int64_t n, int d, int k, int64_t q0, int64_t q1, int64_t* ind_out, double* dist_out, int device,
                    bool long_lists, glx_knn_result* capture, const int64_t* cell_starts = nullptr, int ncells = 0, int auto_cells = 0) {
  GLX_CHECK(X && ((ind_out && dist_out) || capture), GLX_EINVAL, "glx_knn_bruteforce: null argument");
  GLX_CHECK(!capture || (q0 == 0 && q1 == n), GLX_EINVAL, "glx_knn_search: a result object holds a full search");
  GLX_CHECK(n >= 1 && d >= 1 && k >= 1, GLX_EINVAL, "glx_knn_bruteforce: need n, d, k >= 1 (n=%lld d=%d k=%d)", (long long)n, d, k);
  GLX_CHECK(k <= n, GLX_EINVAL, "glx_knn_bruteforce: k=%d exceeds the number of points %lld", k, (long long)n);
  GLX_CHECK(n < (1ll << 31) - BR_MAX, GLX_EINVAL, "glx_knn_bruteforce: n must fit int32");
  GLX_CHECK(0 <= q0 && q0 <= q1 && q1 <= n, GLX_EINVAL, "glx_knn_bruteforce: bad query range");
  GLX_CHECK(k <= 60, GLX_EUNSUPPORTED, "glx_knn_bruteforce: k=%d (incl. self) above the supported 60", k);
  GLX_CHECK(d <= 16382, GLX_EUNSUPPORTED, "glx_knn_bruteforce: d=%d above the supported 16382", d);
  const int64_t nq = q1 - q0;
  if (nq == 0) return GLX_OK;
  const bool timing = getenv("GLX_TIMING") != nullptr;
  const auto t_host0 = std::chrono::steady_clock::now();
  auto stamp = [&](const char* what) {
    if (timing) fprintf(stderr, "[glx] knn: %-28s %.2f ms since the call\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count());
  };
  GLX_HIP(hipSetDevice(device));
  // d + 2 <= 132: the query's features stay in registers; above that the feature dimension is blocked
  int KP = k <= 12 ? 16 : (k <= 28 ? 32 : 64);
  // Short lists.  A query's candidates are kept in 2*nsplit separate lists (two half-wavefronts x
  // nsplit ref ranges); with >= 8 lists, 8 entries per list hold the k <= 12 nearest unless 8 of
  // them fall into the same list (5e-5 per query for k = 11; the acceptance test of the re-rank
  // sees a full list whose threshold is too small and sends the row to the exact fallback).  The
  // shorter lists free LDS for a third workgroup per CU and halve the list rescans: 5.1 -> 3.6 ms
  // at config 2, 94 -> 108 TFLOP/s at d = 64.  Not for the blocked variant: at large d the fp32
  // error margin of the acceptance test makes short lists fall back too often.
  // The same argument one size up: 16 entries for k <= 28 (3e-7 per query at k = 28), 32 for k <= 60.
  const bool short_lists = !long_lists && d + 2 <= 132 && g_knn_opt.lists != 2;
  if (short_lists) KP = k <= 12 ? 8 : (k <= 28 ? 16 : 32);
  int DH = knn_kb(KP), nkb = 1;
  if (d + 2 <= 132 && !(KP == 64 && d + 2 > 36)) {   // (KP = 64 lists + a wide double-buffered tile exceed the LDS)
    for (int cand : {8, 12, 18, 34, 66})
      if (2 * cand >= d + 2) { DH = cand; break; }
  } else {
    nkb = (d + 2 + 2 * DH - 1) / (2 * DH);
  }
  // Filter arithmetic.  Default: split-bf16 operands on the bf16 matrix cores (d <= 128 with the short lists); the fp32-input
  // MFMA kernel serves everything else (and glx_knn_options::filter = 2).
  const bool use_bf16 = short_lists && d <= 128 && KP <= 32 && g_knn_opt.filter != 2;
  int NKB = 0;
  if (use_bf16) {
    for (int cand : {1, 2, 4, 6, 8})
      if (16 * cand >= d) { NKB = cand; break; }
  }
  const int dpa = use_bf16 ? 16 * NKB : 2 * DH * nkb;
  const int64_t nqb = (nq + BQ - 1) / BQ;
  const int BR = use_bf16 ? 32 * bf16_nsub(NKB, KP) : 32 * tile_nsub(DH, KP);
  const int64_t ntiles = (n + BR - 1) / BR;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, ntiles), (1024 + nqb - 1) / nqb));
  if (short_lists) {
    // >= 8 lists per query; 16 for the 8-entry lists once the data no longer sits in cache (an exact
    // fallback row then streams all of X k times: 329 rows cost 0.6 s at n = 2e6 -- with 16 lists 5 rows are left)
    const int64_t want = (KP == 8 && (double)n * d * 8.0 > 64.0 * 1024 * 1024) ? 8 : 4;
    nsplit = (int)std::max<int64_t>(nsplit, std::min<int64_t>(want, ntiles));
  }
  if (g_knn_opt.nsplit > 0) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, ntiles), g_knn_opt.nsplit));
  const int lists = nsplit * 2;
  const int ncand = lists * KP;
  int M = 64;
  while (M < ncand) M *= 2;

  // (host buffers of the cell order: declared in front of `b`, whose destructor drains the stream they are filled through)
  std::vector<int> oc_sample, oc_cid, oc_perm, oc_place;
  std::vector<double> oc_cen;
  KnnBufs b;
  {
    int rcw = glx_work_acquire(device, &b.work);
    if (rcw) return rcw;
  }
  b.stream = b.work->stream;
  b.e0 = b.work->ev[0]; b.e1 = b.work->ev[1]; b.e2 = b.work->ev[2]; b.e3 = b.work->ev[3];
  hipStream_t st = b.stream;
  GLX_POOL(glx_pool_alloc((void**)&b.X, (size_t)n * d * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.mean, d * 8));
  stamp("stream, events, buffers");
  int rc0 = GLX_OK;
  // (hipMemcpyDefault: X may also be a DEVICE pointer -- glx_knn_bruteforce_range / glx_knn_cells_range of the sharded build, whose
  // features are generated, ordered and kept on the GPU; the library-formed cells below read sample rows on the host and need a host X)
  bool on_host = true;
  {
    hipPointerAttribute_t at;
    on_host = hipPointerGetAttributes(&at, X) != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged);
    (void)hipGetLastError();
  }
  if (on_host) {
    // the caller's (pageable) array goes up through the library's page-locked staging area, checked (glx_internal.h: why)
    rc0 = glx_upload(b.X, X, (size_t)n * d * 8, st, "features of a search");
    if (rc0) return rc0;
  } else {
    GLX_HIP(hipMemcpyAsync(b.X, X, (size_t)n * d * 8, hipMemcpyDefault, st));
  }
  stamp("X enqueued");
  if (g_debug_flags & 1) {
    if (on_host) {
      const int rcv = knn_verify_upload(X, b.X, n, d, st, "after the upload");
      if (rcv) return rcv;
    }
  }
  // Cells formed by the library (auto_cells).  > 1: that many cells (nearest of evenly spaced sample rows), the rows reordered by
  // cell and searched with the cell pruning of glx_knn_cells_range; the re-rank ranks by and returns the caller's indices.
  // < -1 (below the size where pruning pays): the rows ARE reordered by -auto_cells chained cells on the device and then searched
  // all pairs.  The 32 queries of a wavefront then come from one corner of feature space, a ref tile holds candidates for many of
  // them or for none, and fewer wave-tiles leave the tile kernel's fast path: config 2 2.06 -> 1.83 ms of search wall time, config
  // 3's shape 3.06 -> 2.63 ms, data without clusters unchanged (profiles/r03_knn_cells_midsize.txt).
  std::vector<int64_t> own_starts;
  const bool reorder_only = auto_cells < -1;
  if (auto_cells < -1) auto_cells = -auto_cells;
  // the cells in a chain of nearest centres (greedy, from the centre farthest from the centres' mean): neighbouring cells of
  // feature space end up next to each other in the row order, which then serves as a locality order for the graph's operators
  // too (one XCD's share of the rows = a few whole clusters; with the cells in arbitrary order the sweep at 10^6 rows ran 20 % slower);
  // then the rows by cell (counting sort, ascending caller index inside a cell).  Host work on oc_cid / oc_cen.
  auto chain_places = [&](const std::vector<double>& cen, int m) -> std::vector<int> {
    std::vector<double> mean(d, 0.0);
    for (int c = 0; c < m; ++c)
      for (int f = 0; f < d; ++f) mean[f] += cen[(size_t)c * d + f] / m;
    const int cfs = (d + 31) / 32;       // (every cfs-th feature, as in the assignment: m^2 d flops on one host thread otherwise)
    auto dist2 = [&](const double* a, const double* bb) { double t = 0; for (int f = 0; f < d; f += cfs) { const double q = a[f] - bb[f]; t += q * q; } return t; };
    int cur = 0;
    double far = -1.0;
    for (int c = 0; c < m; ++c) { const double t = dist2(&cen[(size_t)c * d], mean.data()); if (t > far) { far = t; cur = c; } }
    std::vector<int> place(m, -1);
    for (int pos = 0; pos < m; ++pos) {
      place[cur] = pos;
      int nxt = -1;
      double best = INFINITY;
      for (int c = 0; c < m; ++c)
        if (place[c] < 0) { const double t = dist2(&cen[(size_t)c * d], &cen[(size_t)cur * d]); if (t < best) { best = t; nxt = c; } }
      if (nxt < 0) break;
      cur = nxt;
    }
    return place;
  };
  bool perm_pending = false;
  if (auto_cells > 1 && q0 == 0 && q1 == n && !long_lists && d <= 128 && n >= 4 * (int64_t)auto_cells) {
    const int m = auto_cells;
    oc_sample.resize(m);
    for (int c = 0; c < m; ++c) oc_sample[c] = (int)(((2 * (int64_t)c + 1) * n) / (2 * (int64_t)m));     // evenly spaced rows
    GLX_POOL(glx_pool_alloc((void**)&b.cen, (size_t)m * d * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.cell_id, (size_t)std::max<int64_t>(n, m) * 4));
    GLX_UP(glx_upload(b.cell_id, oc_sample.data(), (size_t)m * 4, st, __func__));
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)(((int64_t)m * d + 255) / 256)), dim3(256), 0, st, (const double*)b.X, (const int*)b.cell_id,
                       (int64_t)m, d, b.cen);
    hipLaunchKernelGGL(knn_assign_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), (size_t)16 * d * 8, st, (const double*)b.X, d, n, (const double*)b.cen, m,
                       b.cell_id, (d + 31) / 32);
    GLX_HIP(hipGetLastError());
    GLX_POOL(glx_pool_alloc((void**)&b.orig, (size_t)n * 4));
    if (reorder_only) {
      // the chain of the cells from the caller's copy of the sample rows (the same doubles the device gathered), the rows into cell
      // order by the three knn_cellrank kernels: nothing here waits for the device
      oc_cen.resize((size_t)m * d);
      for (int c = 0; c < m; ++c) memcpy(&oc_cen[(size_t)c * d], X + (size_t)oc_sample[c] * d, (size_t)d * 8);
      oc_place = chain_places(oc_cen, m);
      const int nb = (int)((n + 255) / 256);
      GLX_POOL(glx_pool_alloc((void**)&b.place, (size_t)m * 4));
      GLX_POOL(glx_pool_alloc((void**)&b.bh, (size_t)(nb + 1) * m * 4));     // (+ one row: the keys' totals / starting positions)
      GLX_UP(glx_upload(b.place, oc_place.data(), (size_t)m * 4, st, __func__));
      hipLaunchKernelGGL(knn_cellrank_hist_kernel, dim3((unsigned)nb), dim3(256), (size_t)m * 4, st, b.cell_id, (const int*)b.place, n, m, b.bh);
      hipLaunchKernelGGL(knn_cellrank_scan_kernel, dim3((unsigned)m), dim3(256), 0, st, b.bh, nb, m);
      hipLaunchKernelGGL(knn_cellrank_base_kernel, dim3(1), dim3(256), 0, st, b.bh, nb, m);
      hipLaunchKernelGGL(knn_cellrank_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const int*)b.cell_id, n, m, (const int*)b.bh, nb, b.orig);
      perm_pending = capture != nullptr;               // the permutation comes back with the results (glx_knn_result_order)
    } else {
      // the pruned search needs the cells' extents on the host: cell ids and centres come back, the rows are counted into chained
      // cells here (stable: ascending caller index inside a cell)
      oc_cid.resize(n);
      oc_cen.resize((size_t)m * d);
      GLX_UP(glx_download(oc_cid.data(), b.cell_id, (size_t)n * 4, st, __func__));
      GLX_UP(glx_download(oc_cen.data(), b.cen, (size_t)m * d * 8, st, __func__));
      GLX_HIP(hipStreamSynchronize(st));
      oc_place = chain_places(oc_cen, m);
      std::vector<int64_t> fill(m + 1, 0);
      for (int64_t i = 0; i < n; ++i) { oc_cid[i] = oc_place[oc_cid[i]]; ++fill[oc_cid[i] + 1]; }
      for (int c = 0; c < m; ++c) fill[c + 1] += fill[c];
      own_starts.assign(fill.begin(), fill.begin() + m);
      oc_perm.resize(n);
      for (int64_t i = 0; i < n; ++i) oc_perm[fill[oc_cid[i]]++] = (int)i;
      // (no synchronisation behind the upload: oc_perm outlives the stream's work -- see its declaration)
      GLX_UP(glx_upload(b.orig, oc_perm.data(), (size_t)n * 4, st, __func__));
      if (capture) capture->order.assign(oc_perm.begin(), oc_perm.end());
      glx_pool_free(b.cen);                          // the cell pass allocates its own
      b.cen = nullptr;
      cell_starts = own_starts.data();
      ncells = m;
    }
    b.Xraw = b.X;
    b.X = nullptr;
    GLX_POOL(glx_pool_alloc((void**)&b.X, (size_t)n * d * 8));
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, st, (const double*)b.Xraw, (const int*)b.orig, n, d, b.X);
    GLX_HIP(hipGetLastError());
    stamp("rows reordered by cell");
  }
  // centring in fp64 (distances are translation invariant; small norms keep the filter sharp), all of it on the device
  const int64_t nb_sum = (n + CENTRE_ROWS - 1) / CENTRE_ROWS, nb_max = (n + 255) / 256;
  GLX_POOL(glx_pool_alloc((void**)&b.part, (size_t)std::max<int64_t>(nb_sum * d, nb_max) * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.rmax, 64));
  {
    int dt = 1;
    while (dt < d && dt < 256) dt *= 2;
    hipLaunchKernelGGL(knn_colsum_kernel, dim3((unsigned)nb_sum), dim3(256), 0, st, (const double*)b.X, n, d, dt, b.part);
    GLX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(knn_mean_kernel, dim3(1), dim3(256), 0, st, (const double*)b.part, nb_sum, d, n, b.mean);
  hipLaunchKernelGGL(knn_maxnorm_kernel, dim3((unsigned)nb_max), dim3(256), 0, st, (const double*)b.X, (const double*)b.mean, n, d, b.part);
  hipLaunchKernelGGL(knn_rmax_kernel, dim3(1), dim3(256), 0, st, (const double*)b.part, nb_max, b.rmax);
  GLX_HIP(hipGetLastError());
  // |filter value - exact dist^2| <= cerr * (|q| + rmax)^2.
  // fp32 filter: input rounding (2^-24 per coordinate), dpa products and sums at 2^-24 each, norms computed in fp32; generous constant.
  // bf16 filter: eps = cerr (|q| + rmax)^2 with cerr ~ 2^-16.  The dropped parts of the split products (lo.lo and the residuals of the
  // two roundings) are <= 3.1 * 2^-16 |q||r| in q.r in the worst case -- every coordinate's errors at their bounds and aligned --, twice
  // that in the distance, i.e. <= 1.55 * 2^-16 (|q| + rmax)^2; plus 3 kpad fp32 accumulations, fp32 norms and input rounding (the second
  // term, doubled: the matrix pipe's internal rounding mode is not documented).  So |filter - exact| < 2 eps ALWAYS, which is what the
  // acceptance test of the re-rank needs (it asks for a margin of 2 eps), and <= 0.52 eps on every pair of the randomised suite's inputs
  // (an emulation of the split arithmetic: profiles/r05_knn_tile_pmc.txt); the re-rank's fp32 screen allows for 2 eps per value as well.
  const double cerr = use_bf16 ? 2.0 * (std::ldexp(1.0, -17) + (1.5 * (3.0 * dpa + 4.0) + d + 16.0) * std::ldexp(1.0, -24))
                               : (double)(dpa + 8) * std::ldexp(1.0, -22);
  GLX_POOL(glx_pool_alloc((void**)&b.qnorm, (size_t)n * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.cand_d, (size_t)nq * ncand * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.cand_i, (size_t)nq * ncand * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.flags, (size_t)nq * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.dk2, (size_t)nq * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.nbad, 4));
  GLX_HIP(hipMemsetAsync(b.nbad, 0, 4, st));
  GLX_POOL(glx_pool_alloc((void**)&b.gtau, (size_t)nq * 4));
  GLX_HIP(hipMemsetD32Async((hipDeviceptr_t)b.gtau, 0x7f800000, (size_t)nq, st));   // +inf: nothing published yet
  GLX_POOL(glx_pool_alloc((void**)&b.rows, (size_t)nq * 4));
  GLX_POOL(glx_pool_alloc((void**)&b.ind, (size_t)nq * k * 8));
  GLX_POOL(glx_pool_alloc((void**)&b.dist, (size_t)nq * k * 8));
  stamp("centred, norms bounded");
  GLX_HIP(hipEventRecord(b.e0, st));
  int rc;
  g_knn_stats[9] = 0.0;      // (the fp32 filter has no concatenated form: not the previous search's value)
  if (use_bf16) {
    // 17 <= d <= 21 (two blocks of 16 per half): the three split products as ONE contraction over concatenated operands,
    // 4 MFMAs per 32 x 32 tile instead of 6 (d <= 16 needs 3 either way)
    // ... and for d <= 20 with the norm folded in (glx_knn_options::concat = 1: without the fold, 0: blocks of 16 features)
    int cat = (d <= KNN_CAT_SEG && NKB == 2) ? (d < KNN_CAT_SEG ? 2 : 1) : 0;
    if (g_knn_opt.concat >= 0) cat = std::min(cat, g_knn_opt.concat);
    GLX_POOL(glx_pool_alloc((void**)&b.Xb, (size_t)(n + KNN_PAD_ROWS) * 2 * dpa * 2));
    GLX_POOL(glx_pool_alloc((void**)&b.nrm, (size_t)(n + KNN_PAD_ROWS) * 4));
    if (cat) {
      GLX_POOL(glx_pool_alloc((void**)&b.Xq, (size_t)(n + KNN_PAD_ROWS) * 64 * 2));
      hipLaunchKernelGGL(knn_prep_bf16_cat_kernel, dim3((unsigned)((n + KNN_PAD_ROWS + 255) / 256)), dim3(256), 0, st, (const double*)b.X,
                         (const double*)b.mean, n, d, b.Xb, b.Xq, b.nrm, b.qnorm, cat == 2 ? 1 : 0);
    } else {
      hipLaunchKernelGGL(knn_prep_bf16_kernel, dim3((unsigned)((n + KNN_PAD_ROWS + 255) / 256)), dim3(256), 0, st, (const double*)b.X, (const double*)b.mean,
                         n, d, dpa, b.Xb, b.nrm, b.qnorm);
    }
    GLX_HIP(hipGetLastError());
    g_knn_stats[9] = (double)cat;
    // The seeding pre-pass (knn_seed_kernel) runs the tile kernel over a sample of the refs first and starts every list of the
    // search proper at a threshold derived from it.  Over all refs it does not pay (measured, profiles/r03_knn_seed.txt: the k-th
    // of a 1/8 sample is the 8k-th of the whole set, 79 % of the wave-tiles still hold a candidate and the pre-pass costs its
    // eighth); the cell-pruned search needs it: its bound ub2 decides which cells a query block visits.
    const bool cells = cell_starts != nullptr && ncells > 1;
    // sample the block's own cells: every tile of small cells, every 8th of cells of >= 128 tiles
    const int seed_sub = cells ? (int)std::max<int64_t>(1, std::min<int64_t>(8, std::max<int64_t>(1, ntiles / ncells) / 16)) : 0;
    const bool seeded = cells && 2 * KP >= k;
    g_knn_stats[10] = seeded ? (double)seed_sub : 0.0;
    g_knn_stats[11] = 0.0;
    g_knn_stats[12] = 0.0;
    if (seeded) {
      GLX_POOL(glx_pool_alloc((void**)&b.cell_starts, (size_t)ncells * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.cen, (size_t)ncells * d * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.rad, (size_t)ncells * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.ub2, (size_t)nq * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.mask, (size_t)nqb * ncells));
      GLX_POOL(glx_pool_alloc((void**)&b.nruns, (size_t)nqb * 4));
      b.maxruns = ncells;
      GLX_POOL(glx_pool_alloc((void**)&b.runs, (size_t)nqb * 2 * ncells * 4));   // (from here on the tile launches follow the runs)
      GLX_UP(glx_upload(b.cell_starts, cell_starts, (size_t)ncells * 8, st, __func__));
      // centres and radii of the cells
      GLX_POOL(glx_pool_alloc((void**)&b.cpart, (size_t)ncells * CELL_SPLIT * (d + 1) * 8));
      double* prad = b.cpart + (size_t)ncells * CELL_SPLIT * d;
      hipLaunchKernelGGL(knn_cell_sum_kernel, dim3((unsigned)ncells, CELL_SPLIT), dim3(256), 0, st, (const double*)b.X, d, (const int64_t*)b.cell_starts, n,
                         ncells, b.cpart);
      hipLaunchKernelGGL(knn_cell_centre_kernel, dim3((unsigned)ncells), dim3(256), 0, st, (const double*)b.cpart, d, (const int64_t*)b.cell_starts, n, ncells,
                         b.cen);
      hipLaunchKernelGGL(knn_cell_rad_kernel, dim3((unsigned)ncells, CELL_SPLIT), dim3(256), (size_t)(d + 256) * 8, st, (const double*)b.X, d,
                         (const int64_t*)b.cell_starts, n, ncells, (const double*)b.cen, prad);
      hipLaunchKernelGGL(knn_cell_radfin_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, (const double*)prad,
                         (const int64_t*)b.cell_starts, n, ncells, b.rad);
      hipLaunchKernelGGL(knn_runs_kernel, dim3((unsigned)((nqb + 255) / 256)), dim3(256), 0, st, (const unsigned char*)nullptr,
                         (const int64_t*)b.cell_starts, n, ncells, BR, q0, q1, nqb, b.maxruns, b.runs, b.nruns, (unsigned long long*)nullptr);
      GLX_HIP(hipGetLastError());
      GLX_POOL(glx_pool_alloc((void**)&b.pre_d, (size_t)nq * 2 * KP * 4));
      GLX_POOL(glx_pool_alloc((void**)&b.pre_i, (size_t)nq * 2 * KP * 4));
      rc = knn_launch_tile_bf16(KP, NKB, b, n, q0, q1, seed_sub, st, cat, true);
      if (rc) return rc;
      rc = knn_launch_seed(KP, b, nq, q0, k, cerr, st);
      if (rc) return rc;
      hipLaunchKernelGGL(knn_cellmask_kernel, dim3((unsigned)nqb), dim3(256), (size_t)16 * d * 8, st, (const double*)b.X, d, q0, q1, (const double*)b.cen,
                         (const double*)b.rad, ncells, (const double*)b.ub2, b.mask);
      GLX_POOL(glx_pool_alloc((void**)&b.visited, 8));
      GLX_HIP(hipMemsetAsync(b.visited, 0, 8, st));
      hipLaunchKernelGGL(knn_runs_kernel, dim3((unsigned)((nqb + 255) / 256)), dim3(256), 0, st, (const unsigned char*)b.mask,
                         (const int64_t*)b.cell_starts, n, ncells, BR, q0, q1, nqb, b.maxruns, b.runs, b.nruns, b.visited);
      GLX_HIP(hipGetLastError());
      g_knn_stats[12] = (double)ncells;
    }
    rc = knn_launch_tile_bf16(KP, NKB, b, n, q0, q1, nsplit, st, cat, false);
  } else {
    GLX_POOL(glx_pool_alloc((void**)&b.Rf, (size_t)n * dpa * 4));
    GLX_POOL(glx_pool_alloc((void**)&b.Qf, (size_t)n * dpa * 4));
    hipLaunchKernelGGL(knn_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)b.X, (const double*)b.mean,
                       n, d, dpa, b.Rf, b.Qf, b.qnorm);
    GLX_HIP(hipGetLastError());
    rc = knn_launch_tile_f32(KP, DH, nkb, b, n, q0, q1, nsplit, st);
  }
  if (rc) return rc;
  GLX_HIP(hipEventRecord(b.e1, st));
  rc = knn_launch_rerank(b, n, d, k, q0, nq, lists, KP, M, cerr, st);
  if (rc) return rc;
  GLX_HIP(hipEventRecord(b.e2, st));
  float h_rmax[2] = {0.f, 0.f};
  int h_nbad = 0;
  GLX_HIP(hipMemcpyAsync(&h_nbad, b.nbad, 4, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipMemcpyAsync(h_rmax, b.rmax, 8, hipMemcpyDeviceToHost, st));
  unsigned long long h_visited = 0;
  if (b.visited) GLX_HIP(hipMemcpyAsync(&h_visited, b.visited, 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  if (b.visited) {
    g_knn_stats[11] = (double)h_visited / ((double)nqb * (double)ntiles);
    if (timing) fprintf(stderr, "[glx] knn: cell pruning, %d cells: %.1f %% of the (query block, ref tile) pairs visited\n", ncells, 100.0 * g_knn_stats[11]);
  }
  stamp("tile + re-rank done, flags on the host");
  GLX_CHECK(h_rmax[1] == 1.0f, GLX_EINVAL, "glx_knn_bruteforce: non-finite input");   // (the first host look at the centring pass)
  struct { size_t n; size_t size() const { return n; } bool empty() const { return n == 0; } } rows = {(size_t)h_nbad};   // (the list itself is on the device: b.rows)
  if (short_lists && rows.size() > 64) {
    // repair row by row, or search again with the long lists?  A fallback row streams the data once (measured: ~5 TB/s);
    // the repeat costs about four tile-kernel times (fp32-input filter, longer lists)
    // The first pass is priced by a MODEL, not by its measured time: the choice must not depend on who else uses the GPU (with six
    // processes sharing it the measured pass came out long enough, once in thirty runs, to send 21 000 rows of
    // tests/test_gpu_knn.py::test_search_on_data_sorted_by_locality through the row-by-row repair -- the right answer, the slow way).
    // 1.26e10 (query, ref, 16-feature block) triples per ms: config 2's 0.78 ms for 70 000^2 pairs of two blocks.
    const double share = b.visited ? std::max(g_knn_stats[11], 0.01) : 1.0;
    // (the fp32-input filter runs at a quarter of that: profiles/r02_knn_filter_probe.txt, d = 64 / 128; 0.03 ms: launches + the host look of a tiny pass)
    const double ms_first = std::max(0.03, (double)nq * (double)n * share * ((double)dpa / 16.0) / 1.26e10 * (use_bf16 ? 1.0 : 4.0));
    const double ms_rows = (double)rows.size() * ((double)n * d * 8.0 / 5e9);        // (one pass per row: knn_fallback_collect_kernel)
    if (ms_rows > 4.0 * ms_first) {
      g_knn_stats[2] = (double)rows.size();
      return KNN_ESCALATE;
    }
  }
  if (!rows.empty()) {
    const size_t nr = rows.size();
    GLX_POOL(glx_pool_alloc((void**)&b.fb_pd, nr * FB_SPLIT * k * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.fb_pi, nr * FB_SPLIT * k * 4));
    GLX_POOL(glx_pool_alloc((void**)&b.fb_cnt, nr * 2 * 4));            // [nr] counts, [nr] redo marks
    GLX_POOL(glx_pool_alloc((void**)&b.fb_bd, nr * FB_CAP * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.fb_bi, nr * FB_CAP * 4));
    GLX_HIP(hipMemsetAsync(b.fb_cnt, 0, nr * 2 * 4, st));
    const int* fb_runs = (const int*)(b.visited ? b.runs : nullptr);   // (b.runs: the main pass's runs when the search was cell-pruned -- the pre-pass's were overwritten by them)
    rc = knn_launch_fallback(b, n, d, k, q0, nr, fb_runs, BR, st);
    if (rc) return rc;
  }
  GLX_HIP(hipEventRecord(b.e3, st));
  if (ind_out) GLX_UP(glx_download(ind_out, b.ind, (size_t)nq * k * 8, st, __func__));
  if (dist_out) GLX_UP(glx_download(dist_out, b.dist, (size_t)nq * k * 8, st, __func__));
  GLX_HIP(hipStreamSynchronize(st));
  stamp("results on the host");
  if (perm_pending) {    // the permutation stays on the device with the result (glx_knn_result_order copies it into the caller's --
    glx_pool_free(capture->order_dev);      // page-locked -- array: a synchronous copy into fresh pageable memory cost 8 ms here)
    capture->order_dev = b.orig;
    b.orig = nullptr;
  }
  if (capture) {         // the lists stay on the device with the caller's result object (everything that writes them has finished)
    glx_pool_free(capture->ind);
    glx_pool_free(capture->dist);
    capture->ind = b.ind;
    capture->dist = b.dist;
    capture->n = n;
    capture->k = k;
    capture->device = device;
    b.ind = nullptr;
    b.dist = nullptr;
  }
  float ms_tile = 0, ms_rr = 0, ms_fb = 0;
  GLX_HIP(hipEventElapsedTime(&ms_tile, b.e0, b.e1));
  GLX_HIP(hipEventElapsedTime(&ms_rr, b.e1, b.e2));
  GLX_HIP(hipEventElapsedTime(&ms_fb, b.e2, b.e3));
  g_knn_stats[0] = ms_tile;
  g_knn_stats[1] = ms_rr;
  g_knn_stats[2] = (double)rows.size();
  g_knn_stats[3] = ms_tile + ms_rr + ms_fb;
  g_knn_stats[4] = ms_fb;
  g_knn_stats[5] = (double)dpa;
  g_knn_stats[6] = (double)nsplit;
  g_knn_stats[7] = use_bf16 ? -(double)KP : (double)KP;   // negative: the bf16 filter ran
  return GLX_OK;
}

static int knn_run(const double* X, int64_t n, int d, int k, int64_t q0, int64_t q1, int64_t* ind_out, double* dist_out, int device,
                   glx_knn_result* capture = nullptr, const int64_t* cell_starts = nullptr, int ncells = 0, int auto_cells = 0) {
  g_knn_stats[8] = 0.0;
  g_knn_stats[10] = g_knn_stats[11] = g_knn_stats[12] = 0.0;
  int rc = knn_pass(X, n, d, k, q0, q1, ind_out, dist_out, device, false, capture, cell_starts, ncells, auto_cells);
  if (rc != KNN_ESCALATE) return rc;
  const double flagged = g_knn_stats[2];
  g_knn_stats[10] = g_knn_stats[11] = g_knn_stats[12] = 0.0;
  if (capture) { capture->order.clear(); glx_pool_free(capture->order_dev); capture->order_dev = nullptr; }
  rc = knn_pass(X, n, d, k, q0, q1, ind_out, dist_out, device, true, capture);     // (long lists: the fp32-input kernel, all refs)
  g_knn_stats[8] = flagged;            // rows the first (short-list) pass could not accept
  return rc;
}

extern "C" int glx_knn_bruteforce(const double* X, int64_t n, int d, int k, int similarity, int64_t* ind_out, double* dist_out,
                                  int device) {
  GLX_CHECK(similarity == 0, GLX_EINVAL,
            "glx_knn_bruteforce: similarity %d; only euclidean (0) -- normalise rows on the host for angular", similarity);
  return knn_run(X, n, d, k, 0, n, ind_out, dist_out, device);
}

extern "C" int glx_knn_bruteforce_range(const double* X, int64_t n, int d, int k, int64_t q_begin, int64_t q_end,
                                        int64_t* ind_out, double* dist_out, int device) {
  return knn_run(X, n, d, k, q_begin, q_end, ind_out, dist_out, device);
}

// The same search -- the same lists, bit for bit -- for rows that come in a coarse geometric order: cell c = the rows
// [cell_starts[c], cell_starts[c + 1]) (the last cell ends at n; empty cells allowed).  Per query block only the cells that can hold
// one of its k nearest are visited (bounds from the cells' centres and radii against the k-th distance within a sample of the
// block's own cells); everything skipped is strictly farther than the k-th neighbour.  Takes the place of the tree the reference
// searches with (scipy cKDTree / annoy, graphlearning/weightmatrix.py:297-429) at sizes where all pairs are too many.
extern "C" int glx_knn_cells_range(const double* X, int64_t n, int d, int k, const int64_t* cell_starts, int ncells, int64_t q_begin,
                                   int64_t q_end, int64_t* ind_out, double* dist_out, int device) {
  GLX_CHECK(cell_starts && ncells >= 1, GLX_EINVAL, "glx_knn_cells_range: null argument");
  GLX_CHECK(ncells <= 4096, GLX_EUNSUPPORTED, "glx_knn_cells_range: %d cells above the supported 4096", ncells);
  GLX_CHECK(cell_starts[0] == 0, GLX_EINVAL, "glx_knn_cells_range: the first cell must start at row 0");
  for (int c = 1; c < ncells; ++c)
    GLX_CHECK(cell_starts[c] >= cell_starts[c - 1] && cell_starts[c] <= n, GLX_EINVAL, "glx_knn_cells_range: cell starts must ascend within [0, n]");
  return knn_run(X, n, d, k, q_begin, q_end, ind_out, dist_out, device, nullptr, cell_starts, ncells);
}

// All n rows in the caller's order, the cells formed here: ncells evenly spaced rows serve as centres, every row joins the
// nearest one, the rows are reordered by cell on the device and searched with the pruning of glx_knn_cells_range; indices and
// output rows are the caller's, ties between equal distances go to the lower caller index -- the lists of glx_knn_bruteforce,
// bit for bit.  On data without cluster structure every cell stays in play and the extra passes cost a few per cent.
// ncells < -1: the rows reordered by -ncells chained cells, then all pairs (coherent wavefronts below the size where pruning pays).
extern "C" int glx_knn_clustered(const double* X, int64_t n, int d, int k, int ncells, int64_t* ind_out, double* dist_out, int device) {
  GLX_CHECK(ncells >= -4096 && ncells <= 4096, GLX_EINVAL, "glx_knn_clustered: ncells=%d outside [-4096, 4096]", ncells);
  return knn_run(X, n, d, k, 0, n, ind_out, dist_out, device, nullptr, nullptr, 0, ncells);
}

// ---- search results as objects -------------------------------------------------------------------------------------------------
// glx_knn_search runs the full search (every row a query) and leaves the lists ON THE DEVICE in a result object the caller owns:
// glx_knn_result_to_csr (assemble.hip) builds the weight matrix from them without a host round trip, glx_knn_result_lists copies
// them out, glx_knn_result_order returns the cell order the search worked out (if it did: contiguous, chained cells of feature
// space -- on clustered data as good a locality order for the graph's operators as the library's own pass over the graph,
// glx_graph_set_order, and free), glx_knn_result_destroy releases everything.  Nothing is handed from one call to the next through
// hidden state.
extern "C" int glx_knn_search(const double* X, int64_t n, int d, int k, int ncells, int device, glx_knn_result** out) {
  GLX_CHECK(out, GLX_EINVAL, "glx_knn_search: null output");
  *out = nullptr;
  GLX_CHECK(ncells >= -4096 && ncells <= 4096, GLX_EINVAL, "glx_knn_search: ncells=%d outside [-4096, 4096]", ncells);
  glx_knn_result* res = new glx_knn_result();
  const auto t_call = std::chrono::steady_clock::now();
  const int rc = knn_run(X, n, d, k, 0, n, nullptr, nullptr, device, res, nullptr, 0, (ncells > 1 || ncells < -1) ? ncells : 0);
  if (getenv("GLX_TIMING"))
    fprintf(stderr, "[glx] knn: search returns after %.2f ms (work buffers released)\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
  if (rc || !res->ind) {
    glx_knn_result_destroy(res);
    if (!rc) glx_set_error("glx_knn_search: the search left no lists behind");
    return rc ? rc : GLX_EINVAL;
  }
  *out = res;
  return GLX_OK;
}

extern "C" int glx_knn_result_lists(const glx_knn_result* res, int64_t* ind_out, double* dist_out) {
  GLX_CHECK(res && res->ind && res->dist, GLX_EINVAL, "glx_knn_result_lists: empty result");
  GLX_HIP(hipSetDevice(res->device));
  const size_t bytes = (size_t)res->n * res->k * 8;
  if (ind_out) GLX_UP(glx_download_sync(ind_out, res->ind, bytes, __func__));
  if (dist_out) GLX_UP(glx_download_sync(dist_out, res->dist, bytes, __func__));
  return GLX_OK;
}

__global__ __launch_bounds__(256) void knn_copy_i32_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

extern "C" int glx_knn_result_order(const glx_knn_result* res, int32_t* perm_out) {
  GLX_CHECK(res && perm_out, GLX_EINVAL, "glx_knn_result_order: null argument");
  if (res->order_dev) {
    GLX_HIP(hipSetDevice(res->device));
    // Page-locked destinations (what _hip.KnnResult.order passes) are written by a kernel: the first copy-engine transfer after the
    // search's allocations took 8 ms (measured: hipMemcpy and hipMemcpyAsync alike, 0.02 ms on every later call), a kernel's
    // stores into mapped host memory 0.03 ms.
    glx_work* w = nullptr;
    int rcw = glx_work_acquire(res->device, &w);
    if (rcw) return rcw;
    void* dev_view = nullptr;
    hipError_t e;
    if (hipHostGetDevicePointer(&dev_view, perm_out, 0) == hipSuccess && dev_view) {
      hipLaunchKernelGGL(knn_copy_i32_kernel, dim3((unsigned)((res->n + 255) / 256)), dim3(256), 0, w->stream, (const int32_t*)res->order_dev,
                         (int32_t*)dev_view, res->n);
      e = hipGetLastError();
    } else {
      (void)hipGetLastError();
      e = hipMemcpyAsync(perm_out, res->order_dev, (size_t)res->n * 4, hipMemcpyDeviceToHost, w->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
    glx_work_release(w);
    GLX_HIP(e);
    return GLX_OK;
  }
  GLX_CHECK((int64_t)res->order.size() == res->n && res->n > 0, GLX_EINVAL, "glx_knn_result_order: this search worked out no cell order");
  memcpy(perm_out, res->order.data(), (size_t)res->n * 4);
  return GLX_OK;
}

extern "C" int glx_knn_result_destroy(glx_knn_result* res) {
  if (!res) return GLX_OK;
  glx_pool_free(res->ind);
  glx_pool_free(res->dist);
  glx_pool_free(res->order_dev);
  delete res;
  return GLX_OK;
}

