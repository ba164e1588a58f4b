// Exact kNN search, stages 2 and 3: the exact fp64 re-rank of the filter's candidates with its acceptance test, and the exact
// fallback for the rows that fail it (see knn.hip; reference graphlearning/weightmatrix.py:349-352, whose cKDTree answer this
// reproduces bit for bit).
#include "knn_internal.h"

// ---- stage 2: exact fp64 re-rank + acceptance check ------------------------------------------
// squared distance with the accumulation pattern of scipy's ckdtree sqeuclidean_distance_double
// (4 partial sums over blocks of 4 coordinates, combined left to right, then the tail)
__device__ __forceinline__ double sqdist_exact(const double* __restrict__ u, const double* __restrict__ v, int d) {
#pragma clang fp contract(off)
  double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
  int i = 0;
  for (; i + 4 <= d; i += 4) {
    const double d0 = u[i] - v[i], d1 = u[i + 1] - v[i + 1], d2 = u[i + 2] - v[i + 2], d3 = u[i + 3] - v[i + 3];
    a0 = a0 + d0 * d0;
    a1 = a1 + d1 * d1;
    a2 = a2 + d2 * d2;
    a3 = a3 + d3 * d3;
  }
  double s = a0 + a1 + a2 + a3;
  for (; i < d; ++i) {
    const double dd = u[i] - v[i];
    s = s + dd * dd;
  }
  return s;
}

__device__ __forceinline__ bool lex_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

// one workgroup of 64 threads per query; M (power of two) candidate slots sorted in LDS
// R = candidate slots per lane (M = 64 R <= 512): the candidates stay in registers and are ranked by a bitonic network over the
// wavefront -- partners 64 or more slots apart sit in the same lane, nearer ones are a lane exchange away -- without LDS arrays or
// barriers; R = 0: the LDS network (longer lists).  The acceptance test takes one lane per list.
template <int R>
__global__ __launch_bounds__(64) void knn_rerank_kernel(const double* __restrict__ X, int64_t n, int d, int k, int64_t q_begin,
                                                        int64_t nq, const float* __restrict__ cand_d, const int* __restrict__ cand_i,
                                                        int lists, int KP, int M, const float* __restrict__ qnorm, const float* __restrict__ rmax_p,
                                                        double cerr, int64_t* __restrict__ ind_out, double* __restrict__ dist_out,
                                                        int* __restrict__ flags, const int* __restrict__ orig, int prefilter,
                                                        double* __restrict__ dk2_out, int* __restrict__ nbad, int* __restrict__ badrows) {
  // nbad / badrows: the flagged rows as a list, appended here (in no particular order: nothing depends on it), so that the host
  // reads one count instead of nq flags
  // dk2_out[query]: the exact k-th smallest distance^2 among the candidates -- an upper bound of the true k-th -- for the rows the
  // acceptance test flags (the fallback looks for the refs within it)
  // orig (glx_knn_clustered: the rows were reordered by cell, orig[position] = the caller's row): candidates are ranked by
  // (distance, the CALLER's index) and the caller's indices go out, into the caller's row -- the lists of the search in the
  // caller's order, ties included
  extern __shared__ __attribute__((aligned(16))) char sm[];
  double* sd = (double*)sm;          // [M]
  int* si = (int*)(sd + M);          // [M]
  const int64_t ql = blockIdx.x;
  if (ql >= nq) return;
  const int lane = threadIdx.x;
  const int64_t q = q_begin + ql;
  const int ncand = lists * KP;
  const double* xq = X + q * d;
  // Exact distances only where they can matter: with v_k the k-th smallest FILTER value of the candidates and E the filter's true
  // error (E < 2 eps always, knn.hip), the exact k-th distance^2 is at most v_k + E, and a candidate with a filter value above
  // v_k + 2 E is more than v_k + E away -- farther than the k-th: everything above v_k + 4 eps is left out.  Of 128 candidates a dozen or two remain; the others' rows (d doubles each, scattered over X) are never
  // fetched, which is what this kernel's time was (64 KB of gathers per query at d = 64).
  float* sv = (float*)(si + M);      // [M] filter values (the kernel's dynamic LDS is M * 16 bytes)
  const double rq0 = (double)qnorm[q] + (double)rmax_p[0];
  const double eps0 = cerr * rq0 * rq0;
  for (int c = threadIdx.x; c < (prefilter ? M : 0); c += 64) {
    float v = INFINITY;
    if (c < ncand) {
      const int ci = cand_i[ql * ncand + c];
      if (ci >= 0 && ci < n) v = cand_d[ql * ncand + c];
    }
    sv[c] = v;
  }
  if (prefilter) __syncthreads();
  // (prefilter: from 32 features on -- measured with the LDS sort of round 2: 0.22 -> 0.29 ms at d = 20, 2.13 -> 1.04 ms at
  // d = 128; with the register sort: d = 20 0.12 ms either way, d = 32 (config 3) 0.31 -> 0.26 ms.  GLX_KNN_PREFILTER_D moves it)
  __shared__ float s_vk;
  if (threadIdx.x == 0) s_vk = INFINITY;
  if (prefilter) __syncthreads();
  for (int c = threadIdx.x; c < (prefilter ? ncand : 0); c += 64) {
    const float v = sv[c];
    if (!(v < INFINITY)) continue;
    int before = 0;
    for (int j = 0; j < ncand; ++j) {
      const float y = sv[j];
      before += (y < v || (y == v && j < c)) ? 1 : 0;
    }
    if (before == k - 1) s_vk = v;               // exactly one candidate has this rank
  }
  __syncthreads();
  const double keep = prefilter ? (double)s_vk + 4.0 * eps0 + 1e-6 * fabs((double)s_vk) : INFINITY;
  auto exact_of = [&](int c, double& dd, int& idx) {
    dd = INFINITY;
    idx = 0x7fffffff;
    if (c < ncand && (!prefilter || (double)sv[c] <= keep)) {    // (an invalid slot holds +inf and is skipped unless nothing can be excluded)
      const int ci = cand_i[ql * ncand + c];
      if (ci >= 0 && ci < n) {
        idx = orig ? orig[ci] : ci;
        dd = sqdist_exact(xq, X + (int64_t)ci * d, d);
      }
    }
  };
  const int64_t orow = orig ? (int64_t)orig[q] - q_begin : ql;
  double dk2;
  if constexpr (R > 0) {
    double rd[R];
    int ri[R];
#pragma unroll
    for (int r = 0; r < R; ++r) exact_of(lane + 64 * r, rd[r], ri[r]);      // slot e = lane + 64 r
    // (from 256 slots on the lane exchanges run as loops over the stride -- only what indexes registers by a constant is
    // unrolled: the fully unrolled network of 512 slots took two minutes to compile; the short networks stay unrolled, which is
    // worth 10 % of this kernel at config 2)
    constexpr int UNR = R <= 2 ? 8 : 1;
    auto exchange = [&](int r, int stride, bool up) {      // partners a lane exchange away
      const int lo = __shfl_xor(__double2loint(rd[r]), stride), hi = __shfl_xor(__double2hiint(rd[r]), stride);
      const double od = __hiloint2double(hi, lo);
      const int oi = __shfl_xor(ri[r], stride);
      const bool lower = (lane & stride) == 0;
      const bool mine_first = lex_less(rd[r], ri[r], od, oi);
      const bool take_min = lower == up;
      const bool keep_mine = take_min ? mine_first : !mine_first;
      rd[r] = keep_mine ? rd[r] : od;
      ri[r] = keep_mine ? ri[r] : oi;
    };
#pragma unroll UNR
    for (int size = 2; size < 64; size <<= 1) {           // runs inside a lane's 64-slot rows: the direction depends on the lane alone
      const bool up = (lane & size) == 0;
#pragma unroll UNR
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) exchange(r, stride, up);
      }
    }
#pragma unroll
    for (int size = 64; size <= 64 * R; size <<= 1) {     // the direction depends on the register (and, at size 64, on nothing else)
#pragma unroll
      for (int stride = size >> 1; stride >= 64; stride >>= 1) {      // partners in the same lane
        const int rs = stride / 64;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & rs) == 0) {
            const int r2 = r | rs;
            const bool up = (((64 * r) & size) == 0);
            const bool sw = up ? lex_less(rd[r2], ri[r2], rd[r], ri[r]) : lex_less(rd[r], ri[r], rd[r2], ri[r2]);
            const double td = sw ? rd[r2] : rd[r], ud = sw ? rd[r] : rd[r2];
            const int ti = sw ? ri[r2] : ri[r], ui = sw ? ri[r] : ri[r2];
            rd[r] = td; ri[r] = ti; rd[r2] = ud; ri[r2] = ui;
          }
        }
      }
#pragma unroll UNR
      for (int stride = 32; stride > 0; stride >>= 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) exchange(r, stride, ((64 * r) & size) == 0);
      }
    }
    // slots 0 .. k - 1 (k <= 60 < 64) are lanes 0 .. k - 1 of register 0
    if (lane < k) {
      ind_out[orow * k + lane] = ri[0] == 0x7fffffff ? -1 : ri[0];
      dist_out[orow * k + lane] = sqrt(rd[0]);
    }
    {
      const int lo = __shfl(__double2loint(rd[0]), k - 1), hi = __shfl(__double2hiint(rd[0]), k - 1);
      dk2 = __hiloint2double(hi, lo);
    }
  } else {
    for (int c = threadIdx.x; c < M; c += 64) {
      double dd;
      int idx;
      exact_of(c, dd, idx);
      sd[c] = dd;
      si[c] = idx;
    }
    __syncthreads();
    for (int size = 2; size <= M; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = threadIdx.x; t < M / 2; t += 64) {
          const int lo = (t / stride) * stride * 2 + (t % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const double dl = sd[lo], dh = sd[hi];
          const int il = si[lo], ih = si[hi];
          const bool sw = up ? lex_less(dh, ih, dl, il) : lex_less(dl, il, dh, ih);
          if (sw) { sd[lo] = dh; sd[hi] = dl; si[lo] = ih; si[hi] = il; }
        }
        __syncthreads();
      }
    }
    for (int c = threadIdx.x; c < k; c += 64) {
      ind_out[orow * k + c] = si[c] == 0x7fffffff ? -1 : si[c];
      dist_out[orow * k + c] = sqrt(sd[c]);
    }
    dk2 = sd[k - 1];
  }
  // every ref outside a full list has fp32 dist^2 >= that list's threshold; accept the row only if no such ref can beat the exact
  // k-th neighbour once the fp32 error is allowed for.  One lane per list (the lists' thresholds = their largest entries: they
  // arrive unsorted; INFINITY while a list is not full)
  int bad = 0;
  for (int l = lane; l < lists; l += 64) {
    float tau = 0.f;
    for (int p = 0; p < KP; ++p) tau = fmaxf(tau, cand_d[ql * ncand + l * KP + p]);
    if (tau < INFINITY && !((double)tau >= dk2 + 2.0 * eps0)) bad = 1;
  }
  bad = __any(bad) || !(dk2 < INFINITY);
  if (lane == 0) {
    flags[ql] = bad;
    if (bad) {
      dk2_out[ql] = dk2;
      badrows[atomicAdd(nbad, 1)] = (int)ql;          // (room for every query)
    }
  }
}

int knn_launch_rerank(const KnnBufs& b, int64_t n, int d, int k, int64_t q0, int64_t nq, int lists, int KP, int M, double cerr, hipStream_t st) {
  // (from 32 features on the candidates are screened in fp32 before the exact distances)
#define GLX_RERANK(RR)                                                                                                                    \
  hipLaunchKernelGGL(knn_rerank_kernel<RR>, dim3((unsigned)nq), dim3(64), (size_t)M * 16, st, (const double*)b.X, n, d, k, q0, nq,          \
                     (const float*)b.cand_d, (const int*)b.cand_i, lists, KP, M, (const float*)b.qnorm, (const float*)b.rmax, cerr, b.ind, b.dist, \
                     b.flags, (const int*)b.orig, d >= 32 ? 1 : 0, b.dk2, b.nbad, b.rows)
  if (M == 64) GLX_RERANK(1);
  else if (M == 128) GLX_RERANK(2);
  else if (M == 256) GLX_RERANK(4);
  else if (M == 512) GLX_RERANK(8);
  else GLX_RERANK(0);
#undef GLX_RERANK
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// ---- stage 3: exact fp64 fallback for flagged rows --------------------------------------------
// A flagged row's refs are split over FB_SPLIT workgroups (a single one would read the whole data set k times: 50 ms per
// row at n = 1e7); each finds the k smallest (dist, idx) of its piece by k rounds of "smallest pair lexicographically greater
// than the last one picked", a second kernel merges the pieces' ascending lists.

// Piece `piece` of the refs, flagged row `row`: the k smallest (distance, index) of the piece in ascending order -- k rounds of
// "smallest pair above the last one picked" INSIDE the kernel (round 2 launched a scan and a pick kernel per round: 2 k launches,
// 0.3 ms for three rows at config 2, more than their arithmetic by two orders of magnitude).
__global__ __launch_bounds__(256) void knn_fallback_piece_kernel(const double* __restrict__ X, int64_t n, int d, int k, int64_t q_begin,
                                                                 const int* __restrict__ rows, double* __restrict__ part_d,
                                                                 int* __restrict__ part_i, const int* __restrict__ orig,
                                                                 const int* __restrict__ runs, const int* __restrict__ nruns, int maxruns, int BR,
                                                                 const int* __restrict__ redo) {
  // runs (the cell-pruned search): the refs are those of the tile runs of the row's query block -- everything else is strictly
  // farther than the row's k-th neighbour (knn_cellmask_kernel) -- cut into FB_SPLIT pieces of equally many tiles
  __shared__ double s_d[256];
  __shared__ int s_i[256];
  __shared__ double cache[FB_CACHE];
  const int row = blockIdx.x, piece = blockIdx.y;
  if (redo && !redo[row]) return;                      // the one-pass fallback (knn_fallback_collect / _select) has done this row
  const int64_t ql = rows[row];
  const double* xq = X + (q_begin + ql) * d;
  const int64_t per = (n + FB_SPLIT - 1) / FB_SPLIT;
  int64_t r0 = piece * per, r1 = min(n, r0 + per);
  const int* rr = nullptr;
  int nr = 0;
  int64_t t_lo = 0, t_hi = 0;
  if (runs) {
    const int64_t qb = ql / BQ;
    rr = runs + qb * 2 * (int64_t)maxruns;
    nr = nruns[qb];
    int64_t tv = 0;
    for (int r = 0; r < nr; ++r) tv += rr[2 * r + 1] - rr[2 * r];
    t_lo = tv * piece / FB_SPLIT;
    t_hi = tv * (piece + 1) / FB_SPLIT;
    r0 = 0;
    r1 = (int64_t)FB_CACHE + 1;                       // (no distance cache on this path)
  }
  const bool cached = r1 - r0 <= FB_CACHE;
  if (cached)
    for (int64_t ref = r0 + threadIdx.x; ref < r1; ref += 256) cache[ref - r0] = sqdist_exact(xq, X + ref * d, d);
  __syncthreads();
  double pd = -1.0;
  int pi = -1;
  for (int r = 0; r < k; ++r) {
    double bd = INFINITY;
    int bi = 0x7fffffff;
    auto look = [&](int64_t ref, double dd) {
      const int id = orig ? orig[ref] : (int)ref;
      if (lex_less(pd, pi, dd, id) && lex_less(dd, id, bd, bi)) { bd = dd; bi = id; }
    };
    if (runs) {
      int64_t off = 0;                                // tiles of the runs in front of run q
      for (int q = 0; q < nr; ++q) {
        const int64_t a = rr[2 * q], b = rr[2 * q + 1];
        const int64_t lo = max(a, a + (t_lo - off)), hi = min(b, a + (t_hi - off));
        off += b - a;
        if (lo >= hi) continue;
        const int64_t s0 = lo * BR, s1 = min(n, hi * BR);
        for (int64_t ref = s0 + threadIdx.x; ref < s1; ref += 256) look(ref, sqdist_exact(xq, X + ref * d, d));
      }
    } else {
      for (int64_t ref = r0 + threadIdx.x; ref < r1; ref += 256) look(ref, cached ? cache[ref - r0] : sqdist_exact(xq, X + ref * d, d));
    }
    s_d[threadIdx.x] = bd;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off && lex_less(s_d[threadIdx.x + off], s_i[threadIdx.x + off], s_d[threadIdx.x], s_i[threadIdx.x])) {
        s_d[threadIdx.x] = s_d[threadIdx.x + off];
        s_i[threadIdx.x] = s_i[threadIdx.x + off];
      }
      __syncthreads();
    }
    pd = s_d[0];
    pi = s_i[0];
    __syncthreads();
    if (threadIdx.x == 0) {
      part_d[((size_t)row * FB_SPLIT + piece) * k + r] = pd;
      part_i[((size_t)row * FB_SPLIT + piece) * k + r] = pi;
    }
    if (pi == 0x7fffffff) {               // the piece is exhausted: the remaining slots stay empty
      for (int r2 = r + 1 + threadIdx.x; r2 < k; r2 += 256) {
        part_d[((size_t)row * FB_SPLIT + piece) * k + r2] = INFINITY;
        part_i[((size_t)row * FB_SPLIT + piece) * k + r2] = 0x7fffffff;
      }
      break;
    }
  }
}

// one wavefront per flagged row, lane p at the head of piece p's ascending list: k rounds of a lexicographic minimum over the lanes
// The one-pass form of the fallback.  The re-rank leaves dk2 = the exact k-th smallest distance^2 among the row's candidates: k
// distinct refs lie within it, so the true k nearest do too.  ONE pass over the refs (the same FB_SPLIT pieces, the same runs)
// appends every ref with exact distance^2 <= dk2 to the row's buffer -- k of them plus the few the lists missed --, a wavefront per
// row ranks them by (distance, index) and writes the first k.  Rows whose buffer overflows (FB_CAP: masses of ties) or whose
// bound is not finite are left to the k-round kernels above (redo[row] = 1).
__global__ __launch_bounds__(256) void knn_fallback_collect_kernel(const double* __restrict__ X, int64_t n, int d, int64_t q_begin,
                                                                   const int* __restrict__ rows, const double* __restrict__ dk2,
                                                                   int* __restrict__ cnt, double* __restrict__ buf_d, int* __restrict__ buf_i,
                                                                   const int* __restrict__ orig, const int* __restrict__ runs,
                                                                   const int* __restrict__ nruns, int maxruns, int BR) {
  const int row = blockIdx.x, piece = blockIdx.y;
  const int64_t ql = rows[row];
  const double bound = dk2[ql];
  if (!(bound < INFINITY)) {
    if (piece == 0 && threadIdx.x == 0) cnt[row] = FB_CAP + 1;
    return;
  }
  const double* xq = X + (q_begin + ql) * d;
  auto look = [&](int64_t ref) {
    const double dd = sqdist_exact(xq, X + ref * d, d);
    if (dd <= bound) {
      const int slot = atomicAdd(&cnt[row], 1);
      if (slot < FB_CAP) {
        buf_d[(size_t)row * FB_CAP + slot] = dd;
        buf_i[(size_t)row * FB_CAP + slot] = orig ? orig[ref] : (int)ref;
      }
    }
  };
  if (runs) {
    const int64_t qb = ql / BQ;
    const int* rr = runs + qb * 2 * (int64_t)maxruns;
    const int nr = nruns[qb];
    int64_t tv = 0;
    for (int r = 0; r < nr; ++r) tv += rr[2 * r + 1] - rr[2 * r];
    const int64_t t_lo = tv * piece / FB_SPLIT, t_hi = tv * (piece + 1) / FB_SPLIT;
    int64_t off = 0;                                // tiles of the runs in front of run q
    for (int q = 0; q < nr; ++q) {
      const int64_t a = rr[2 * q], b = rr[2 * q + 1];
      const int64_t lo = max(a, a + (t_lo - off)), hi = min(b, a + (t_hi - off));
      off += b - a;
      if (lo >= hi) continue;
      const int64_t s0 = lo * BR, s1 = min(n, hi * BR);
      for (int64_t ref = s0 + threadIdx.x; ref < s1; ref += 256) look(ref);
    }
  } else {
    const int64_t per = (n + FB_SPLIT - 1) / FB_SPLIT;
    const int64_t r0 = piece * per, r1 = min(n, r0 + per);
    for (int64_t ref = r0 + threadIdx.x; ref < r1; ref += 256) look(ref);
  }
}

__global__ __launch_bounds__(64) void knn_fallback_select_kernel(const int* __restrict__ cnt, const double* __restrict__ buf_d,
                                                                 const int* __restrict__ buf_i, const int* __restrict__ rows, int nrows, int k,
                                                                 int64_t* __restrict__ ind_out, double* __restrict__ dist_out,
                                                                 const int* __restrict__ orig, int64_t q_begin, int* __restrict__ redo) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row >= nrows) return;
  const int c = cnt[row];
  if (c > FB_CAP || c < k) {            // (c < k cannot happen with a sound bound: left to the k-round kernels all the same)
    if (lane == 0) redo[row] = 1;
    return;
  }
  if (lane == 0) redo[row] = 0;
  const int64_t ql = orig ? (int64_t)orig[q_begin + rows[row]] - q_begin : rows[row];
  constexpr int R = FB_CAP / 64;
  double rd[R];
  int ri[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane + 64 * r;
    rd[r] = e < c ? buf_d[(size_t)row * FB_CAP + e] : INFINITY;
    ri[r] = e < c ? buf_i[(size_t)row * FB_CAP + e] : 0x7fffffff;
  }
#pragma unroll
  for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64) {
        const int rs = stride / 64;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & rs) == 0) {
            const int r2 = r | rs;
            const bool up = (((lane + 64 * r) & size) == 0);
            const bool sw = up ? lex_less(rd[r2], ri[r2], rd[r], ri[r]) : lex_less(rd[r], ri[r], rd[r2], ri[r2]);
            const double td = sw ? rd[r2] : rd[r], ud = sw ? rd[r] : rd[r2];
            const int ti = sw ? ri[r2] : ri[r], ui = sw ? ri[r] : ri[r2];
            rd[r] = td; ri[r] = ti; rd[r2] = ud; ri[r2] = ui;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int lo = __shfl_xor(__double2loint(rd[r]), stride), hi = __shfl_xor(__double2hiint(rd[r]), stride);
          const double od = __hiloint2double(hi, lo);
          const int oi = __shfl_xor(ri[r], stride);
          const bool up = (((lane + 64 * r) & size) == 0), lower = (lane & stride) == 0;
          const bool mine_first = lex_less(rd[r], ri[r], od, oi);
          const bool keep_mine = (lower == up) ? mine_first : !mine_first;
          rd[r] = keep_mine ? rd[r] : od;
          ri[r] = keep_mine ? ri[r] : oi;
        }
      }
    }
  }
  if (lane < k) {                       // (k <= 60: the first k slots are lanes 0 .. k - 1 of register 0)
    ind_out[ql * k + lane] = ri[0];
    dist_out[ql * k + lane] = sqrt(rd[0]);
  }
}

__global__ __launch_bounds__(64) void knn_fallback_merge_kernel(const double* __restrict__ part_d, const int* __restrict__ part_i,
                                                                const int* __restrict__ rows, int nrows, int k,
                                                                int64_t* __restrict__ ind_out, double* __restrict__ dist_out,
                                                                const int* __restrict__ orig, int64_t q_begin, const int* __restrict__ redo) {
  static_assert(FB_SPLIT == 64, "one lane per piece");
  const int row = blockIdx.x, p = threadIdx.x;
  if (row >= nrows) return;
  if (redo && !redo[row]) return;
  const int64_t ql = orig ? (int64_t)orig[q_begin + rows[row]] - q_begin : rows[row];
  int head = 0;
  const size_t base = ((size_t)row * FB_SPLIT + p) * k;
  double dd = part_d[base];
  int ii = part_i[base];
  for (int r = 0; r < k; ++r) {
    double bd = dd;
    int bi = ii;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = __shfl_xor(bd, off);
      const int oi = __shfl_xor(bi, off);
      if (lex_less(od, oi, bd, bi)) { bd = od; bi = oi; }
    }
    if (p == 0) {
      ind_out[ql * k + r] = bi == 0x7fffffff ? -1 : bi;
      dist_out[ql * k + r] = sqrt(bd);
    }
    if (bi != 0x7fffffff && dd == bd && ii == bi) {       // (indices are unique: exactly one lane holds the winner)
      ++head;
      dd = head < k ? part_d[base + head] : INFINITY;
      ii = head < k ? part_i[base + head] : 0x7fffffff;
    }
  }
}

int knn_launch_fallback(const KnnBufs& b, int64_t n, int d, int k, int64_t q0, size_t nr, const int* fb_runs, int BR, hipStream_t st) {
  int* redo = b.fb_cnt + nr;
  // one pass: every ref within the bound the re-rank left, ranked by a wavefront per row
  hipLaunchKernelGGL(knn_fallback_collect_kernel, dim3((unsigned)nr, FB_SPLIT), dim3(256), 0, st, (const double*)b.X, n, d, q0, (const int*)b.rows,
                     (const double*)b.dk2, b.fb_cnt, b.fb_bd, b.fb_bi, (const int*)b.orig, fb_runs, (const int*)b.nruns, b.maxruns, BR);
  hipLaunchKernelGGL(knn_fallback_select_kernel, dim3((unsigned)nr), dim3(64), 0, st, (const int*)b.fb_cnt, (const double*)b.fb_bd, (const int*)b.fb_bi,
                     (const int*)b.rows, (int)nr, k, b.ind, b.dist, (const int*)b.orig, q0, redo);
  // the k-round kernels: only the rows the one pass could not finish (their workgroups return at once otherwise)
  hipLaunchKernelGGL(knn_fallback_piece_kernel, dim3((unsigned)nr, FB_SPLIT), dim3(256), 0, st, (const double*)b.X, n, d, k, q0, (const int*)b.rows,
                     b.fb_pd, b.fb_pi, (const int*)b.orig, fb_runs, (const int*)b.nruns, b.maxruns, BR, (const int*)redo);
  hipLaunchKernelGGL(knn_fallback_merge_kernel, dim3((unsigned)nr), dim3(64), 0, st, (const double*)b.fb_pd, (const int*)b.fb_pi,
                     (const int*)b.rows, (int)nr, k, b.ind, b.dist, (const int*)b.orig, q0, (const int*)redo);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}


// ---- distance to the nearest of a few reference rows (graph.reweight(method='properly'), reference graph.py:455-457) ----------------
// `Xtree = cKDTree(X[idx]); D, J = Xtree.query(X)`: for every row of X the euclidean distance to the nearest labelled row -- m labelled
// rows against n, all pairs, with cKDTree's accumulation pattern and its final square root, so D is the reference's bit for bit.
// One thread per query row, the reference rows staged through LDS in pieces.
__global__ __launch_bounds__(256) void knn_nearest_dist_kernel(const double* __restrict__ X, int64_t n, int d, const double* __restrict__ R, int64_t m,
                                                               int piece, double* __restrict__ out) {
  extern __shared__ double s_ref[];       // [piece][d]
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double* xq = X + (q < n ? q : n - 1) * d;
  double best = INFINITY;
  for (int64_t j0 = 0; j0 < m; j0 += piece) {
    const int cnt = (int)(m - j0 < piece ? m - j0 : piece);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * d; i += 256) s_ref[i] = R[j0 * d + i];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const double dd = sqdist_exact(xq, s_ref + (size_t)j * d, d);
      best = dd < best ? dd : best;
    }
  }
  if (q < n) out[q] = sqrt(best);
}

extern "C" int glx_nearest_dist(const double* X, int64_t n, int d, const int64_t* idx, int64_t m, double* dist_out, int device) {
  GLX_CHECK(X && idx && dist_out && n >= 1 && d >= 1 && m >= 1, GLX_EINVAL, "glx_nearest_dist: bad argument");
  GLX_CHECK((size_t)d * 8 <= 48 * 1024, GLX_EUNSUPPORTED, "glx_nearest_dist: %d features exceed one row of the LDS stage", d);
  for (int64_t j = 0; j < m; ++j) GLX_CHECK(idx[j] >= 0 && idx[j] < n, GLX_EINVAL, "glx_nearest_dist: row %lld out of range", (long long)idx[j]);
  GLX_HIP(hipSetDevice(device));
  std::vector<double> R((size_t)m * d);
  for (int64_t j = 0; j < m; ++j) memcpy(R.data() + (size_t)j * d, X + (size_t)idx[j] * d, (size_t)d * 8);
  double *dX = nullptr, *dR = nullptr, *dO = nullptr;
  int rc = glx_pool_alloc((void**)&dX, (size_t)n * d * 8);
  if (!rc) rc = glx_pool_alloc((void**)&dR, R.size() * 8);
  if (!rc) rc = glx_pool_alloc((void**)&dO, (size_t)n * 8);
  if (!rc) {
    const int piece = (int)std::max<int64_t>(1, std::min<int64_t>(m, (48 * 1024 / 8) / d));
    rc = glx_upload_sync(dX, X, (size_t)n * d * 8, "glx_nearest_dist");
    if (!rc) rc = glx_upload_sync(dR, R.data(), R.size() * 8, "glx_nearest_dist");
    hipError_t e = hipSuccess;
    if (!rc) {
      hipLaunchKernelGGL(knn_nearest_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)piece * d * 8, 0, (const double*)dX, n, d,
                         (const double*)dR, m, piece, dO);
      e = hipGetLastError();
    }
    if (!rc && e == hipSuccess) rc = glx_download_sync(dist_out, dO, (size_t)n * 8, "glx_nearest_dist");
    if (!rc && e != hipSuccess) { glx_set_error("glx_nearest_dist: %s", hipGetErrorString(e)); rc = GLX_EHIP; }
  }
  glx_pool_free(dX);
  glx_pool_free(dR);
  glx_pool_free(dO);
  return rc;
}
