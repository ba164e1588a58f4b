// Exact kNN search: cells formed by the library, the row reorder and the cell pruning (see knn.hip; reference
// graphlearning/weightmatrix.py:297-429 -- the cells take the place of the tree the reference searches with).
#include "knn_internal.h"

// ---- cell pruning (glx_knn_cells_range) --------------------------------------------------------
// The rows come in an order in which `cells` are contiguous (a coarse geometric order: nearest of a few dozen sample points, a
// k-means leaf, a tree leaf -- whatever the caller has).  Per cell a centre (the mean) and a radius (the farthest member); a query
// whose k-th neighbour is known to lie within sqrt(ub2) needs no ref of a cell with |q - centre| - radius > sqrt(ub2).  The bound
// ub2 comes from the seeding pre-pass over a sample of the query block's OWN cells; the search proper then visits, per block of
// 128 queries, the tiles of the cells any of its queries still needs.  Exact: a skipped ref is strictly farther than the k-th
// neighbour.  On clustered data (config 4: ten Gaussian blobs in 64 dimensions) nine tenths of the tiles go.
// ---- glx_knn_clustered: cells formed by the library ---------------------------------------------
// out[i] = X[rows[i]] (rows of d doubles)
__global__ __launch_bounds__(256) void knn_gather_rows_kernel(const double* __restrict__ X, const int* __restrict__ rows, int64_t m, int d,
                                                              double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m * d) return;
  out[i] = X[(int64_t)rows[i / d] * d + i % d];
}

// cell[i] = the nearest of m centres (lowest index on ties); centres in batches of 8 through LDS, one walk over a point's
// features per batch
// Rows into the order of their cells on the device (stable: ascending caller index inside a cell) -- three small kernels instead of a
// trip to the host: key = the cell's place in the chain, a histogram per block of 256 rows, a scan of the (block, key) table per key,
// and a scatter that ranks a row among the earlier rows of its block with the same key.  The permutation equals the host's
// counting sort (finish_order); the search never waits for it.
__global__ __launch_bounds__(256) void knn_cellrank_hist_kernel(int* __restrict__ cell, const int* __restrict__ place, int64_t n, int m, int* __restrict__ bh) {
  extern __shared__ int h_[];
  for (int c = threadIdx.x; c < m; c += 256) h_[c] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int key = place[cell[i]];
    cell[i] = key;
    atomicAdd(&h_[key], 1);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += 256) bh[(int64_t)blockIdx.x * m + c] = h_[c];
}

// exclusive prefix sums of 256 values, one per thread (wave scans + the four wave totals through LDS); returns the block's total
__device__ __forceinline__ int block_excl_scan256(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  __syncthreads();                       // (s_w may still be read by the previous call)
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  int before = 0;
  for (int q = 0; q < w; ++q) before += s_w[q];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return before + incl - v;
}

// The (block, key) table of the histogram pass -> where every (block, key) group of rows starts inside its key's run: one workgroup
// per KEY scans that key's column of the table over the blocks (256 at a time) and leaves the key's total in row `nb` of the table;
// knn_cellrank_base_kernel turns the totals into the keys' starting positions.  (Round 3's form -- ONE workgroup walking every
// key's column with a dependent load per block -- took 90 us at 70 000 rows: a tenth of the search.)
__global__ __launch_bounds__(256) void knn_cellrank_scan_kernel(int* __restrict__ bh, int nb, int m) {
  __shared__ int s_w[4];
  const int c = blockIdx.x;
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const int v = b < nb ? bh[(int64_t)b * m + c] : 0;
    int total;
    const int ex = block_excl_scan256(v, s_w, total);
    if (b < nb) bh[(int64_t)b * m + c] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) bh[(int64_t)nb * m + c] = carry;
}

__global__ __launch_bounds__(256) void knn_cellrank_base_kernel(int* __restrict__ bh, int nb, int m) {
  __shared__ int s_w[4];
  int* tot = bh + (int64_t)nb * m;         // [m] totals in, starting positions out
  int carry = 0;
  for (int c0 = 0; c0 < m; c0 += 256) {
    const int c = c0 + (int)threadIdx.x;
    const int v = c < m ? tot[c] : 0;
    int total;
    const int ex = block_excl_scan256(v, s_w, total);
    if (c < m) tot[c] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void knn_cellrank_scatter_kernel(const int* __restrict__ key, int64_t n, int m, const int* __restrict__ bh, int nb,
                                                                   int* __restrict__ perm) {
  __shared__ int k_[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kk = i < n ? key[i] : -1;
  k_[threadIdx.x] = kk;
  __syncthreads();
  if (i < n) {
    int r = 0;
    for (int j = 0; j < (int)threadIdx.x; ++j) r += (k_[j] == kk) ? 1 : 0;
    perm[bh[(int64_t)nb * m + kk] + bh[(int64_t)blockIdx.x * m + kk] + r] = (int)i;
  }
}

__global__ __launch_bounds__(256) void knn_assign_kernel(const double* __restrict__ X, int d, int64_t n, const double* __restrict__ cen, int m,
                                                         int* __restrict__ cell, int fs) {
  // fs: feature stride -- beyond 32 features every fs-th one decides the cell (ds = ceil(d / fs) <= 32 of them).  The cells only
  // order the rows (any partition gives the same lists); at d = 128 the full distances cost 0.4 ms in front of a 2.3 ms search
  // four lanes per row, each with a quarter of the centres (lane s: centres s, s + 4, ...), the lowest index among equal minima as
  // a single pass in ascending order would pick it: four times the wavefronts of the one-thread-per-row form (61 -> ~20 us at
  // 70 000 x 20, 128 centres -- the kernel now sits in front of every search below 2^17 rows)
  extern __shared__ double cc[];
  constexpr int S = 4, CB = 16, E = CB / S;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) / S;
  const int sl = threadIdx.x % S;
  const int ds = (d + fs - 1) / fs;
  const double* x = X + (i < n ? i : n - 1) * d;
  double best = INFINITY;
  int bc = 0x7fffffff;
  for (int c0 = 0; c0 < m; c0 += CB) {
    __syncthreads();
    for (int u = threadIdx.x; u < CB * ds; u += 256) {
      const int c = c0 + u / ds;
      cc[u] = c < m ? cen[(int64_t)c * d + (u % ds) * fs] : 0.0;
    }
    __syncthreads();
    double s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s2[e] = 0.0;
    for (int f = 0; f < ds; ++f) {
      const double xf = x[f * fs];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const double df = xf - cc[(sl + S * e) * ds + f];
        s2[e] += df * df;
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = c0 + sl + S * e;
      if (c < m && (s2[e] < best || (s2[e] == best && c < bc))) { best = s2[e]; bc = c; }
    }
  }
#pragma unroll
  for (int off = 1; off < S; off <<= 1) {
    const int lo = __shfl_xor(__double2loint(best), off), hi = __shfl_xor(__double2hiint(best), off);
    const double ob = __hiloint2double(hi, lo);
    const int oc = __shfl_xor(bc, off);
    if (ob < best || (ob == best && oc < bc)) { best = ob; bc = oc; }
  }
  if (i < n && sl == 0) cell[i] = bc == 0x7fffffff ? 0 : bc;
}


// partial column sums of piece s of cell c (fixed order inside a piece; the pieces are added in order by knn_cell_centre_kernel)
__global__ __launch_bounds__(256) void knn_cell_sum_kernel(const double* __restrict__ X, int d, const int64_t* __restrict__ cell_starts,
                                                           int64_t n, int ncells, double* __restrict__ part) {
  __shared__ double red[256];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t a0 = cell_starts[c], b0 = c + 1 < ncells ? cell_starts[c + 1] : n;
  const int64_t len = b0 > a0 ? b0 - a0 : 0;
  const int64_t a = a0 + len * sp / CELL_SPLIT, b = a0 + len * (sp + 1) / CELL_SPLIT;
  int dt = 1;
  while (dt < d && dt < 256) dt *= 2;
  const int col = threadIdx.x % dt, rl = threadIdx.x / dt, rstep = 256 / dt;
  for (int f0 = 0; f0 < d; f0 += dt) {
    double sum = 0.0;
    if (f0 + col < d)
      for (int64_t r = a + rl; r < b; r += rstep) sum += X[r * d + f0 + col];
    red[threadIdx.x] = sum;
    __syncthreads();
    if (rl == 0 && f0 + col < d) {
      double t = 0.0;
      for (int q = 0; q < rstep; ++q) t += red[q * dt + col];
      part[((int64_t)c * CELL_SPLIT + sp) * d + f0 + col] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void knn_cell_centre_kernel(const double* __restrict__ part, int d, const int64_t* __restrict__ cell_starts,
                                                              int64_t n, int ncells, double* __restrict__ cen) {
  const int c = blockIdx.x;
  const int64_t a0 = cell_starts[c], b0 = c + 1 < ncells ? cell_starts[c + 1] : n;
  for (int f = threadIdx.x; f < d; f += 256) {
    double t = 0.0;
    for (int sp = 0; sp < CELL_SPLIT; ++sp) t += part[((int64_t)c * CELL_SPLIT + sp) * d + f];
    cen[(int64_t)c * d + f] = b0 > a0 ? t / (double)(b0 - a0) : 0.0;
  }
}

// largest squared distance of a member of piece s of cell c from the cell's centre
__global__ __launch_bounds__(256) void knn_cell_rad_kernel(const double* __restrict__ X, int d, const int64_t* __restrict__ cell_starts, int64_t n,
                                                           int ncells, const double* __restrict__ cen, double* __restrict__ prad) {
  extern __shared__ double cs[];                 // [d] centre, [256] scratch
  double* red = cs + d;
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t a0 = cell_starts[c], b0 = c + 1 < ncells ? cell_starts[c + 1] : n;
  const int64_t len = b0 > a0 ? b0 - a0 : 0;
  const int64_t a = a0 + len * sp / CELL_SPLIT, b = a0 + len * (sp + 1) / CELL_SPLIT;
  for (int f = threadIdx.x; f < d; f += 256) cs[f] = cen[(int64_t)c * d + f];
  __syncthreads();
  double m = 0.0;
  for (int64_t r = a + threadIdx.x; r < b; r += 256) {
    double s2 = 0.0;
    for (int f = 0; f < d; ++f) {
      const double df = X[r * d + f] - cs[f];
      s2 += df * df;
    }
    m = fmax(m, s2);
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) prad[c * CELL_SPLIT + sp] = red[0];
}

__global__ __launch_bounds__(256) void knn_cell_radfin_kernel(const double* __restrict__ prad, const int64_t* __restrict__ cell_starts, int64_t n,
                                                              int ncells, double* __restrict__ rad) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const int64_t a0 = cell_starts[c], b0 = c + 1 < ncells ? cell_starts[c + 1] : n;
  double m = 0.0;
  for (int sp = 0; sp < CELL_SPLIT; ++sp) m = fmax(m, prad[c * CELL_SPLIT + sp]);
  rad[c] = b0 > a0 ? sqrt(m) * (1.0 + 1e-9) + 1e-300 : -1.0;        // -1: empty cell
}

// one workgroup per query block (BQ = 128 queries): mask[block][c] = some query of the block may have one of its k nearest in cell c.
// Cells in batches of 16 (centres in LDS); a thread keeps the squared distances of its query to 8 of them while it walks the
// query's features once per batch (direct differences: no cancellation whatever the data's offset).
__global__ __launch_bounds__(256) void knn_cellmask_kernel(const double* __restrict__ X, int d, int64_t q_begin, int64_t q_end,
                                                           const double* __restrict__ cen, const double* __restrict__ rad, int ncells,
                                                           const double* __restrict__ ub2, unsigned char* __restrict__ mask) {
  extern __shared__ double cc[];                 // centres of a batch of cells [CB][d]
  __shared__ int need[4096];
  constexpr int CB = 16, PER = CB / (256 / BQ);
  const int64_t qb = blockIdx.x;
  const int j = threadIdx.x & (BQ - 1), g = threadIdx.x / BQ;       // two thread groups share the cells of a batch
  const int64_t q = q_begin + qb * BQ + j;
  const bool live = q < q_end;
  const double* xq = X + (live ? q : q_end - 1) * d;
  const double u2 = live ? ub2[q - q_begin] : -1.0;
  for (int c = threadIdx.x; c < ncells; c += 256) need[c] = 0;
  for (int c0 = 0; c0 < ncells; c0 += CB) {
    __syncthreads();
    for (int i = threadIdx.x; i < CB * d; i += 256) {
      const int c = c0 + i / d;
      cc[i] = c < ncells ? cen[(int64_t)c * d + i % d] : 0.0;
    }
    __syncthreads();
    double s2[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) s2[e] = 0.0;
    const double* cp = cc + g * PER * d;
    for (int f = 0; f < d; ++f) {
      const double xf = xq[f];
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const double df = xf - cp[e * d + f];
        s2[e] += df * df;
      }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int c = c0 + g * PER + e;
      if (c >= ncells || !live) continue;
      const double r = rad[c];
      if (r < 0.0) continue;
      const double gap = sqrt(s2[e]) - r;          // every member of the cell is at least this far from the query
      if (!(gap > 0.0) || !(gap * gap > u2 * (1.0 + 1e-9))) need[c] = 1;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ncells; c += 256) mask[qb * ncells + c] = (unsigned char)need[c];
}

// one thread per query block: the ascending, disjoint runs of ref tiles the block visits.  mask == nullptr: the block's OWN cells
// (those its 128 rows lie in) -- the sample the seeding pre-pass looks at.
__global__ __launch_bounds__(256) void knn_runs_kernel(const unsigned char* __restrict__ mask, const int64_t* __restrict__ cell_starts, int64_t n,
                                                       int ncells, int BR, int64_t q_begin, int64_t q_end, int64_t nqb, int maxruns,
                                                       int* __restrict__ runs, int* __restrict__ nruns, unsigned long long* __restrict__ visited) {
  const int64_t qb = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (qb >= nqb) return;
  const int64_t r0 = q_begin + qb * BQ, r1 = min(q_end, r0 + BQ);
  int* out = runs + qb * 2 * (int64_t)maxruns;
  int nr = 0;
  int64_t last_b = 0;
  for (int c = 0; c < ncells; ++c) {
    const int64_t a = cell_starts[c], b = c + 1 < ncells ? cell_starts[c + 1] : n;
    if (b <= a) continue;
    const bool want = mask ? mask[qb * ncells + c] != 0 : (a < r1 && b > r0);
    if (!want) continue;
    int64_t ta = a / BR, tb = (b + BR - 1) / BR;
    if (ta < last_b) ta = last_b;                  // the tile a cell shares with its predecessor is visited once
    if (tb <= ta) continue;
    if (nr > 0 && out[2 * nr - 1] == ta) {
      out[2 * nr - 1] = (int)tb;
    } else {
      out[2 * nr] = (int)ta;
      out[2 * nr + 1] = (int)tb;
      ++nr;
    }
    last_b = tb;
  }
  nruns[qb] = nr;
  if (visited) {                                    // tiles this block visits (statistics: glx_knn_stats [11])
    unsigned long long tot = 0;
    for (int r = 0; r < nr; ++r) tot += (unsigned long long)(out[2 * r + 1] - out[2 * r]);
    atomicAdd(visited, tot);
  }
}
