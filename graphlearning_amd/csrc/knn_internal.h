// Shared by the translation units of the exact kNN search (weightmatrix.knnsearch of the reference,
// graphlearning/weightmatrix.py:297-429):
//   knn.hip            the driver (one pass of the search, the escalation, the C-ABI entry points)
//   knn_prep.hip       centring, the filter's operand images, the seeding pre-pass's thresholds
//   knn_tile_bf16_*.hip / knn_tile_f32_*.hip   the MFMA candidate filters (knn_tile_bf16.h / knn_tile_f32.h hold the kernels;
//                      one translation unit per list length so that they compile side by side)
//   knn_cells.hip      cells formed by the library, the row reorder, the cell pruning
//   knn_rerank.hip     the exact fp64 re-rank with its acceptance test, the exact fallback
#pragma once
#include "glx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#define GLX_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static const int BQ = 128;             // queries per workgroup (4 waves x 32)
static const int BR_MAX = 128;         // refs per LDS tile: 32 * NSUB
static const int KBUF = 8;             // per-lane append slots between list merges
static const int KNN_PAD_ROWS = 256;   // spare rows behind Xb / nrm (>= the widest ref tile): the staging loads of the last tile need no predicates
static const int KNN_CAT_SEG = 21;     // concatenated split operands: three segments of this many bf16 per row (d <= 21)
static const int CENTRE_ROWS = 512;    // rows per workgroup of the column-sum pass
static const int CELL_SPLIT = 64;      // workgroups per cell in the centre / radius passes (a cell of config 4 at n = 1e7 is 80 MB)
static const int FB_SPLIT = 64;        // pieces a fallback row's refs are cut into
static const int FB_CACHE = 2048;      // a piece of at most this many refs keeps its distances in LDS between the rounds
static const int FB_CAP = 128;         // candidates per row the one-pass fallback can hold

// features per half per block of the blocked (d > 130) fp32 variant; 16 where the KP = 64 lists leave less LDS
constexpr int knn_kb(int KP) { return KP == 64 ? 16 : 32; }   // (KP = 8 never takes the blocked variant)

// fp32-input filter: refs per tile = 32*NSUB, as many as fit LDS (160 KiB) beside the candidate lists
constexpr int tile_nsub(int DH, int KP) {
  const int stride = 2 * DH + 2;
  if (KP == 8) {   // short lists: aim at three workgroups per CU
    for (int ns = 4; ns >= 2; ns /= 2)
      if (2 * 32 * ns * stride * 4 + (KP + KBUF) * 256 * 8 <= 53 * 1024) return ns;
    return 1;
  }
  for (int ns = 4; ns >= 2; ns /= 2)
    if (2 * 32 * ns * stride * 4 + (KP + KBUF) * 256 * 8 <= 78 * 1024) return ns;   // two workgroups per CU
  return 1;
}

// bf16 filter: refs per tile = 32 * NSUB.  Measured (one box): 16-32 features: NSUB 2 (config 2: 1.88 vs 2.12 ms, config 3: 2.74 vs
// 3.09 ms); 64 features: NSUB 1 -- a 17 KB tile lets three workgroups share a CU (n = 1e6: 376 vs 401 ms)
constexpr int bf16_nsub(int NKB, int KP) { return (NKB >= 4 || KP >= 32) ? 1 : 2; }
// 8-entry lists in registers where that buys a fourth workgroup per CU (d = 49 .. 64: 122 registers, 34 KB of LDS; measured
// +4 % at n = 3e5 .. 1e6; at fewer feature blocks the registers spill, at more the kernel is register-bound anyway)
constexpr bool bf16_reglists(int NKB, int KP) { return KP == 8 && NKB == 4; }

// device buffers of one pass of the search (pooled blocks; the destructor drains the stream first)
struct KnnBufs {
  unsigned short* Xb = nullptr;      // bf16 hi | lo image (bf16 filter); concatenated form: the ref image [hi | hi | lo]
  unsigned short* Xq = nullptr;      // concatenated form only: the query image [hi | lo | hi]
  float* nrm = nullptr;              // fp32 squared norms (bf16 filter)
  double* part = nullptr;            // per-block partial column sums / maxima of the centring pass
  float* rmax = nullptr;             // [0] largest centred norm (1 + 1e-6), [1] 1 if the input is finite: written by knn_rmax_kernel
  double *X = nullptr, *mean = nullptr, *dist = nullptr;
  float *Rf = nullptr, *Qf = nullptr, *qnorm = nullptr, *cand_d = nullptr;
  float* pre_d = nullptr;
  int* pre_i = nullptr;
  // cell pruning: tile runs of the query blocks (current launch), cell geometry, the per-query bound of the pre-pass
  int *runs = nullptr, *nruns = nullptr;
  int maxruns = 0;
  int64_t* cell_starts = nullptr;
  double *cen = nullptr, *rad = nullptr, *ub2 = nullptr, *cpart = nullptr;
  unsigned char* mask = nullptr;
  unsigned long long* visited = nullptr;      // ref tiles the query blocks visit, summed (statistics)
  // cells formed by the library: the rows reordered by cell (X points at the reordered copy), orig[position] = the caller's row
  double* Xraw = nullptr;
  int *orig = nullptr, *cell_id = nullptr;
  int *cand_i = nullptr, *flags = nullptr, *rows = nullptr, *fb_pi = nullptr, *gtau = nullptr;
  double* fb_pd = nullptr;
  double *dk2 = nullptr, *fb_bd = nullptr;   // exact k-th candidate distance^2 of flagged rows; the one-pass fallback's buffers
  int *fb_cnt = nullptr, *fb_bi = nullptr, *nbad = nullptr, *place = nullptr, *bh = nullptr;
  int64_t* ind = nullptr;
  glx_work* work = nullptr;           // the device's cached stream + events
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
  ~KnnBufs() {
    if (stream) hipStreamSynchronize(stream);   // pooled blocks are reused at once: nothing may still be running on them
    if (work && work->side) hipStreamSynchronize(work->side);
    void* blocks[] = {Xb, Xq, nrm, part, rmax, X, mean, dist, Rf, Qf, qnorm, cand_d, pre_d, pre_i, runs, nruns, cell_starts, cen, rad, ub2, cpart,
                      mask, visited, Xraw, orig, cell_id, cand_i, flags, rows, fb_pi, gtau, fb_pd, dk2, fb_bd, fb_cnt, fb_bi, nbad, place, bh, ind};
    for (void* p : blocks) glx_pool_free(p);
    glx_work_release(work);
  }
};

// ---- knn_prep.hip ------------------------------------------------------------------------------------------------------
__global__ void knn_prep_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d, int dpa,
                                float* __restrict__ Rf, float* __restrict__ Qf, float* __restrict__ qnorm);
__global__ void knn_prep_bf16_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d, int kpad,
                                     unsigned short* __restrict__ Xb, float* __restrict__ nrm, float* __restrict__ qnorm);
__global__ void knn_prep_bf16_cat_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d,
                                         unsigned short* __restrict__ Xa, unsigned short* __restrict__ Xq, float* __restrict__ nrm,
                                         float* __restrict__ qnorm, int fold);
__global__ __launch_bounds__(256) void knn_colsum_kernel(const double* __restrict__ X, int64_t n, int d, int dt, double* __restrict__ part);
__global__ __launch_bounds__(256) void knn_mean_kernel(const double* __restrict__ part, int64_t nblk, int d, int64_t n, double* __restrict__ mean);
__global__ __launch_bounds__(256) void knn_maxnorm_kernel(const double* __restrict__ X, const double* __restrict__ mean, int64_t n, int d,
                                                          double* __restrict__ part);
__global__ __launch_bounds__(256) void knn_rmax_kernel(const double* __restrict__ part, int64_t nblk, float* __restrict__ rmax_out);
// the pre-pass's two lists per query (2 KP candidates) -> a starting threshold per query (gtau) and the bound ub2 the cell pruning uses
int knn_launch_seed(int KP, const KnnBufs& b, int64_t nq, int64_t q0, int k, double cerr, hipStream_t st);

// ---- knn_tile_bf16_k*.hip / knn_tile_f32_*.hip: the candidate filters ------------------------------------------------------
// nsplit ref ranges with a tile stride of nsplit (the search proper), or -- seed = true -- ONE range with a stride of nsplit
// writing the pre-pass's own two lists per query (KnnBufs::pre_d / pre_i).  cat: 0 blocks of 16 features, 1 concatenated operands
// (17 <= d <= 21), 2 also the norm folded into them (d <= 20)
int knn_launch_tile_bf16(int KP, int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed);
int knn_launch_tile_f32(int KP, int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st);

// ---- knn_cells.hip -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_gather_rows_kernel(const double* __restrict__ X, const int* __restrict__ rows, int64_t m, int d,
                                                              double* __restrict__ out);
__global__ __launch_bounds__(256) void knn_cellrank_hist_kernel(int* __restrict__ cell, const int* __restrict__ place, int64_t n, int m, int* __restrict__ bh);
__global__ __launch_bounds__(256) void knn_cellrank_scan_kernel(int* __restrict__ bh, int nb, int m);
__global__ __launch_bounds__(256) void knn_cellrank_base_kernel(int* __restrict__ bh, int nb, int m);
__global__ __launch_bounds__(256) void knn_cellrank_scatter_kernel(const int* __restrict__ key, int64_t n, int m, const int* __restrict__ bh, int nb,
                                                                   int* __restrict__ perm);
__global__ __launch_bounds__(256) void knn_assign_kernel(const double* __restrict__ X, int d, int64_t n, const double* __restrict__ cen, int m,
                                                         int* __restrict__ cell, int fs);
__global__ __launch_bounds__(256) void knn_cell_sum_kernel(const double* __restrict__ X, int d, const int64_t* __restrict__ cell_starts,
                                                           int64_t n, int ncells, double* __restrict__ part);
__global__ __launch_bounds__(256) void knn_cell_centre_kernel(const double* __restrict__ part, int d, const int64_t* __restrict__ cell_starts,
                                                              int64_t n, int ncells, double* __restrict__ cen);
__global__ __launch_bounds__(256) void knn_cell_rad_kernel(const double* __restrict__ X, int d, const int64_t* __restrict__ cell_starts, int64_t n,
                                                           int ncells, const double* __restrict__ cen, double* __restrict__ prad);
__global__ __launch_bounds__(256) void knn_cell_radfin_kernel(const double* __restrict__ prad, const int64_t* __restrict__ cell_starts, int64_t n,
                                                              int ncells, double* __restrict__ rad);
__global__ __launch_bounds__(256) void knn_cellmask_kernel(const double* __restrict__ X, int d, int64_t q_begin, int64_t q_end,
                                                           const double* __restrict__ cen, const double* __restrict__ rad, int ncells,
                                                           const double* __restrict__ ub2, unsigned char* __restrict__ mask);
__global__ __launch_bounds__(256) void knn_runs_kernel(const unsigned char* __restrict__ mask, const int64_t* __restrict__ cell_starts, int64_t n,
                                                       int ncells, int BR, int64_t q_begin, int64_t q_end, int64_t nqb, int maxruns,
                                                       int* __restrict__ runs, int* __restrict__ nruns, unsigned long long* __restrict__ visited);

// ---- knn_rerank.hip ----------------------------------------------------------------------------------------------------
// exact fp64 distances of the M = lists * KP (rounded up to a power of two) candidates of every query, ranked by (distance, index);
// rows whose lists cannot be proven complete are flagged (b.flags, b.rows, b.nbad, b.dk2)
int knn_launch_rerank(const KnnBufs& b, int64_t n, int d, int k, int64_t q0, int64_t nq, int lists, int KP, int M, double cerr, hipStream_t st);
// the flagged rows (nr of them, listed in b.rows) redone exactly; fb_runs: the tile runs of a cell-pruned search (or null), BR its tile
int knn_launch_fallback(const KnnBufs& b, int64_t n, int d, int k, int64_t q0, size_t nr, const int* fb_runs, int BR, hipStream_t st);
