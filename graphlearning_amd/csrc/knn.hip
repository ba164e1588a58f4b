// Exact k-nearest-neighbour search on the MI355X: weightmatrix.knnsearch of the reference
// (graphlearning/weightmatrix.py:297-429; kdtree branch :349-352 is the exact answer we
// reproduce).  Three stages:
//   1. candidate filter -- brute-force tiled pairwise squared distances as an
//      (n x d) @ (d x n) contraction on the fp32 matrix cores (v_mfma_f32_32x32x2_f32),
//      refs staged through LDS, the norms folded into the contraction as two extra
//      features so the accumulator IS |q|^2 + |r|^2 - 2 q.r; every lane owns one query
//      column and keeps the KP best of its half of the refs in an LDS list (unsorted, maximum tracked);
//   2. exact re-rank -- fp64 direct-difference distances (the accumulation pattern of
//      scipy cKDTree's sqeuclidean_distance_double) of the candidates, sorted by
//      (distance, index); a row is accepted only if every candidate list's threshold
//      exceeds the exact k-th distance by twice a bound on the fp32 error;
//   3. fallback -- rows that fail the check are redone by an exact fp64 scan.
#include "glx_internal.h"
#include <chrono>
#define GLX_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

static double g_knn_stats[16];
// the row order of the last glx_knn_clustered search (perm[position] = caller's row; cells contiguous, neighbouring cells chained): a
// locality order of the vertices that whoever builds an operator on the graph can hand to glx_graph_set_order instead of the
// library's pass over the graph
static std::vector<int32_t> g_knn_last_order;
static std::mutex g_knn_order_mu;
extern "C" int glx_knn_stats(double stats[16]) {
  GLX_CHECK(stats, GLX_EINVAL, "glx_knn_stats: null output");
  for (int i = 0; i < 16; ++i) stats[i] = g_knn_stats[i];
  return GLX_OK;
}

static const int BQ = 128;   // queries per workgroup (4 waves x 32)
static const int BR_MAX = 128; // refs per LDS tile: 32 * NSUB
static const int KBUF = 8;     // per-lane append slots between list merges

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

// ---- stage 1: MFMA tile kernel -------------------------------------------------------------
// KBLK = false: the whole (padded) feature vector of a query lives in registers (DH features per
// half, d + 2 <= 2*DH <= 132).  KBLK = true (any d): the features are processed in nkb blocks of
// DH per half; each step stages one feature block of the ref tile into LDS, reloads the lane's
// query fragment for that block (prefetched one step ahead) and accumulates into the same MFMA
// accumulators; the selection runs after the last block of a tile.
template <int DH, int KP, int NSUB, bool KBLK>
__global__ __launch_bounds__(256) void knn_tile_kernel(const float* __restrict__ Rf, const float* __restrict__ Qf, int64_t n,
                                                       int64_t q_begin, int64_t q_end, int nsplit, float* __restrict__ cand_d,
                                                       int* __restrict__ cand_i, int ablate, int nkb_arg) {
  static_assert(!KBLK || DH % 4 == 0, "blocked variant loads the query fragment as float4");
  constexpr int DPA = 2 * DH;
  constexpr int BR = 32 * NSUB;
  constexpr int STRIDE = (DH % 2 == 1) ? DPA : DPA + 2;   // floats; ds_read_b64 of 32 rows hits 64 distinct banks
  const int nkb = KBLK ? nkb_arg : 1;
  const int DHT = DH * nkb;                            // features per half over all blocks
  const int64_t dpa = 2 * (int64_t)DHT;                // row stride of Rf / Qf
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                                  // [2][BR][STRIDE]
  float* ld = smem + 2 * BR * STRIDE;                  // [KP + KBUF][256]: the lane's KP best so far, then append slots
  int* li = (int*)(ld + (KP + KBUF) * 256);            // [KP + KBUF][256] indices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t qb = blockIdx.x, sp = blockIdx.y;
  const int64_t q = q_begin + qb * BQ + wave * 32 + j;   // this lane's query
  const int64_t qc = q < q_end ? q : q_end - 1;
  // query fragment: B[k = h][j] for k-step s of feature block kb is feature h*DHT + kb*DH + s
  float bq[DH];
  float bqn[KBLK ? DH : 1];
  const float* qrow = Qf + qc * dpa + (int64_t)h * DHT;
  auto load_bq_next = [&](int kb) {
    if constexpr (KBLK) {
#pragma unroll
      for (int s = 0; s < DH; s += 4) {
        const float4 v = *(const float4*)(qrow + kb * DH + s);
        bqn[s] = v.x; bqn[s + 1] = v.y; bqn[s + 2] = v.z; bqn[s + 3] = v.w;
      }
    }
  };
  if constexpr (!KBLK) {
#pragma unroll
    for (int s = 0; s < DH; ++s) bq[s] = qrow[s];
    // a use in front of the loop: the compiler waits for these loads HERE.  Left pending into the
    // loop they make its wait-counter pass put a vmcnt(0) before the first MFMA of every tile, which
    // also drains the next tile's prefetch that was issued just before
#pragma unroll
    for (int s = 0; s < DH; ++s) asm volatile("" ::"v"(bq[s]));
  }
#pragma unroll
  for (int p = 0; p < KP; ++p) { ld[p * 256 + tid] = INFINITY; li[p * 256 + tid] = -1; }
  float tau = INFINITY;

  const int64_t ntiles = (n + BR - 1) / BR;
  // ref range `sp` = the tiles sp, sp + nsplit, sp + 2 nsplit, ...: INTERLEAVED, not a contiguous block of refs.  Data often comes
  // sorted (by class, along a curve, by a locality order): a query's neighbours are then neighbours in index too, a contiguous
  // range would put all of them into the two lists of one range and overflow them (29 % of the rows of locality-ordered
  // config-4 data took the exact fallback); interleaved, any 32 * nsplit consecutive refs are spread over all the lists
  const int64_t t0 = sp, t1 = ntiles;
  // staging split in two (issue early / write late): the global loads of step u+1 are issued
  // before the MFMAs of step u and land in LDS only after them, so their latency hides
  // under the matrix work instead of stalling the wavefront in front of it.
  constexpr int UNITS = (BR * DH + 255) / 256;   // float2 units per thread per step
  float2 pre[UNITS];
  auto stage_load = [&](int64_t t, int kb) {
    // BR rows x DH float2 units (DH/2 per half when blocked); rows beyond n become "infinitely far" refs
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = tid + i * 256;
      const int r = u / DH, f2 = u % DH;
      const int64_t ref = t * BR + r;
      float2 v;
      v.x = (f2 == DH - 1 && kb == nkb - 1) ? 1e30f : 0.f;   // norm slot (feature dpa-2) of a padding ref
      v.y = 0.f;
      if (u < BR * DH && ref < n) {
        if constexpr (KBLK) v = *(const float2*)(Rf + ref * dpa + (int64_t)(f2 / (DH / 2)) * DHT + kb * DH + 2 * (f2 % (DH / 2)));
        else v = *(const float2*)(Rf + ref * DPA + 2 * f2);
      }
      pre[i] = v;
    }
  };
  auto stage_store = [&](int buf) {
    float* dst = tile + buf * BR * STRIDE;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = tid + i * 256;
      if (u < BR * DH) *(float2*)(dst + (u / DH) * STRIDE + 2 * (u % DH)) = pre[i];
    }
  };
  int cnt = 0;
  const bool share_tau = !(ablate & 2);
  // The per-lane list is kept UNSORTED with its maximum tracked (value tau_own at slot pmax): an
  // accepted candidate overwrites the maximum and the KP entries are rescanned with independent
  // LDS reads -- no dependent shift chain.  The re-rank kernel sorts anyway.
  float tau_own = INFINITY;
  int pmax = 0;
  auto compact = [&]() {
    for (int a = 0; __any(a < cnt); ++a) {
      if (a < cnt) {
        const float v = ld[(KP + a) * 256 + tid];
        if (v < tau_own) {
          ld[pmax * 256 + tid] = v;
          li[pmax * 256 + tid] = li[(KP + a) * 256 + tid];
          float m2 = ld[tid];
          int pm = 0;
#pragma unroll
          for (int p = 1; p < KP; ++p) {
            const float x = ld[p * 256 + tid];
            if (x > m2) { m2 = x; pm = p; }
          }
          tau_own = m2;
          pmax = pm;
        }
      }
    }
    cnt = 0;
    // lanes l and l^32 serve the same query: at least KP refs lie below the smaller of their two
    // thresholds, so that bound filters both halves (the acceptance check in the re-rank kernel,
    // min over all lists of the final thresholds, is unaffected)
    tau = share_tau ? fminf(tau_own, __shfl_xor(tau_own, 32)) : tau_own;
  };
  if (t0 < t1) { stage_load(t0, 0); stage_store(0); load_bq_next(0); }
  __syncthreads();
  int buf = 0;
  f32x16 acc[NSUB];
  for (int64_t t = t0; t < t1; t += nsplit)
  for (int kb = 0; kb < nkb; ++kb) {
    const bool last_kb = kb == nkb - 1;
    const bool has_next = !(last_kb && t + nsplit >= t1);
    if constexpr (KBLK) {
#pragma unroll
      for (int s = 0; s < DH; ++s) bq[s] = bqn[s];
    }
    if (has_next) {
      stage_load(last_kb ? t + nsplit : t, last_kb ? 0 : kb + 1);
      load_bq_next(last_kb ? 0 : kb + 1);
    }
    const float* tl = tile + buf * BR * STRIDE;
    if (kb == 0) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[sub][e] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < DH; s += 2) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const float2 a = *(const float2*)(tl + (sub * 32 + j) * STRIDE + h * DH + s);
        acc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[s], acc[sub], 0, 0, 0);
        if (s + 1 < DH) acc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[s + 1], acc[sub], 0, 0, 0);
      }
    }
    if (last_kb) {
    // selection: acc[sub][e] = dist^2(query j, ref sub*32 + (e&3) + 8*(e>>2) + 4*h).
    // Candidates below the lane's threshold are APPENDED to the lane's LDS slots (cheap, even
    // when only a few lanes have one); when any lane's slots run low the whole wavefront
    // merges its appended candidates into the sorted lists in lockstep, so the insertion
    // cost is paid once per wavefront, not once per lane.
    float m = acc[0][0];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fminf(m, acc[sub][e]);
    if (ablate == 1) {   // developer probe: matrix work + staging only
      if (m == 12345.f) tau = m;
    } else if (__any(m < tau)) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
        for (int eg = 0; eg < 16; eg += 4) {
#pragma unroll
          for (int e = eg; e < eg + 4; ++e) {
            const float v = acc[sub][e];
            if (v < tau && !(ablate & 4)) {
              ld[(KP + cnt) * 256 + tid] = v;
              li[(KP + cnt) * 256 + tid] = (int)(t * BR) + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
              ++cnt;
            }
          }
          if (__any(cnt > KBUF - 4)) compact();
        }
      }
    }
    }   // last_kb
    if (has_next) stage_store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  compact();
  if (q < q_end) {
    const int64_t lists = (int64_t)nsplit * 2;
    const int64_t base = ((q - q_begin) * lists + sp * 2 + h) * KP;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      cand_d[base + p] = ld[p * 256 + tid];
      cand_i[base + p] = li[p * 256 + tid];
    }
  }
}


// ---- stage 1b: candidate filter on the bf16 matrix cores, split operands ----------------------------
// The filter only has to be accurate to a KNOWN bound (the re-rank is exact fp64 and its acceptance test allows for the
// bound), so the contraction does not need fp32 operands: every centred coordinate x is split into two bfloat16 numbers,
// x = hi + lo + e with |e| <= 2^-18 |x|, and q.r is formed as hi.hi + hi.lo + lo.hi -- three v_mfma_f32_32x32x16_bf16
// per 16 features, accumulated in fp32 -- at 16x the rate of the f32-input MFMA: 96 matrix-pipe cycles per 16 features of
// a 32 x 32 tile instead of 512.  bf16 products are exact in fp32; what is dropped (lo.lo and the e terms) is below
// 3.1 * 2^-18 |q||r|, which the acceptance bound `cerr` carries.  The squared norms stay fp32 and are added after the
// contraction (two extra features would lose them to bf16): value = |r|^2 - 2 q.r, compared with tau - |q|^2.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef KNN_GROUP_GUARD
#define KNN_GROUP_GUARD 1          // measured: 438 -> 394 ms
#endif
#ifndef KNN_DIRECT
#define KNN_DIRECT 0                // experiment: A fragments straight from global memory into registers (no LDS tiles, no barrier per tile)
#endif
#ifndef KNN_REGL2
#define KNN_REGL2 0                 // the same for two feature blocks (d = 17 .. 32): see the note at bf16_nstg
#endif
#ifndef KNN_REGLISTS
#define KNN_REGLISTS 1              // 8-entry lists live in registers (LDS then holds tile + append slots only: a fourth workgroup per CU)
#endif
#ifndef KNN_GTAU
#define KNN_GTAU 1
#endif
static const int KNN_PAD_ROWS = 256;   // spare rows behind Xb / nrm (>= the widest ref tile): the staging loads of the last tile need no predicates
#ifndef KNN_COUNT
#define KNN_COUNT 0               // developer probe: event counters of the list maintenance (printed to stderr)
#endif
#if KNN_COUNT
__device__ unsigned long long g_knn_cnt[16];
#define KNN_CNT(i, v) do { knn_ev[i] += (unsigned long long)(v); } while (0)   // per-wave totals in registers, flushed once at the end
#else
#define KNN_CNT(i, v) do { } while (0)
#endif
#if KNN_COUNT == 2   // cycle attribution (s_memtime) per code region, summed over the waves
#define KNN_TIC(var) const unsigned long long var = __builtin_readcyclecounter()
#define KNN_TOC(acc, var) acc += __builtin_readcyclecounter() - var
#else
#define KNN_TIC(var) do { } while (0)
#define KNN_TOC(acc, var) do { } while (0)
#endif
#ifndef KNN_ABLATE
#define KNN_ABLATE 0              // developer probes (wrong results): 1 no list maintenance, 2 no barrier per tile, 4 no staging loads
#endif

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
#ifndef KNN_NORMS_FIRST
#define KNN_NORMS_FIRST 1
#endif
#ifndef KNN_PACKED_SELECT
#define KNN_PACKED_SELECT 0   // measured (profiles/r03_knn_host.txt): v_pk_fma_f32 + nested minima change nothing at config 2 (1.46 ms either way) and cost 3-4 % at d >= 64;
                              // again with the norms read in front of the contraction (scripts/r03_run69.sh): config 3 1.72 vs 1.71 ms, n = 1e6 39.9 vs 38.3 ms
#endif
static const int KNN_CAT_SEG = 21;
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

// (KNN_DIRECT experiment) the ref image in the order the wavefronts fetch it: per tile of BR rows, per sub-tile of 32, per block of 16
// features, per half (first / second 16 KPAD-slots region), the 16 bytes of lane (h, j) side by side
__global__ void knn_frag_layout_kernel(const unsigned short* __restrict__ Xrow, int64_t nrows, int KPAD, int NSUB, int NKB,
                                       unsigned short* __restrict__ Xf, int64_t ntiles) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte unit
  const int64_t total = ntiles * NSUB * NKB * 2 * 64;
  if (u >= total) return;
  const int lane = (int)(u % 64);
  int64_t f = u / 64;
  const int part = (int)(f % 2); f /= 2;
  const int kb = (int)(f % NKB); f /= NKB;
  const int sub = (int)(f % NSUB); f /= NSUB;
  const int64_t tt = f;
  const int h = lane >> 5, j = lane & 31;
  const int64_t row = tt * 32 * NSUB + sub * 32 + j;
  uint4 v = {0u, 0u, 0u, 0u};
  if (row < nrows) v = *(const uint4*)(Xrow + row * 2 * KPAD + part * KPAD + kb * 16 + 8 * h);
  ((uint4*)Xf)[u] = v;
}

// NKB blocks of 16 features (kpad = 16 NKB <= 128); refs are the A operand (LDS), queries the B operand (registers: lane =
// query column j, k-half h); list handling as in knn_tile_kernel.
// CAT (NKB = 2 only): the rows are the concatenated operands above, refs from Xb, queries from Xq.
// NSTG: sub-tiles of 32 NSUB refs staged (and synchronised) together (an experiment that did not pay, see bf16_nstg)
template <int NKB, int KP, int NSUB, int CAT = 0, bool RUNS = false, int NSTG = 1>   // CAT: 1 concatenated operands, 2 also the norm folded into them
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KNN_REGLISTS && KP == 8 && (NKB == 4 || (KNN_REGL2 && NKB == 2))) ? 4 : 1, 4))) void knn_tile_bf16_kernel(const unsigned short* __restrict__ Xb, const unsigned short* __restrict__ Xq, const float* __restrict__ nrm, int64_t n,
                                                            int64_t q_begin, int64_t q_end, int nsplit, float* __restrict__ cand_d,
                                                            int* __restrict__ cand_i, int* __restrict__ gtau, const int* __restrict__ runs,
                                                            const int* __restrict__ nruns, int maxruns) {
  // RUNS (the cell-pruned search, glx_knn_cells_range): this query block visits only the ref tiles of its runs
  // [runs[2 r], runs[2 r + 1]), r < nruns[block] (ascending, disjoint; knn_runs_kernel), not all of them
  // nsplit = the tile stride of a ref range; the number of ranges is the grid's y extent (equal in the search proper; the
  // seeding pre-pass runs ONE range with a larger stride: every 8th tile, say -- a sample of the refs)
  constexpr int KPAD = 16 * NKB;
  constexpr int BR = 32 * NSUB * NSTG;
  constexpr int ROWB = 4 * KPAD + 16;                  // bytes per ref row in LDS: hi | lo, +16 so that 16 rows cover all 64 banks
  constexpr int U_ROW = 4 * KPAD / 16;                 // 16-byte units per row
  constexpr int UNITS = (BR * U_ROW + 255) / 256;
  constexpr bool EXACT_UNITS = (BR * U_ROW) % 256 == 0;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* tile = smem_b;                                  // [2][BR][ROWB]
  float* rn = (float*)(smem_b + 2 * BR * ROWB);         // [2][BR]
  // 8-entry lists in registers where that buys a fourth workgroup per CU (d = 49 .. 64: 122 registers, 34 KB of LDS; measured
  // +4 % at n = 3e5 .. 1e6; at fewer feature blocks the registers spill, at more the kernel is register-bound anyway)
  constexpr bool REGL = KNN_REGLISTS && KP == 8 && (NKB == 4 || (KNN_REGL2 && NKB == 2));   // LDS rows [KP, KP + KBUF) are the append slots either way
  constexpr int LROWS = REGL ? KBUF : KP + KBUF;
  float* ld = rn + 2 * BR - (REGL ? KP * 256 : 0);      // [KP + KBUF][256] (rows [0, KP) do not exist with register lists)
  int* li = (int*)(rn + 2 * BR + LROWS * 256) - (REGL ? KP * 256 : 0);
  float lv[REGL ? 8 : 1];
  int lx[REGL ? 8 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t qb = blockIdx.x, sp = blockIdx.y;
  const int64_t q = q_begin + qb * BQ + wave * 32 + j;
  const int64_t qc = q < q_end ? q : q_end - 1;
  // query fragments: B[k][j], lane holds k = 8h .. 8h+7 of every block, hi and lo
  bf16x8 bh[NKB], bl[NKB];
  {
    const uint4* qrow = (const uint4*)((CAT ? Xq : Xb) + qc * 2 * KPAD);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bh[kb] = __builtin_bit_cast(bf16x8, qrow[kb * 2 + h]);
      bl[kb] = __builtin_bit_cast(bf16x8, qrow[KPAD / 8 + kb * 2 + h]);
    }
    // a use in front of the loop: the compiler waits for these loads HERE.  Left pending into the loop they make its
    // wait-counter pass put decreasing vmcnt waits in front of the MFMAs of EVERY tile, which drain the tile's own staging
    // loads (issued just before) instead of letting them travel under the matrix work
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const uint4 a = __builtin_bit_cast(uint4, bh[kb]), c = __builtin_bit_cast(uint4, bl[kb]);
      asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w));
    }
  }
  const float qn = nrm[qc];
  asm volatile("" ::"v"(qn));
#pragma unroll
  for (int p = 0; p < KP; ++p) {
    if constexpr (REGL) { lv[p] = INFINITY; lx[p] = -1; }
    else { ld[p * 256 + tid] = INFINITY; li[p * 256 + tid] = -1; }
  }
  float tau = INFINITY;          // thresholds are kept WITHOUT the query's norm: values are |r|^2 - 2 q.r
#if KNN_GTAU
  if (q < q_end) {               // the seed of the pre-pass (knn_seed_kernel), +inf without one
    int best = gtau[q - q_begin];
    best ^= (best >> 31) & 0x7fffffff;
    tau = __int_as_float(best);
  }
#endif

  const int64_t ntiles = (n + BR - 1) / BR;
  // ref range `sp` = the tiles sp, sp + nsplit, sp + 2 nsplit, ...: INTERLEAVED, not a contiguous block of refs.  Data often comes
  // sorted (by class, along a curve, by a locality order): a query's neighbours are then neighbours in index too, a contiguous
  // range would put all of them into the two lists of one range and overflow them (29 % of the rows of locality-ordered
  // config-4 data took the exact fallback); interleaved, any 32 * nsplit consecutive refs are spread over all the lists
  const int t1 = (int)ntiles;         // (tile numbers fit 32 bits -- ref indices do, cand_i is int --: scalar compares instead of 64-bit vector ones)
  // the tile iterator: tiles congruent to sp modulo nsplit, of all tiles or of the block's runs (wave-uniform arithmetic)
  int run = -1, nrun = 0;
  int run_b = 0;
  const int* myruns = nullptr;
  if constexpr (RUNS) {
    myruns = runs + qb * 2 * (int64_t)maxruns;
    nrun = nruns[qb];
  }
  auto next_tile = [&](int tc) -> int {       // t1 (or beyond) = no further tile
    int tn = tc + nsplit;
    if constexpr (RUNS) {
      while (tn >= run_b) {
        if (++run >= nrun) return t1;
        const int a = myruns[2 * run];
        run_b = myruns[2 * run + 1];
        tn = a + ((int)sp - a % nsplit + nsplit) % nsplit;
      }
    }
    return tn;
  };
  const int t0 = RUNS ? next_tile(-nsplit) : (int)sp;
  uint4 pre[UNITS];
  float pre_rn = 0.f;
  // the lane's first 16-byte unit within a tile, +2048: with -2048 in the instruction the 13-bit signed offset field reaches the unit
  // at +4096 as well (a scalar tile base + this 32-bit lane offset + an immediate; opaque to the compiler, which would fold it back)
  unsigned lane_off = (unsigned)(tid * 16 + 2048);       // (the OFFSET is made opaque, not the pointer: a laundered pointer loses its
  asm volatile("" : "+v"(lane_off));                     //  address space and the loads become flat_load, which LDS waits then wait for)
  auto stage_load = [&](int64_t t) {
    // no bounds predicates: Xb / nrm carry KNN_PAD_ROWS spare rows (zero features, norm 1e30) behind the data
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = tid + i * 256;
      const int r = u / U_ROW, c = u % U_ROW;
      uint4 v = {0u, 0u, 0u, 0u};
      (void)r; (void)c;
      // (the rows of a tile are contiguous: a wave-uniform tile base + a 32-bit lane offset, no 64-bit vector address arithmetic)
      if (EXACT_UNITS || u < BR * U_ROW) v = *(const uint4*)((const char*)Xb + t * (int64_t)(BR * 4 * KPAD) + (size_t)lane_off + (i * 4096 - 2048));
      pre[i] = v;
    }
    if (CAT != 2 && tid < BR) {       // (CAT == 2: the norm is part of the contraction)
      const int64_t ref = t * BR + tid;
      pre_rn = nrm[ref];
    }
  };
  auto stage_store = [&](int buf) {
    char* dst = tile + buf * BR * ROWB;
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int u = tid + i * 256;
      if (EXACT_UNITS || u < BR * U_ROW) *(uint4*)(dst + (u / U_ROW) * ROWB + (u % U_ROW) * 16) = pre[i];
    }
    if (CAT != 2 && tid < BR) rn[buf * BR + tid] = pre_rn;
  };
  int cnt = 0;
  float tau_own = INFINITY;
  int pmax = 0;
#if KNN_COUNT
  unsigned long long knn_ev[6] = {0, 0, 0, 0, 0, 0};
#endif
  unsigned long long cy_slow = 0, cy_comp = 0, cy_bar = 0, cy_all = 0, cy_store = 0;
  (void)cy_slow; (void)cy_comp; (void)cy_bar; (void)cy_all; (void)cy_store;
  auto compact = [&]() {
    KNN_TIC(tc);
#if KNN_COUNT
    int mx = cnt;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
#endif
#if KNN_COUNT
    {
      int tot = cnt;
      for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
      KNN_CNT(3, tot);
      KNN_CNT(4, 1);
      KNN_CNT(5, mx);
    }
#endif
    for (int a = 0; __any(a < cnt); ++a) {      // (a ballot per step instead of a cross-lane maximum up front: 6 ds_bpermute round trips)
      if (a < cnt) {
        const float v = ld[(KP + a) * 256 + tid];
        if (v < tau_own) {
          if constexpr (REGL) {
            // the candidate replaces the (first) largest entry; select chains instead of indexed LDS accesses
            const int vi = li[(KP + a) * 256 + tid];
            bool placed = false;
            float m2 = -INFINITY;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
              const bool hit = !placed && lv[p] == tau_own;
              lv[p] = hit ? v : lv[p];
              lx[p] = hit ? vi : lx[p];
              placed = placed || hit;
              m2 = fmaxf(m2, lv[p]);
            }
            tau_own = m2;
          } else {
            ld[pmax * 256 + tid] = v;
            li[pmax * 256 + tid] = li[(KP + a) * 256 + tid];
            float m2 = ld[tid];
            int pm = 0;
#pragma unroll
            for (int p = 1; p < KP; ++p) {
              const float x = ld[p * 256 + tid];
              if (x > m2) { m2 = x; pm = p; }
            }
            tau_own = m2;
            pmax = pm;
          }
        }
      }
    }
    cnt = 0;
    // lanes l and l^32 serve the same query (same |q|^2 offset); never above what is already known (the seed, published thresholds)
    const float tau_was = tau;
    tau = fminf(tau, fminf(tau_own, __shfl_xor(tau_own, 32)));
#if KNN_GTAU
    // the query's lists of the OTHER ref ranges run in other workgroups: the smallest threshold any of them has reached is
    // published per query (an ordered-int image of the float, atomicMin) and adopted here.  Sound for the same reason the pair's
    // minimum is: whatever a list rejects lies above the smallest FINAL threshold of the query's lists, which is what the
    // acceptance test of the re-rank compares with the exact k-th distance.
    if (tau < tau_was && q < q_end) {       // (only a threshold that moved: the atomic's round trip is a stall of the whole wavefront)
      int key = __float_as_int(tau);
      key ^= (key >> 31) & 0x7fffffff;
      const int old = atomicMin(&gtau[q - q_begin], key);
      int best = min(old, key);
      best ^= (best >> 31) & 0x7fffffff;
      tau = fminf(tau, __int_as_float(best));
    }
#endif
    KNN_TOC(cy_comp, tc);
  };
  if (!(KNN_DIRECT && CAT == 2 && NSTG == 1)) {
    if (t0 < t1) { stage_load(t0); stage_store(0); }
    __syncthreads();
  }
  int buf = 0;
#if KNN_ABLATE & 1
  float abl_sink = INFINITY;
#endif
  KNN_TIC(ta);
  int it = 0;
  // DIRECT (build option, norm-folded concatenated form only): no LDS tiles and no barrier -- every wavefront fetches the A fragments of
  // the next tile from global memory (L1 / L2: the four wavefronts of a workgroup walk the same tiles) into registers while it
  // contracts the current one
  constexpr bool DIRECT = KNN_DIRECT && CAT == 2 && NSTG == 1;
  constexpr int NF = NSUB * NKB * 2;
  uint4 fa[DIRECT ? NF : 1], fb[DIRECT ? NF : 1];
  auto direct_load = [&](int64_t tt, uint4 (&dst)[DIRECT ? NF : 1]) {
    if constexpr (DIRECT) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          // (Xb here is the FRAGMENT image written by knn_frag_layout_kernel: 64 lanes x 16 bytes contiguous per fragment)
          const uint4* fp = (const uint4*)Xb + (((tt * NSUB + sub) * NKB + kb) * 2) * 64 + lane;
          dst[(sub * NKB + kb) * 2 + 0] = fp[0];
          dst[(sub * NKB + kb) * 2 + 1] = fp[64];
        }
    }
  };
  int t = t0, tn = -1;
  if constexpr (DIRECT) { if (t0 < t1) direct_load(t0, fa); }
  auto tile_body = [&](uint4 (&cur)[DIRECT ? NF : 1], uint4 (&nxt)[DIRECT ? NF : 1]) {
    tn = next_tile(t);
    const bool has_next = tn < t1;
#if KNN_GTAU
    if ((it & 15) == 15) {        // (every lane reads -- rows past q_end their clamped query's --: a scalar branch, no exec-mask bookkeeping per tile)
      int best = gtau[qc - q_begin];
      best ^= (best >> 31) & 0x7fffffff;
      tau = fminf(tau, __int_as_float(best));
    }
#endif
#if !(KNN_ABLATE & 4)
    if constexpr (DIRECT) { if (has_next) direct_load(tn, nxt); }
    else { if (has_next) stage_load(tn); }
#endif
#pragma unroll 1
    for (int stg = 0; stg < NSTG; ++stg) {
    const char* tl = tile + (buf * BR + stg * 32 * NSUB) * ROWB;
    const float* rnb = rn + buf * BR + stg * 32 * NSUB;
    f32x16 acc[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[sub][e] = 0.f;     // (the first MFMA of a chain takes the constant 0 as its C operand)
    // the tile's norms, read IN FRONT of the contraction: behind it (where they are used) every one of the 4 NSUB reads was a
    // round trip of its own -- ds_read_b128, s_waitcnt lgkmcnt(0), four fmas, next read -- with the matrix pipe idle
    float4 r4s[(CAT != 2 && KNN_NORMS_FIRST) ? NSUB : 1][4];
    if constexpr (CAT != 2 && KNN_NORMS_FIRST) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int eg = 0; eg < 4; ++eg) r4s[sub][eg] = *(const float4*)(rnb + sub * 32 + 8 * eg + 4 * h);
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        // A[i][k]: lane holds row i = j of the sub-tile, k = 8h .. 8h+7 of block kb
        const char* rowp = tl + (sub * 32 + j) * ROWB + (kb * 16 + 8 * h) * 2;
        bf16x8 ah, al;
        if constexpr (DIRECT) {
          ah = __builtin_bit_cast(bf16x8, cur[(sub * NKB + kb) * 2 + 0]);
          al = __builtin_bit_cast(bf16x8, cur[(sub * NKB + kb) * 2 + 1]);
        } else {
          ah = __builtin_bit_cast(bf16x8, *(const uint4*)rowp);
          al = __builtin_bit_cast(bf16x8, *(const uint4*)(rowp + 2 * KPAD));
        }
        if constexpr (CAT) {     // fragments kb and 2 + kb of the one concatenated contraction
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kb], acc[sub], 0, 0, 0);
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl[kb], acc[sub], 0, 0, 0);
        } else {
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kb], acc[sub], 0, 0, 0);
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kb], acc[sub], 0, 0, 0);
          acc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kb], acc[sub], 0, 0, 0);
        }
      }
    }
    // selection on value = |r|^2 - 2 q.r (element e of sub-tile `sub` is ref sub*32 + (e&3) + 8*(e>>2) + 4h, query j): per group
    // of 4 elements an extremum first, so that groups without a candidate in any lane cost one compare (with 64 lanes per
    // wavefront SOME lane has a candidate in almost every tile)
    float m4[NSUB][4];
    float m = INFINITY;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int eg = 0; eg < 4; ++eg) {
        if constexpr (CAT == 2) {     // the accumulator already is |r|^2 - 2 q.r
          // (written as ONE chain ending in +inf: two v_min3_f32 on the raw accumulators.  A two-input minimum of raw MFMA results
          // costs a v_max_f32 x, x per input first -- the compiler quiets possible signalling NaNs for v_min_f32, not for
          // v_min3_f32: 37 -> 20 vector instructions per wave-tile for this reduction)
          m4[sub][eg] = fminf(fminf(fminf(fminf(acc[sub][eg * 4 + 0], acc[sub][eg * 4 + 1]), acc[sub][eg * 4 + 2]), acc[sub][eg * 4 + 3]), INFINITY);
          continue;
        }
        const float4 r4 = KNN_NORMS_FIRST ? r4s[KNN_NORMS_FIRST ? sub : 0][eg] : *(const float4*)(rnb + sub * 32 + 8 * eg + 4 * h);
#if KNN_PACKED_SELECT
        // two fp32 fmas per instruction (v_pk_fma_f32 on adjacent accumulator registers) and three-input minima (v_min3_f32):
        // the same values, half the vector instructions of the per-element form -- at K <= 64 the selection, not the
        // contraction, is what the tile kernel waits for
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 m2 = {-2.f, -2.f};
        f32x2 a01 = {acc[sub][eg * 4 + 0], acc[sub][eg * 4 + 1]}, a23 = {acc[sub][eg * 4 + 2], acc[sub][eg * 4 + 3]};
        a01 = __builtin_elementwise_fma(m2, a01, f32x2{r4.x, r4.y});
        a23 = __builtin_elementwise_fma(m2, a23, f32x2{r4.z, r4.w});
        acc[sub][eg * 4 + 0] = a01.x;
        acc[sub][eg * 4 + 1] = a01.y;
        acc[sub][eg * 4 + 2] = a23.x;
        acc[sub][eg * 4 + 3] = a23.y;
        m4[sub][eg] = fminf(fminf(fminf(a01.x, a01.y), a23.x), a23.y);
#else
        acc[sub][eg * 4 + 0] = fmaf(-2.f, acc[sub][eg * 4 + 0], r4.x);
        acc[sub][eg * 4 + 1] = fmaf(-2.f, acc[sub][eg * 4 + 1], r4.y);
        acc[sub][eg * 4 + 2] = fmaf(-2.f, acc[sub][eg * 4 + 2], r4.z);
        acc[sub][eg * 4 + 3] = fmaf(-2.f, acc[sub][eg * 4 + 3], r4.w);
        m4[sub][eg] = fminf(fminf(acc[sub][eg * 4 + 0], acc[sub][eg * 4 + 1]), fminf(acc[sub][eg * 4 + 2], acc[sub][eg * 4 + 3]));
#endif
        m = fminf(m, m4[sub][eg]);
      }
    if constexpr (CAT == 2) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) m = fminf(fminf(fminf(fminf(m, m4[sub][0]), m4[sub][1]), m4[sub][2]), m4[sub][3]);
    }
#if KNN_ABLATE & 1
    abl_sink = fminf(abl_sink, m);  // developer probe: no list maintenance (the minimum is kept alive: without a use the
    if (false) {                    // compiler removes the whole contraction, as the first version of this probe found out)
#else
    KNN_CNT(0, 1);
    KNN_TIC(ts);
    if (__any(m < tau)) {
#endif
      KNN_CNT(1, 1);
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
        for (int eg = 0; eg < 4; ++eg) {
          if (!KNN_GROUP_GUARD || __any(m4[sub][eg] < tau)) {
            KNN_CNT(2, 1);
#pragma unroll
            for (int e = eg * 4; e < eg * 4 + 4; ++e) {
              const float v = acc[sub][e];
              if (v < tau) {
                ld[(KP + cnt) * 256 + tid] = v;
                li[(KP + cnt) * 256 + tid] = (int)(t * BR) + (stg * NSUB + sub) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                ++cnt;
              }
            }
            if (__any(cnt > KBUF - 4)) compact();
          }
        }
      }
    }
    KNN_TOC(cy_slow, ts);
    }   // stg
    KNN_TIC(tb);
    if constexpr (!DIRECT) {
    if (has_next) stage_store(buf ^ 1);
#if KNN_COUNT == 2
    __builtin_amdgcn_s_waitcnt(0);      // (probe only) the store's own waits end here, the rest is the barrier
    KNN_TOC(cy_store, tb);
#endif
#if !(KNN_ABLATE & 2)
    __syncthreads();
#endif
    }
    KNN_TOC(cy_bar, tb);
    buf ^= 1;
    t = tn;
    ++it;
  };
  while (t < t1) {
    tile_body(fa, fb);
    if constexpr (DIRECT) {
      if (t >= t1) break;
      tile_body(fb, fa);
    }
  }
  KNN_TOC(cy_all, ta);
#if KNN_ABLATE & 1
  if (abl_sink == 12345.f) cand_d[0] = abl_sink;
#endif
#if KNN_COUNT
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_knn_cnt[i], knn_ev[i]);
    atomicAdd(&g_knn_cnt[8], cy_all); atomicAdd(&g_knn_cnt[9], cy_slow); atomicAdd(&g_knn_cnt[10], cy_comp); atomicAdd(&g_knn_cnt[11], cy_bar); atomicAdd(&g_knn_cnt[12], cy_store);
  }
#endif
  compact();
  if (q < q_end) {
    const int64_t lists = (int64_t)gridDim.y * 2;
    const int64_t base = ((q - q_begin) * lists + sp * 2 + h) * KP;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      if constexpr (REGL) {
        cand_d[base + p] = lv[p] + qn;                 // back to squared distances (inf stays inf)
        cand_i[base + p] = lx[p];
      } else {
        cand_d[base + p] = ld[p * 256 + tid] + qn;
        cand_i[base + p] = li[p * 256 + tid];
      }
    }
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
// trip to the host: key = the cell's place in the chain, a histogram per block of 256 rows, one block scanning the (block, key) table,
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

__global__ __launch_bounds__(256) void knn_cellrank_scan_kernel(int* __restrict__ bh, int nb, int m) {
  extern __shared__ int tot_[];
  for (int c = threadIdx.x; c < m; c += 256) {
    int run = 0;
    for (int b = 0; b < nb; ++b) {
      const int t = bh[(int64_t)b * m + c];
      bh[(int64_t)b * m + c] = run;
      run += t;
    }
    tot_[c] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int c = 0; c < m; ++c) { const int t = tot_[c]; tot_[c] = run; run += t; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += 256) {
    const int base = tot_[c];
    for (int b = 0; b < nb; ++b) bh[(int64_t)b * m + c] += base;
  }
}

__global__ __launch_bounds__(256) void knn_cellrank_scatter_kernel(const int* __restrict__ key, int64_t n, int m, const int* __restrict__ bh,
                                                                   int* __restrict__ perm) {
  __shared__ int k_[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kk = i < n ? key[i] : -1;
  k_[threadIdx.x] = kk;
  __syncthreads();
  if (i < n) {
    int r = 0;
    for (int j = 0; j < (int)threadIdx.x; ++j) r += (k_[j] == kk) ? 1 : 0;
    perm[bh[(int64_t)blockIdx.x * m + kk] + r] = (int)i;
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

static const int CELL_SPLIT = 64;      // workgroups per cell in the centre / radius passes (a cell of config 4 at n = 1e7 is 80 MB)

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
  // Exact distances only where they can matter: with v_k the k-th smallest FILTER value of the candidates, the exact k-th
  // distance^2 is at most v_k + eps, and a candidate with a filter value above v_k + 2 eps is at least v_k + eps away -- farther
  // than the k-th.  Of 128 candidates a dozen or two remain; the others' rows (d doubles each, scattered over X) are never
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
  const double keep = prefilter ? (double)s_vk + 2.0 * eps0 + 1e-6 * fabs((double)s_vk) : INFINITY;
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
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        if (stride >= 64) {                       // partners in the same lane
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
        } else {                                  // partners a lane exchange away
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int lo = __shfl_xor(__double2loint(rd[r]), stride), hi = __shfl_xor(__double2hiint(rd[r]), stride);
            const double od = __hiloint2double(hi, lo);
            const int oi = __shfl_xor(ri[r], stride);
            const bool up = (((lane + 64 * r) & size) == 0), lower = (lane & stride) == 0;
            const bool mine_first = lex_less(rd[r], ri[r], od, oi);
            const bool take_min = lower == up;
            const bool keep_mine = take_min ? mine_first : !mine_first;
            rd[r] = keep_mine ? rd[r] : od;
            ri[r] = keep_mine ? ri[r] : oi;
          }
        }
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

// ---- stage 3: exact fp64 fallback for flagged rows --------------------------------------------
// A flagged row's refs are split over FB_SPLIT workgroups (a single one would read the whole data set k times: 50 ms per
// row at n = 1e7); each finds the k smallest (dist, idx) of its piece by k rounds of "smallest pair lexicographically greater
// than the last one picked", a second kernel merges the pieces' ascending lists.
static const int FB_SPLIT = 64;
static const int FB_CACHE = 2048;     // a piece of at most this many refs keeps its distances in LDS between the rounds

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
static const int FB_CAP = 128;
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

// features per half per block of the blocked (d > 130) variant; 16 where the KP = 64 lists leave less LDS
constexpr int knn_kb(int KP) { return KP == 64 ? 16 : 32; }   // (KP = 8 never takes the blocked variant)

// refs per tile = 32*NSUB, as many as fit LDS (160 KiB) beside the candidate lists
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

struct KnnBufs {
  unsigned short* Xb = nullptr;      // bf16 hi | lo image (bf16 filter); concatenated form: the ref image [hi | hi | lo]
  unsigned short* Xq = nullptr;      // concatenated form only: the query image [hi | lo | hi]
  unsigned short* Xf = nullptr;      // (KNN_DIRECT experiment) second ref image
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
  // glx_knn_clustered: the rows reordered by cell (X points at the reordered copy), orig[position] = the caller's row
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
    glx_pool_free(Xb); glx_pool_free(Xq); glx_pool_free(Xf); glx_pool_free(nrm); glx_pool_free(part); glx_pool_free(rmax);
    glx_pool_free(X); glx_pool_free(mean); glx_pool_free(dist); glx_pool_free(Rf); glx_pool_free(Qf); glx_pool_free(qnorm); glx_pool_free(cand_d);
    glx_pool_free(runs); glx_pool_free(nruns); glx_pool_free(cell_starts); glx_pool_free(cen); glx_pool_free(rad); glx_pool_free(ub2); glx_pool_free(cpart); glx_pool_free(mask); glx_pool_free(visited); glx_pool_free(Xraw); glx_pool_free(orig); glx_pool_free(cell_id);
    glx_pool_free(pre_d); glx_pool_free(pre_i); glx_pool_free(gtau); glx_pool_free(cand_i); glx_pool_free(flags); glx_pool_free(rows); glx_pool_free(ind); glx_pool_free(fb_pi); glx_pool_free(fb_pd); glx_pool_free(dk2); glx_pool_free(fb_bd); glx_pool_free(fb_cnt); glx_pool_free(fb_bi); glx_pool_free(nbad); glx_pool_free(place); glx_pool_free(bh);
    glx_work_release(work);
  }
};


// ---- centring on the device: column means and the largest centred norm without a host round trip.
// Round 2's column-sum kernel gave each of d threads a 1024-long strided chain (20 of 256 threads active at d = 20: 0.39 ms
// for 11 MB at config 2, a quarter of the tile kernel); now the 256 threads of a workgroup tile its CENTRE_ROWS x d slab as
// (rows in flight) x (columns side by side), every thread sums its column over its rows in a register, LDS combines the row
// lanes in a fixed order, and one more small kernel folds the block partials -- in block order -- into the mean.  The largest centred norm is
// reduced on the device too; the re-rank kernel reads it from memory, the host looks at it (is the input finite?) together
// with the acceptance flags at the end.  Any FIXED summation order serves: the mean only centres the filter's operands,
// distances come from the uncentred fp64 data.
static const int CENTRE_ROWS = 512;
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

#ifdef KNN_BF16_NSUB
constexpr int bf16_nsub(int NKB, int KP) { return (NKB >= 6 || KP >= 32) ? 1 : KNN_BF16_NSUB; }
#else
// refs per tile = 32 * NSUB.  Measured (one box): 16-32 features: NSUB 2 (config 2: 1.88 vs 2.12 ms, config 3: 2.74 vs 3.09 ms);
// 64 features: NSUB 1 -- a 17 KB tile lets three workgroups share a CU (n = 1e6: 376 vs 401 ms)
constexpr int bf16_nsub(int NKB, int KP) { return (NKB >= 4 || KP >= 32) ? 1 : 2; }
#endif

// nsplit ref ranges with a tile stride of nsplit (the search proper), or -- seed = true -- ONE range with a stride of nsplit
// writing the pre-pass's own two lists per query (KnnBufs::pre_d / pre_i)
#ifndef KNN_BF16_NSTG
#define KNN_BF16_NSTG 1
#endif
// sub-tiles per barrier (knn_tile_bf16_kernel).  Measured (round 3, gpurun_out/r03af): 2 -> config 2 1.32 -> 1.58 ms, config 3
// 1.88 -> 3.20 ms, 4 -> 2.45 ms: the doubled tile buffers cost the third workgroup per CU, which matters more than the barrier
constexpr int bf16_nstg(int NKB, int KP) { return (NKB <= 2 && KP <= 16) ? KNN_BF16_NSTG : 1; }

template <int NKB, int KP, int CAT = 0>
static int launch_tile_bf16(const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, bool seed = false) {
  constexpr int NSUB = bf16_nsub(NKB, KP);
  constexpr int NSTG = bf16_nstg(NKB, KP);
  constexpr int BR = 32 * NSUB * NSTG;
  constexpr int ROWB = 4 * 16 * NKB + 16;
  const size_t shm = (size_t)2 * BR * ROWB + (size_t)2 * BR * 4 + (size_t)((KNN_REGLISTS && KP == 8 && (NKB == 4 || (KNN_REGL2 && NKB == 2))) ? KBUF : KP + KBUF) * 256 * 8;
  GLX_CHECK(shm <= 160 * 1024, GLX_EUNSUPPORTED, "glx_knn_bruteforce: bf16 filter needs %zu bytes of LDS", shm);
  const dim3 grid((unsigned)((q1 - q0 + BQ - 1) / BQ), (unsigned)(seed ? 1 : nsplit));
  if (b.runs) {
    GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, true, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, true, NSTG>), grid, dim3(256), shm, st, (const unsigned short*)b.Xb, (const unsigned short*)(CAT ? b.Xq : b.Xb),
                       (const float*)b.nrm, n, q0, q1, nsplit, seed ? b.pre_d : b.cand_d, seed ? b.pre_i : b.cand_i, b.gtau, (const int*)b.runs,
                       (const int*)b.nruns, b.maxruns);
  } else {
    GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, false, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, false, NSTG>), grid, dim3(256), shm, st, (const unsigned short*)b.Xb, (const unsigned short*)(CAT ? b.Xq : b.Xb),
                       (const float*)b.nrm, n, q0, q1, nsplit, seed ? b.pre_d : b.cand_d, seed ? b.pre_i : b.cand_i, b.gtau, (const int*)nullptr,
                       (const int*)nullptr, 0);
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <int KP>
static int launch_tile_bf16_nkb(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat = 0,
                                bool seed = false) {
  if (cat == 2) return launch_tile_bf16<2, KP, 2>(b, n, q0, q1, nsplit, st, seed);
  if (cat) return launch_tile_bf16<2, KP, 1>(b, n, q0, q1, nsplit, st, seed);
  switch (NKB) {
    case 1: return launch_tile_bf16<1, KP>(b, n, q0, q1, nsplit, st, seed);
    case 2: return launch_tile_bf16<2, KP>(b, n, q0, q1, nsplit, st, seed);
    case 4: return launch_tile_bf16<4, KP>(b, n, q0, q1, nsplit, st, seed);
    case 6: return launch_tile_bf16<6, KP>(b, n, q0, q1, nsplit, st, seed);
    case 8: return launch_tile_bf16<8, KP>(b, n, q0, q1, nsplit, st, seed);
  }
  glx_set_error("knn: no bf16 tile kernel for %d feature blocks", NKB);
  return GLX_EUNSUPPORTED;
}

template <int DH, int KP, bool KBLK = false>
static int launch_tile(const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int nkb = 1) {
  constexpr int DPA = 2 * DH;
  constexpr int STRIDE = (DH % 2 == 1) ? DPA : DPA + 2;
  constexpr int NSUB = tile_nsub(DH, KP);
  const size_t shm = (size_t)2 * 32 * NSUB * STRIDE * 4 + (size_t)(KP + KBUF) * 256 * 8;
  GLX_CHECK(shm <= 160 * 1024, GLX_EUNSUPPORTED, "glx_knn_bruteforce: this (d, k) needs %zu bytes of LDS per workgroup (160 KiB available)", shm);
  GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_kernel<DH, KP, NSUB, KBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const dim3 grid((unsigned)((q1 - q0 + BQ - 1) / BQ), (unsigned)nsplit);
  hipLaunchKernelGGL((knn_tile_kernel<DH, KP, NSUB, KBLK>), grid, dim3(256), shm, st, (const float*)b.Rf, (const float*)b.Qf, n, q0, q1,
                     nsplit, b.cand_d, b.cand_i, getenv("GLX_KNN_ABLATE") ? atoi(getenv("GLX_KNN_ABLATE")) : 0, nkb);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <int KP>
static int launch_tile_dh(int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st) {
  if (nkb > 1) return launch_tile<knn_kb(KP), KP, true>(b, n, q0, q1, nsplit, st, nkb);
  switch (DH) {
    case 8: return launch_tile<8, KP>(b, n, q0, q1, nsplit, st);
    case 12: return launch_tile<12, KP>(b, n, q0, q1, nsplit, st);
    case 18: return launch_tile<18, KP>(b, n, q0, q1, nsplit, st);
    case 34: return launch_tile<34, KP>(b, n, q0, q1, nsplit, st);
    case 66: return launch_tile<66, KP>(b, n, q0, q1, nsplit, st);
  }
  glx_set_error("knn: no tile kernel for %d features per half", DH);
  return GLX_EUNSUPPORTED;
}

static const int KNN_ESCALATE = 1;    // knn_pass: too many rows failed the acceptance test of the short lists -- search again with long ones

// One pass of the search.  long_lists = false: the default (short lists where they apply); if then so many query rows fail
// the acceptance test that repairing them row by row -- each streams the whole data set k times -- would take longer than
// searching again, KNN_ESCALATE is returned: the caller repeats the search with the long lists (one list
// holds all k neighbours of a query, whatever their arrangement in the data).  It takes data whose k nearest neighbours
// sit in the same 16 of 32 consecutive points to get there (tight groups stored one after another); interleaving the ref tiles
// over the ranges already spreads anything coarser.
// glx_knn_retain_next: the next FULL search (all rows as queries) keeps its neighbour indices on the device for the assembly that
// follows it (glx_knn_to_csr with ind = NULL adopts them), and may be called with ind_out = NULL -- weightmatrix.knn's own flow,
// where the lists never need to visit the host (2 x 6 MB over PCIe at config 2)
// (per calling thread: request, search and assembly are three calls of ONE thread -- another thread's search must neither take the
// request nor replace what is retained)
static thread_local int g_knn_keep_next = 0;
// glx_knn_search: the search in progress on this thread hands its device-resident lists (and its cell order) to this result
// instead of freeing them -- set and cleared inside that one call
static thread_local glx_knn_result* g_knn_capture = nullptr;
static thread_local struct { int64_t* ind; int64_t n; int k; int device; } g_knn_kept = {nullptr, 0, 0, 0};

extern "C" int glx_knn_retain_next(int on) {
  g_knn_keep_next = on ? 1 : 0;
  if (!on && g_knn_kept.ind) {
    glx_pool_free(g_knn_kept.ind);
    g_knn_kept.ind = nullptr;
  }
  return GLX_OK;
}

int glx_knn_take_retained(int64_t n, int k, int device, int64_t** ind_dev) {
  if (!g_knn_kept.ind || g_knn_kept.n != n || g_knn_kept.k != k || g_knn_kept.device != device) {
    glx_set_error("glx_knn_to_csr: ind = NULL, but no search result of %lld x %d indices is retained on device %d (glx_knn_retain_next)",
                  (long long)n, k, device);
    return GLX_EINVAL;
  }
  *ind_dev = g_knn_kept.ind;
  g_knn_kept.ind = nullptr;
  return GLX_OK;
}

static int knn_pass(const double* X, int64_t n, int d, int k, int64_t q0, int64_t q1, int64_t* ind_out, double* dist_out, int device,
                    bool long_lists, const int64_t* cell_starts = nullptr, int ncells = 0, int auto_cells = 0) {
  glx_knn_result* capture = (q0 == 0 && q1 == n) ? g_knn_capture : nullptr;
  const bool keep_ind = !capture && g_knn_keep_next && q0 == 0 && q1 == n;
  GLX_CHECK(X && ((ind_out && dist_out) || (keep_ind && dist_out) || capture), GLX_EINVAL, "glx_knn_bruteforce: null argument");
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
  const bool short_lists = !long_lists && d + 2 <= 132 && !(getenv("GLX_KNN_SHORT") && atoi(getenv("GLX_KNN_SHORT")) == 0);
  if (short_lists) KP = k <= 12 ? 8 : (k <= 28 ? 16 : 32);
  int DH = knn_kb(KP), nkb = 1;
  if (d + 2 <= 132 && !(KP == 64 && d + 2 > 36)) {   // (KP = 64 lists + a wide double-buffered tile exceed the LDS)
    for (int cand : {8, 12, 18, 34, 66})
      if (2 * cand >= d + 2) { DH = cand; break; }
  } else {
    nkb = (d + 2 + 2 * DH - 1) / (2 * DH);
  }
  // Filter arithmetic.  Default: split-bf16 operands on the bf16 matrix cores (d <= 128 with the short lists); the fp32-input
  // MFMA kernel serves everything else and GLX_KNN_FILTER=f32.
  const char* fenv = getenv("GLX_KNN_FILTER");
  const bool use_bf16 = short_lists && d <= 128 && KP <= 32 && !(fenv && strcmp(fenv, "f32") == 0);
  int NKB = 0;
  if (use_bf16) {
    for (int cand : {1, 2, 4, 6, 8})
      if (16 * cand >= d) { NKB = cand; break; }
  }
  const int dpa = use_bf16 ? 16 * NKB : 2 * DH * nkb;
  const int64_t nqb = (nq + BQ - 1) / BQ;
  const int BR = use_bf16 ? 32 * bf16_nsub(NKB, KP) * bf16_nstg(NKB, KP) : 32 * tile_nsub(DH, KP);
  const int64_t ntiles = (n + BR - 1) / BR;
  int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, ntiles), (1024 + nqb - 1) / nqb));
  if (short_lists) {
    // >= 8 lists per query; 16 for the 8-entry lists once the data no longer sits in cache (an exact
    // fallback row then streams all of X k times: 329 rows cost 0.6 s at n = 2e6 -- with 16 lists 5 rows are left)
    const int64_t want = (KP == 8 && (double)n * d * 8.0 > 64.0 * 1024 * 1024) ? 8 : 4;
    nsplit = (int)std::max<int64_t>(nsplit, std::min<int64_t>(want, ntiles));
  }
  if (const char* e = getenv("GLX_KNN_NSPLIT")) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, ntiles), atoi(e)));
  const int lists = nsplit * 2;
  const int ncand = lists * KP;
  int M = 64;
  while (M < ncand) M *= 2;

  // (host buffers of the cell-order by-product: declared in front of `b`, whose destructor drains the stream they are filled through)
  std::vector<int> oc_sample, oc_cid, oc_perm;
  std::vector<double> oc_cen;
  std::vector<int> oc_place;
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
  GLX_HIP(hipMemcpyAsync(b.X, X, (size_t)n * d * 8, hipMemcpyHostToDevice, st));
  stamp("X enqueued");
  // glx_knn_clustered: form auto_cells cells (nearest of that many sample rows), reorder the rows by cell ON THE DEVICE and
  // search with the cell pruning of glx_knn_cells_range; the re-rank ranks by and returns the caller's indices
  std::vector<int64_t> own_starts;
  // auto_cells < -1, since the end of round 3: the rows ARE reordered by -auto_cells chained cells and then searched all pairs.  The
  // 32 queries of a wavefront then come from one corner of feature space, a ref tile holds candidates for many of them or for none,
  // and fewer wave-tiles leave the tile kernel's fast path: config 2 2.06 -> 1.83 ms of search wall time, config 3's shape
  // 3.06 -> 2.63 ms, data without clusters unchanged (profiles/r03_knn_cells_midsize.txt).  GLX_KNN_REORDER=0: the order alone,
  // worked out on a side stream behind an all-pairs search in the caller's order (what was there before).
  const bool order_only = auto_cells < -1 && getenv("GLX_KNN_REORDER") && atoi(getenv("GLX_KNN_REORDER")) == 0;
  const bool reorder_only = auto_cells < -1 && !order_only;
  if (auto_cells < -1) auto_cells = -auto_cells;
  int oc_m = 0;
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
  auto finish_order = [&]() {
    const int m = oc_m;
    std::vector<int>& cid = oc_cid;
    const std::vector<int> place = chain_places(oc_cen, m);
    for (int64_t i = 0; i < n; ++i) cid[i] = place[cid[i]];
    own_starts.assign(m, 0);
    std::vector<int64_t> fill(m + 1, 0);
    for (int64_t i = 0; i < n; ++i) ++fill[cid[i] + 1];
    for (int c = 0; c < m; ++c) fill[c + 1] += fill[c];
    for (int c = 0; c < m; ++c) own_starts[c] = fill[c];
    oc_perm.resize(n);
    for (int64_t i = 0; i < n; ++i) oc_perm[fill[cid[i]]++] = (int)i;
    std::lock_guard<std::mutex> lk(g_knn_order_mu);
    g_knn_last_order.assign(oc_perm.begin(), oc_perm.end());
  };
  bool order_pending = false, perm_pending = false;
  if (auto_cells > 1 && q0 == 0 && q1 == n && !long_lists && d <= 128 && n >= 4 * (int64_t)auto_cells) {
    const int m = oc_m = auto_cells;
    oc_sample.resize(m);
    for (int c = 0; c < m; ++c) oc_sample[c] = (int)(((2 * (int64_t)c + 1) * n) / (2 * (int64_t)m));     // evenly spaced rows
    GLX_POOL(glx_pool_alloc((void**)&b.cen, (size_t)m * d * 8));
    GLX_POOL(glx_pool_alloc((void**)&b.cell_id, (size_t)std::max<int64_t>(n, m) * 4));
    // order only: beside the search, on the work set's second stream (it needs the uploaded rows and nothing else)
    hipStream_t so = order_only ? b.work->side : st;
    if (order_only) {
      GLX_HIP(hipEventRecord(b.work->ev_side, st));
      GLX_HIP(hipStreamWaitEvent(so, b.work->ev_side, 0));
    }
    GLX_HIP(hipMemcpyAsync(b.cell_id, oc_sample.data(), (size_t)m * 4, hipMemcpyHostToDevice, so));
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)(((int64_t)m * d + 255) / 256)), dim3(256), 0, so, (const double*)b.X, (const int*)b.cell_id,
                       (int64_t)m, d, b.cen);
    hipLaunchKernelGGL(knn_assign_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), (size_t)16 * d * 8, so, (const double*)b.X, d, n, (const double*)b.cen, m,
                       b.cell_id, (d + 31) / 32);
    GLX_HIP(hipGetLastError());
    if (reorder_only && !getenv("GLX_KNN_REORDER_HOST")) {     // (GLX_KNN_REORDER_HOST: the host's counting sort, for A/B runs)
      // the chain of the cells from the caller's copy of the sample rows (the same doubles the device gathered), the rows into cell
      // order by the three knn_cellrank kernels: nothing here waits for the device (the host sort cost 0.2 - 0.4 ms of waiting --
      // for the upload's tail, the cell ids, the permutation's way back)
      oc_cen.resize((size_t)m * d);
      for (int c = 0; c < m; ++c) memcpy(&oc_cen[(size_t)c * d], X + (size_t)oc_sample[c] * d, (size_t)d * 8);
      oc_place = chain_places(oc_cen, m);
      const int nb = (int)((n + 255) / 256);
      GLX_POOL(glx_pool_alloc((void**)&b.place, (size_t)m * 4));
      GLX_POOL(glx_pool_alloc((void**)&b.bh, (size_t)nb * m * 4));
      GLX_POOL(glx_pool_alloc((void**)&b.orig, (size_t)n * 4));
      GLX_HIP(hipMemcpyAsync(b.place, oc_place.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(knn_cellrank_hist_kernel, dim3((unsigned)nb), dim3(256), (size_t)m * 4, st, b.cell_id, (const int*)b.place, n, m, b.bh);
      hipLaunchKernelGGL(knn_cellrank_scan_kernel, dim3(1), dim3(256), (size_t)m * 4, st, b.bh, nb, m);
      hipLaunchKernelGGL(knn_cellrank_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, st, (const int*)b.cell_id, n, m, (const int*)b.bh, b.orig);
      b.Xraw = b.X;
      b.X = nullptr;
      GLX_POOL(glx_pool_alloc((void**)&b.X, (size_t)n * d * 8));
      hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, st, (const double*)b.Xraw, (const int*)b.orig, n, d, b.X);
      GLX_HIP(hipGetLastError());
      perm_pending = true;                             // the permutation comes back with the results (glx_knn_last_order)
      stamp("rows reordered by cell (on the device)");
    } else {
    oc_cid.resize(n);
    oc_cen.resize((size_t)m * d);
    GLX_HIP(hipMemcpyAsync(oc_cid.data(), b.cell_id, (size_t)n * 4, hipMemcpyDeviceToHost, so));
    GLX_HIP(hipMemcpyAsync(oc_cen.data(), b.cen, (size_t)m * d * 8, hipMemcpyDeviceToHost, so));
    if (order_only) {
      order_pending = true;                          // finished behind the search (finish_order at the end of the pass)
    } else {
    GLX_HIP(hipStreamSynchronize(st));
    finish_order();
    std::vector<int>& perm = oc_perm;
    GLX_POOL(glx_pool_alloc((void**)&b.orig, (size_t)n * 4));
    GLX_HIP(hipMemcpyAsync(b.orig, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    b.Xraw = b.X;
    b.X = nullptr;
    GLX_POOL(glx_pool_alloc((void**)&b.X, (size_t)n * d * 8));
    hipLaunchKernelGGL(knn_gather_rows_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, st, (const double*)b.Xraw, (const int*)b.orig, n, d, b.X);
    GLX_HIP(hipGetLastError());
    // (no synchronisation: `perm` = oc_perm outlives the stream's work -- it is declared in front of `b`, whose destructor drains the
    // stream -- and the kernels that read b.cen finished before the synchronisation in front of finish_order)
    glx_pool_free(b.cen);                          // the cell pass allocates its own
    b.cen = nullptr;
    cell_starts = own_starts.data();
    ncells = m;
    if (reorder_only) {          // the rows in cell order, then all pairs: no pre-pass, no pruning
      cell_starts = nullptr;
      ncells = 0;
    }
    stamp("rows reordered by cell");
    }
    }
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
  // bf16 filter: the dropped parts of the split products (lo.lo and the residuals, <= 3.1 * 2^-18 |q||r| in q.r, twice that in the
  // distance, |q||r| <= (|q|+rmax)^2 / 4), 3*kpad fp32 accumulations, fp32 norms and input rounding -- all of it doubled
  // (the matrix pipe's internal rounding mode is not documented).
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
  if (use_bf16) {
    // 17 <= d <= 21 (two blocks of 16 per half): the three split products as ONE contraction over concatenated operands,
    // 4 MFMAs per 32 x 32 tile instead of 6 (d <= 16 needs 3 either way)
    // ... and for d <= 20 with the norm folded in (GLX_KNN_CAT=1: without the fold, =0: blocks of 16 features)
    int cat = (d <= KNN_CAT_SEG && NKB == 2) ? (d < KNN_CAT_SEG ? 2 : 1) : 0;
    if (const char* e = getenv("GLX_KNN_CAT")) cat = std::min(cat, atoi(e));
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
#if KNN_DIRECT
    if (cat == 2) {          // (experiment) the tile kernel of this form reads the fragment image
      const int nsub = bf16_nsub(NKB, KP);
      const int64_t nt = (n + KNN_PAD_ROWS) / (32 * nsub);
      GLX_POOL(glx_pool_alloc((void**)&b.Xf, (size_t)nt * 32 * nsub * 2 * dpa * 2));
      const int64_t units = nt * nsub * NKB * 2 * 64;
      hipLaunchKernelGGL(knn_frag_layout_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, (const unsigned short*)b.Xb, n + KNN_PAD_ROWS, dpa, nsub, NKB,
                         b.Xf, nt);
      GLX_HIP(hipGetLastError());
      std::swap(b.Xb, b.Xf);   // (Xq keeps the row layout for the queries)
    }
#endif
    g_knn_stats[9] = (double)cat;
    // The seeding pre-pass (knn_seed_kernel).  Over all refs it does not pay (measured, profiles/r03_knn_seed.txt: the k-th of a
    // 1/8 sample is the 8k-th of the whole set, 79 % of the wave-tiles still hold a candidate and the pre-pass costs its eighth):
    // GLX_KNN_SEED=<sample stride> turns it on for experiments.  The cell-pruned search needs it: its bound ub2 decides which
    // cells a query block visits.
    const bool cells = cell_starts != nullptr && ncells > 1;
    int seed_sub = 0;
    if (const char* e = getenv("GLX_KNN_SEED")) seed_sub = atoi(e);
    if (cells) {
      // sample the block's own cells: every tile of small cells, every 8th of cells of >= 128 tiles
      const int64_t avg_tiles = std::max<int64_t>(1, ntiles / ncells);
      seed_sub = (int)std::max<int64_t>(1, std::min<int64_t>(8, avg_tiles / 16));
    }
    const bool seeded = (cells || (seed_sub > 1 && ntiles >= (int64_t)64 * seed_sub)) && 2 * KP >= k;
    g_knn_stats[10] = seeded ? (double)seed_sub : 0.0;
    g_knn_stats[11] = 0.0;
    g_knn_stats[12] = 0.0;
    if (cells && seeded) {
      GLX_POOL(glx_pool_alloc((void**)&b.cell_starts, (size_t)ncells * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.cen, (size_t)ncells * d * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.rad, (size_t)ncells * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.ub2, (size_t)nq * 8));
      GLX_POOL(glx_pool_alloc((void**)&b.mask, (size_t)nqb * ncells));
      GLX_POOL(glx_pool_alloc((void**)&b.nruns, (size_t)nqb * 4));
      b.maxruns = ncells;
      GLX_POOL(glx_pool_alloc((void**)&b.runs, (size_t)nqb * 2 * ncells * 4));   // (from here on the tile launches follow the runs)
      GLX_HIP(hipMemcpyAsync(b.cell_starts, cell_starts, (size_t)ncells * 8, hipMemcpyHostToDevice, st));
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
    }
    if (seeded) {
      GLX_POOL(glx_pool_alloc((void**)&b.pre_d, (size_t)nq * 2 * KP * 4));
      GLX_POOL(glx_pool_alloc((void**)&b.pre_i, (size_t)nq * 2 * KP * 4));
      if (KP == 8) rc = launch_tile_bf16_nkb<8>(NKB, b, n, q0, q1, seed_sub, st, cat, true);
      else if (KP == 16) rc = launch_tile_bf16_nkb<16>(NKB, b, n, q0, q1, seed_sub, st, cat, true);
      else rc = launch_tile_bf16_nkb<32>(NKB, b, n, q0, q1, seed_sub, st, cat, true);
      if (rc) return rc;
      const dim3 sg((unsigned)((nq + 255) / 256));
      if (KP == 8)
        hipLaunchKernelGGL(knn_seed_kernel<16>, sg, dim3(256), 0, st, (const float*)b.pre_d, (const int*)b.pre_i, nq, q0, k, (const float*)b.qnorm,
                           (const float*)b.nrm, (const float*)b.rmax, cerr, b.gtau, b.ub2);
      else if (KP == 16)
        hipLaunchKernelGGL(knn_seed_kernel<32>, sg, dim3(256), 0, st, (const float*)b.pre_d, (const int*)b.pre_i, nq, q0, k, (const float*)b.qnorm,
                           (const float*)b.nrm, (const float*)b.rmax, cerr, b.gtau, b.ub2);
      else
        hipLaunchKernelGGL(knn_seed_kernel<64>, sg, dim3(256), 0, st, (const float*)b.pre_d, (const int*)b.pre_i, nq, q0, k, (const float*)b.qnorm,
                           (const float*)b.nrm, (const float*)b.rmax, cerr, b.gtau, b.ub2);
      GLX_HIP(hipGetLastError());
    }
    if (cells && seeded) {
      hipLaunchKernelGGL(knn_cellmask_kernel, dim3((unsigned)nqb), dim3(256), (size_t)16 * d * 8, st, (const double*)b.X, d, q0, q1, (const double*)b.cen,
                         (const double*)b.rad, ncells, (const double*)b.ub2, b.mask);
      GLX_POOL(glx_pool_alloc((void**)&b.visited, 8));
      GLX_HIP(hipMemsetAsync(b.visited, 0, 8, st));
      hipLaunchKernelGGL(knn_runs_kernel, dim3((unsigned)((nqb + 255) / 256)), dim3(256), 0, st, (const unsigned char*)b.mask,
                         (const int64_t*)b.cell_starts, n, ncells, BR, q0, q1, nqb, b.maxruns, b.runs, b.nruns, b.visited);
      GLX_HIP(hipGetLastError());
      g_knn_stats[12] = (double)ncells;
    }
    if (KP == 8) rc = launch_tile_bf16_nkb<8>(NKB, b, n, q0, q1, nsplit, st, cat);
    else if (KP == 16) rc = launch_tile_bf16_nkb<16>(NKB, b, n, q0, q1, nsplit, st, cat);
    else rc = launch_tile_bf16_nkb<32>(NKB, b, n, q0, q1, nsplit, st, cat);
  } else {
    GLX_POOL(glx_pool_alloc((void**)&b.Rf, (size_t)n * dpa * 4));
    GLX_POOL(glx_pool_alloc((void**)&b.Qf, (size_t)n * dpa * 4));
    hipLaunchKernelGGL(knn_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)b.X, (const double*)b.mean,
                       n, d, dpa, b.Rf, b.Qf, b.qnorm);
    GLX_HIP(hipGetLastError());
    if (KP == 8) rc = launch_tile_dh<8>(DH, nkb, b, n, q0, q1, nsplit, st);
    else if (KP == 16) rc = launch_tile_dh<16>(DH, nkb, b, n, q0, q1, nsplit, st);
    else if (KP == 32) rc = launch_tile_dh<32>(DH, nkb, b, n, q0, q1, nsplit, st);
    else rc = launch_tile_dh<64>(DH, nkb, b, n, q0, q1, nsplit, st);
  }
  if (rc) return rc;
  GLX_HIP(hipEventRecord(b.e1, st));
#if KNN_COUNT
  if (use_bf16) {
    unsigned long long c[16];
    GLX_HIP(hipStreamSynchronize(st));
    GLX_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_knn_cnt), sizeof(c)));
    fprintf(stderr, "knn counters: wave-tiles %llu, with a candidate %llu (%.1f %%), active groups %llu, appended %llu, compactions %llu, compaction steps %llu; nsplit %d\n",
            c[0], c[1], 100.0 * c[1] / (double)std::max(1ull, c[0]), c[2], c[3], c[4], c[5], nsplit);
    if (c[8])
      fprintf(stderr, "knn cycles (sum over waves): tile loop %.3g = 100 %%, threshold test + appends + merges %.1f %% (merges alone %.1f %%), staging store + barrier %.1f %% (store and its waits %.1f %%); per wave-tile %.0f cycles\n",
              (double)c[8], 100.0 * c[9] / (double)c[8], 100.0 * c[10] / (double)c[8], 100.0 * c[11] / (double)c[8], 100.0 * c[12] / (double)c[8], (double)c[8] / (double)std::max(1ull, c[0]));
    memset(c, 0, sizeof(c));
    GLX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_cnt), c, sizeof(c)));
  }
#endif
  static const int prefilter_from = getenv("GLX_KNN_PREFILTER_D") ? atoi(getenv("GLX_KNN_PREFILTER_D")) : 32;
#define GLX_RERANK(RR)                                                                                                                    \
  hipLaunchKernelGGL(knn_rerank_kernel<RR>, dim3((unsigned)nq), dim3(64), (size_t)M * 16, st, (const double*)b.X, n, d, k, q0, nq,          \
                     (const float*)b.cand_d, (const int*)b.cand_i, lists, KP, M, (const float*)b.qnorm, (const float*)b.rmax, cerr, b.ind, b.dist, \
                     b.flags, (const int*)b.orig, d >= prefilter_from ? 1 : 0, b.dk2, b.nbad, b.rows)
  if (M == 64) GLX_RERANK(1);
  else if (M == 128) GLX_RERANK(2);
  else if (M == 256) GLX_RERANK(4);
  else if (M == 512) GLX_RERANK(8);
  else GLX_RERANK(0);
#undef GLX_RERANK
  GLX_HIP(hipGetLastError());
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
  if (KNN_ABLATE) rows.n = 0;     // developer probes produce wrong candidate lists: do not repair them
  if (short_lists && rows.size() > 64) {
    // repair row by row, or search again with the long lists?  A fallback row streams the data once (measured: ~5 TB/s);
    // the repeat costs about four tile-kernel times (fp32-input filter, longer lists)
    float ms_first = 0;
    GLX_HIP(hipEventElapsedTime(&ms_first, b.e0, b.e1));
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
    int* redo = getenv("GLX_KNN_FALLBACK_ROUNDS") ? nullptr : b.fb_cnt + nr;       // (the k-round kernels alone: a developer switch)
    if (redo) {
      // one pass: every ref within the bound the re-rank left, ranked by a wavefront per row
      hipLaunchKernelGGL(knn_fallback_collect_kernel, dim3((unsigned)nr, FB_SPLIT), dim3(256), 0, st, (const double*)b.X, n, d, q0, (const int*)b.rows,
                         (const double*)b.dk2, b.fb_cnt, b.fb_bd, b.fb_bi, (const int*)b.orig, fb_runs, (const int*)b.nruns, b.maxruns, BR);
      hipLaunchKernelGGL(knn_fallback_select_kernel, dim3((unsigned)nr), dim3(64), 0, st, (const int*)b.fb_cnt, (const double*)b.fb_bd, (const int*)b.fb_bi,
                         (const int*)b.rows, (int)nr, k, b.ind, b.dist, (const int*)b.orig, q0, redo);
    }
    // the k-round kernels: only the rows the one pass could not finish (their workgroups return at once otherwise)
    hipLaunchKernelGGL(knn_fallback_piece_kernel, dim3((unsigned)nr, FB_SPLIT), dim3(256), 0, st, (const double*)b.X, n, d, k, q0, (const int*)b.rows,
                       b.fb_pd, b.fb_pi, (const int*)b.orig, fb_runs, (const int*)b.nruns, b.maxruns, BR, (const int*)redo);
    hipLaunchKernelGGL(knn_fallback_merge_kernel, dim3((unsigned)nr), dim3(64), 0, st, (const double*)b.fb_pd, (const int*)b.fb_pi,
                       (const int*)b.rows, (int)nr, k, b.ind, b.dist, (const int*)b.orig, q0, (const int*)redo);
    GLX_HIP(hipGetLastError());
  }
  GLX_HIP(hipEventRecord(b.e3, st));
  if (ind_out) GLX_HIP(hipMemcpyAsync(ind_out, b.ind, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
  if (dist_out) GLX_HIP(hipMemcpyAsync(dist_out, b.dist, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  stamp("results on the host");
  if (order_pending) {
    GLX_HIP(hipStreamSynchronize(b.work->side));
    finish_order();
    stamp("cell order worked out");
  }
  if (perm_pending) {
    oc_perm.resize(n);
    GLX_HIP(hipMemcpy(oc_perm.data(), b.orig, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lk(g_knn_order_mu);
    g_knn_last_order.assign(oc_perm.begin(), oc_perm.end());
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
    std::lock_guard<std::mutex> lk(g_knn_order_mu);
    capture->order = g_knn_last_order;
  }
  if (keep_ind) {        // (everything that writes b.ind has finished: the stream was synchronised above)
    if (g_knn_kept.ind) glx_pool_free(g_knn_kept.ind);
    g_knn_kept.ind = b.ind;
    g_knn_kept.n = n;
    g_knn_kept.k = k;
    g_knn_kept.device = device;
    b.ind = nullptr;
    g_knn_keep_next = 0;                       // one search
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
                   const int64_t* cell_starts = nullptr, int ncells = 0, int auto_cells = 0) {
  g_knn_stats[8] = 0.0;
  g_knn_stats[10] = g_knn_stats[11] = g_knn_stats[12] = 0.0;
  {
    std::lock_guard<std::mutex> lk(g_knn_order_mu);
    g_knn_last_order.clear();            // (an order on record always belongs to the search that ran last)
  }
  int rc = knn_pass(X, n, d, k, q0, q1, ind_out, dist_out, device, false, cell_starts, ncells, auto_cells);
  if (rc != KNN_ESCALATE) return rc;
  const double flagged = g_knn_stats[2];
  g_knn_stats[10] = g_knn_stats[11] = g_knn_stats[12] = 0.0;
  rc = knn_pass(X, n, d, k, q0, q1, ind_out, dist_out, device, true);     // (long lists: the fp32-input kernel, all refs)
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
  return knn_run(X, n, d, k, q_begin, q_end, ind_out, dist_out, device, cell_starts, ncells);
}

// All n rows in the caller's order, the cells formed here: ncells evenly spaced rows serve as centres, every row joins the
// nearest one, the rows are reordered by cell on the device and searched with the pruning of glx_knn_cells_range; indices and
// output rows are the caller's, ties between equal distances go to the lower caller index -- the lists of glx_knn_bruteforce,
// bit for bit.  On data without cluster structure every cell stays in play and the extra passes cost a few per cent.
extern "C" int glx_knn_clustered(const double* X, int64_t n, int d, int k, int ncells, int64_t* ind_out, double* dist_out, int device) {
  // ncells < -1: the all-pairs search, with the order of -ncells chained cells left for glx_knn_last_order as a by-product
  GLX_CHECK(ncells >= -4096 && ncells <= 4096, GLX_EINVAL, "glx_knn_clustered: ncells=%d outside [-4096, 4096]", ncells);
  return knn_run(X, n, d, k, 0, n, ind_out, dist_out, device, nullptr, 0, ncells);
}

// perm_out[position] = caller's row in the cell order of the last glx_knn_clustered search over n rows (GLX_EINVAL if there is none
// of that size): contiguous, chained cells of feature space -- on clustered data as good a locality order for the graph's operators
// as the library's own pass over the graph (glx_graph_set_order), and free.
extern "C" int glx_knn_last_order(int64_t n, int32_t* perm_out) {
  GLX_CHECK(perm_out, GLX_EINVAL, "glx_knn_last_order: null output");
  std::lock_guard<std::mutex> lk(g_knn_order_mu);
  GLX_CHECK((int64_t)g_knn_last_order.size() == n && n > 0, GLX_EINVAL, "glx_knn_last_order: no clustered search over %lld rows on record", (long long)n);
  memcpy(perm_out, g_knn_last_order.data(), (size_t)n * 4);
  return GLX_OK;
}

// ---- search results as objects -------------------------------------------------------------------------------------------------
// glx_knn_search runs the full search (every row a query) and leaves the lists ON THE DEVICE in a result object the caller owns:
// glx_knn_result_to_csr (assemble.hip) builds the weight matrix from them without a host round trip, glx_knn_result_lists copies
// them out, glx_knn_result_order returns the cell order the search worked out (if it did), glx_knn_result_destroy releases
// everything.  Nothing is handed from one call to the next through hidden state.
extern "C" int glx_knn_search(const double* X, int64_t n, int d, int k, int ncells, int device, glx_knn_result** out) {
  GLX_CHECK(out, GLX_EINVAL, "glx_knn_search: null output");
  *out = nullptr;
  GLX_CHECK(ncells >= -4096 && ncells <= 4096, GLX_EINVAL, "glx_knn_search: ncells=%d outside [-4096, 4096]", ncells);
  glx_knn_result* res = new glx_knn_result();
  g_knn_capture = res;
  const int rc = knn_run(X, n, d, k, 0, n, nullptr, nullptr, device, nullptr, 0, (ncells > 1 || ncells < -1) ? ncells : 0);
  g_knn_capture = nullptr;
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
  if (ind_out) GLX_HIP(hipMemcpy(ind_out, res->ind, bytes, hipMemcpyDeviceToHost));
  if (dist_out) GLX_HIP(hipMemcpy(dist_out, res->dist, bytes, hipMemcpyDeviceToHost));
  return GLX_OK;
}

extern "C" int glx_knn_result_order(const glx_knn_result* res, int32_t* perm_out) {
  GLX_CHECK(res && perm_out, GLX_EINVAL, "glx_knn_result_order: null argument");
  GLX_CHECK((int64_t)res->order.size() == res->n && res->n > 0, GLX_EINVAL, "glx_knn_result_order: this search worked out no cell order");
  memcpy(perm_out, res->order.data(), (size_t)res->n * 4);
  return GLX_OK;
}

extern "C" int glx_knn_result_destroy(glx_knn_result* res) {
  if (!res) return GLX_OK;
  glx_pool_free(res->ind);
  glx_pool_free(res->dist);
  delete res;
  return GLX_OK;
}
