// The fp32-input candidate filter of the exact kNN search (v_mfma_f32_32x32x2_f32): every (d, k) the split-bf16 filter does not
// take -- d > 128, lists of 64, the long lists of an escalated search, GLX_KNN_FILTER=f32.  Included by knn_tile_f32_*.hip, which
// instantiate it per list length (see knn.hip; reference graphlearning/weightmatrix.py:297-429).
#pragma once
#include "knn_internal.h"

// ---- stage 1: MFMA tile kernel -------------------------------------------------------------
// KBLK = false: the whole (padded) feature vector of a query lives in registers (DH features per
// half, d + 2 <= 2*DH <= 132).  KBLK = true (any d): the features are processed in nkb blocks of
// DH per half; each step stages one feature block of the ref tile into LDS, reloads the lane's
// query fragment for that block (prefetched one step ahead) and accumulates into the same MFMA
// accumulators; the selection runs after the last block of a tile.
template <int DH, int KP, int NSUB, bool KBLK>
__global__ __launch_bounds__(256) void knn_tile_kernel(const float* __restrict__ Rf, const float* __restrict__ Qf, int64_t n,
                                                       int64_t q_begin, int64_t q_end, int nsplit, float* __restrict__ cand_d,
                                                       int* __restrict__ cand_i, int nkb_arg) {
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
    tau = fminf(tau_own, __shfl_xor(tau_own, 32));
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
    if (__any(m < tau)) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
        for (int eg = 0; eg < 16; eg += 4) {
#pragma unroll
          for (int e = eg; e < eg + 4; ++e) {
            const float v = acc[sub][e];
            if (v < tau) {
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

template <int DH, int KP, bool KBLK = false>
static int launch_tile_f32(const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int nkb = 1) {
  constexpr int DPA = 2 * DH;
  constexpr int STRIDE = (DH % 2 == 1) ? DPA : DPA + 2;
  constexpr int NSUB = tile_nsub(DH, KP);
  const size_t shm = (size_t)2 * 32 * NSUB * STRIDE * 4 + (size_t)(KP + KBUF) * 256 * 8;
  GLX_CHECK(shm <= 160 * 1024, GLX_EUNSUPPORTED, "glx_knn_bruteforce: this (d, k) needs %zu bytes of LDS per workgroup (160 KiB available)", shm);
  GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_kernel<DH, KP, NSUB, KBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const dim3 grid((unsigned)((q1 - q0 + BQ - 1) / BQ), (unsigned)nsplit);
  hipLaunchKernelGGL((knn_tile_kernel<DH, KP, NSUB, KBLK>), grid, dim3(256), shm, st, (const float*)b.Rf, (const float*)b.Qf, n, q0, q1,
                     nsplit, b.cand_d, b.cand_i, nkb);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// DH features per half in registers (d + 2 <= 2 DH <= 132), or nkb > 1 blocks of knn_kb(KP)
template <int KP>
static int launch_tile_f32_kp(int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st) {
  if (nkb > 1) {
    if constexpr (KP == 8) {
      glx_set_error("knn: the 8-entry lists have no blocked fp32 tile kernel");
      return GLX_EUNSUPPORTED;
    } else {
      return launch_tile_f32<knn_kb(KP), KP, true>(b, n, q0, q1, nsplit, st, nkb);
    }
  }
  switch (DH) {
    case 8: return launch_tile_f32<8, KP>(b, n, q0, q1, nsplit, st);
    case 12: return launch_tile_f32<12, KP>(b, n, q0, q1, nsplit, st);
    case 18: return launch_tile_f32<18, KP>(b, n, q0, q1, nsplit, st);
    case 34: return launch_tile_f32<34, KP>(b, n, q0, q1, nsplit, st);
    case 66: return launch_tile_f32<66, KP>(b, n, q0, q1, nsplit, st);
  }
  glx_set_error("knn: no tile kernel for %d features per half", DH);
  return GLX_EUNSUPPORTED;
}
