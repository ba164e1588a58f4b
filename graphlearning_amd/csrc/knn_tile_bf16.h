// The split-bf16 candidate filter of the exact kNN search (v_mfma_f32_32x32x16_bf16): d <= 128 with lists of 8 / 16 / 32 entries,
// the default at configs 2-4.  Included by knn_tile_bf16_k*.hip, which instantiate it per list length (see knn.hip; reference
// graphlearning/weightmatrix.py:297-429).
//
// The filter only has to be accurate to a KNOWN bound (the re-rank is exact fp64 and its acceptance test allows for the
// bound), so the contraction does not need fp32 operands: every centred coordinate x is split into two bfloat16 numbers,
// x = hi + lo + e with |e| <= 2^-16 |x| (bf16 carries 8 significant bits: each rounding leaves at most 2^-8 of what it rounds),
// and q.r is formed as hi.hi + hi.lo + lo.hi -- three v_mfma_f32_32x32x16_bf16
// per 16 features, accumulated in fp32 -- at 16x the rate of the f32-input MFMA: 96 matrix-pipe cycles per 16 features of
// a 32 x 32 tile instead of 512.  bf16 products are exact in fp32; what is dropped (lo.lo and the e terms) is below
// 3.1 * 2^-16 |q||r| in the WORST case (every rounding error of every coordinate at its bound and of one sign; measured on the
// inputs of the randomised suite: at most 0.52 of `cerr`'s bound), which the acceptance test of the re-rank allows for TWICE over.  The squared norms stay fp32 and are added after the
// contraction (two extra features would lose them to bf16): value = |r|^2 - 2 q.r, compared with tau - |q|^2.
#pragma once
#include "knn_internal.h"
#include <type_traits>

// NKB blocks of 16 features (kpad = 16 NKB <= 128); refs are the A operand (LDS), queries the B operand (registers: lane =
// query column j, k-half h).  Every lane owns one query column and keeps the KP best of its half of the refs of its range in
// an unsorted list with the maximum tracked; candidates below the lane's threshold are APPENDED to the lane's LDS slots
// (cheap, even when only a few lanes have one) and merged by the whole wavefront in lockstep when any lane's slots run low.
// CAT (NKB = 2 only): the rows are the concatenated operands of knn_prep_bf16_cat_kernel, refs from Xb, queries from Xq.
// (Round 5 measured a variant with TWO sets of 32 queries per wavefront -- one ref-fragment read feeding two MFMAs, 16 MFMAs per barrier,
// both sets' lists in registers: identical lists, 18-25 % SLOWER, profiles/r05_knn_tile_pmc.txt.  The experiment is closed; the variant
// is kept as scripts/probes/knn_two_sets_experiment.patch.)
template <int NKB, int KP, int NSUB, int CAT = 0, bool RUNS = false>   // CAT: 1 concatenated operands, 2 also the norm folded into them
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(bf16_reglists(NKB, KP) ? 4 : 1, 4)))
void knn_tile_bf16_kernel(const unsigned short* __restrict__ Xb, const unsigned short* __restrict__ Xq, const float* __restrict__ nrm, int64_t n,
                          int64_t q_begin, int64_t q_end, int nsplit, float* __restrict__ cand_d, int* __restrict__ cand_i,
                          int* __restrict__ gtau, const int* __restrict__ runs, const int* __restrict__ nruns, int maxruns) {
  // RUNS (the cell-pruned search, glx_knn_cells_range): this query block visits only the ref tiles of its runs
  // [runs[2 r], runs[2 r + 1]), r < nruns[block] (ascending, disjoint; knn_runs_kernel), not all of them
  // nsplit = the tile stride of a ref range; the number of ranges is the grid's y extent (equal in the search proper; the
  // seeding pre-pass runs ONE range with a larger stride: every 8th tile, say -- a sample of the refs)
  constexpr int KPAD = 16 * NKB;
  constexpr int BR = 32 * NSUB;
  constexpr int ROWB = 4 * KPAD + 16;                  // bytes per ref row in LDS: hi | lo, +16 so that 16 rows cover all 64 banks
  constexpr int U_ROW = 4 * KPAD / 16;                 // 16-byte units per row
  constexpr int UNITS = (BR * U_ROW + 255) / 256;
  constexpr bool EXACT_UNITS = (BR * U_ROW) % 256 == 0;
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  char* tile = smem_b;                                  // [2][BR][ROWB]
  float* rn = (float*)(smem_b + 2 * BR * ROWB);         // [2][BR]
  constexpr bool REGL = bf16_reglists(NKB, KP);       // the lists in registers: LDS then holds tile + append slots only
  constexpr int LROWS = REGL ? KBUF : KP + KBUF;
  // slot a: ld[(KP + a) * 256 + tid] (rows [0, KP) do not exist with register lists)
  float* ld = rn + 2 * BR - (REGL ? KP * 256 : 0);
  int* li = (int*)(rn + 2 * BR + LROWS * 256) - (REGL ? KP * 256 : 0);
  float lv[REGL ? 8 : 1];
  int lx[REGL ? 8 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, j = lane & 31;
  const int64_t qb = blockIdx.x, sp = blockIdx.y;
  int64_t q, qc;
  // query fragments: B[k][j], lane holds k = 8h .. 8h+7 of every block, hi and lo
  bf16x8 bh[NKB], bl[NKB];
  float qn;
  {
    q = q_begin + qb * BQ + wave * 32 + j;
    qc = q < q_end ? q : q_end - 1;
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
    qn = nrm[qc];
    asm volatile("" ::"v"(qn));
  }
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      if constexpr (REGL) { lv[p] = INFINITY; lx[p] = -1; }
      else { ld[p * 256 + tid] = INFINITY; li[p * 256 + tid] = -1; }
    }
  // thresholds are kept WITHOUT the query's norm: values are |r|^2 - 2 q.r.  gtau[query]: the smallest threshold any of the
  // query's lists has published (an ordered-int image of the float) -- at the start the seed of the pre-pass (knn_seed_kernel),
  // +inf without one
  auto gtau_read = [&](int64_t qq) -> float {
    int best = gtau[qq - q_begin];
    best ^= (best >> 31) & 0x7fffffff;
    return __int_as_float(best);
  };
  float tau = INFINITY;
  if (q < q_end) tau = gtau_read(q);

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
  // All pairs: the range's tiles are walked from the one nearest the block's own rows, wrapping around at the end.  When the rows
  // come in a locality order (weightmatrix.knn reorders them by chained cells) a query's neighbours sit near its own index: the
  // thresholds are tight after the first few tiles instead of after half the refs, and what follows rarely appends (tile kernel
  // 0.90 -> 0.76 ms at config 2, 1.38 -> 1.13 ms at config 3, identical lists; profiles/r04_knn_near_start.txt).
  int left = 0;                        // (all pairs) tiles of the range still to visit
  int t0;
  if constexpr (RUNS) {
    t0 = next_tile(-nsplit);
  } else {
    left = t1 > (int)sp ? (t1 - (int)sp + nsplit - 1) / nsplit : 0;
    const int own = (int)((q_begin + qb * BQ) / BR);          // the tile in the middle of the block's rows
    int j0 = own > (int)sp ? (own - (int)sp + nsplit - 1) / nsplit : 0;
    if (j0 >= left) j0 = 0;
    t0 = (int)sp + j0 * nsplit;
  }
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
      uint4 v = {0u, 0u, 0u, 0u};
      // (the rows of a tile are contiguous: a wave-uniform tile base + a 32-bit lane offset, no 64-bit vector address arithmetic)
      if (EXACT_UNITS || u < BR * U_ROW) v = *(const uint4*)((const char*)Xb + t * (int64_t)(BR * 4 * KPAD) + (size_t)lane_off + (i * 4096 - 2048));
      pre[i] = v;
    }
    if (CAT != 2 && tid < BR) pre_rn = nrm[t * BR + tid];       // (CAT == 2: the norm is part of the contraction)
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
  auto compact = [&]() {
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
    // the query's lists of the OTHER ref ranges run in other workgroups: the smallest threshold any of them has reached is
    // published per query (atomicMin on the ordered-int image) and adopted here.  Sound for the same reason the pair's
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
  };
  // The staging pipeline is two tiles deep: at the top of the iteration of tile t the registers hold tile t+1 (loaded during the
  // iteration of t-1, a whole tile of matrix work ago) and go to the other LDS buffer at once -- everyone left it at the barrier
  // that ended t-1 --, then the loads of tile t+2 are issued.  (One tile deep -- load at the top, store at the bottom -- the store
  // waited for its loads and the barrier for the store: 190 + 210 of ~1500 cycles per tile, profiles/r04_knn_tile_pmc.txt.)
  int t = t0;
  int tn;
  bool has_next;
  if constexpr (RUNS) {
    if (t0 < t1) { stage_load(t0); stage_store(0); }
    tn = t0 < t1 ? next_tile(t0) : t1;
    has_next = tn < t1;
  } else {
    if (left > 0) { stage_load(t0); stage_store(0); }
    tn = t0 + nsplit;
    if (tn >= t1) tn = (int)sp;
    has_next = left > 1;
  }
  if (has_next) stage_load(tn);
  __syncthreads();
  int buf = 0;
  int it = 0;
  while (RUNS ? t < t1 : left > 0) {
    int tnn;
    bool has_nn;
    if constexpr (RUNS) {
      tnn = has_next ? next_tile(tn) : t1;
      has_nn = tnn < t1;
    } else {
      tnn = tn + nsplit;
      if (tnn >= t1) tnn = (int)sp;
      has_nn = left > 2;
      --left;
    }
    if (has_next) stage_store(buf ^ 1);
    if (has_nn) stage_load(tnn);
    // (every lane reads -- rows past q_end their clamped query's --: a scalar branch, no exec-mask bookkeeping per tile)
    if ((it & 15) == 15) {
      tau = fminf(tau, gtau_read(qc));
    }
    const char* tl = tile + buf * BR * ROWB;
    const float* rnb = rn + buf * BR;
    f32x16 acc[NSUB];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[sub][e] = 0.f;     // (the first MFMA of a chain takes the constant 0 as its C operand)
    // the tile's norms, read IN FRONT of the contraction: behind it (where they are used) every one of the 4 NSUB reads was a
    // round trip of its own -- ds_read_b128, s_waitcnt lgkmcnt(0), four fmas, next read -- with the matrix pipe idle
    float4 r4s[CAT != 2 ? NSUB : 1][4];
    if constexpr (CAT != 2) {
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
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)rowp);
        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(rowp + 2 * KPAD));
        {
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
    }
    // selection on value = |r|^2 - 2 q.r (element e of sub-tile `sub` is ref sub*32 + (e&3) + 8*(e>>2) + 4h, query j): per group
    // of 4 elements an extremum first, so that groups without a candidate in any lane cost one compare (with 64 lanes per
    // wavefront SOME lane has a candidate in almost every tile)
    auto select = [&]() {
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
          } else {
            const float4 r4 = r4s[sub][eg];
            acc[sub][eg * 4 + 0] = fmaf(-2.f, acc[sub][eg * 4 + 0], r4.x);
            acc[sub][eg * 4 + 1] = fmaf(-2.f, acc[sub][eg * 4 + 1], r4.y);
            acc[sub][eg * 4 + 2] = fmaf(-2.f, acc[sub][eg * 4 + 2], r4.z);
            acc[sub][eg * 4 + 3] = fmaf(-2.f, acc[sub][eg * 4 + 3], r4.w);
            m4[sub][eg] = fminf(fminf(acc[sub][eg * 4 + 0], acc[sub][eg * 4 + 1]), fminf(acc[sub][eg * 4 + 2], acc[sub][eg * 4 + 3]));
            m = fminf(m, m4[sub][eg]);
          }
        }
      if constexpr (CAT == 2) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) m = fminf(fminf(fminf(fminf(m, m4[sub][0]), m4[sub][1]), m4[sub][2]), m4[sub][3]);
      }
      if (__any(m < tau)) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
          for (int eg = 0; eg < 4; ++eg) {
            if (__any(m4[sub][eg] < tau)) {       // (the guard per group of four: 438 -> 394 ms at n = 1e6)
#pragma unroll
              for (int e = eg * 4; e < eg * 4 + 4; ++e) {
                const float v = acc[sub][e];
                if (v < tau) {
                  ld[(KP + cnt) * 256 + tid] = v;
                  li[(KP + cnt) * 256 + tid] = t * BR + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                  ++cnt;
                }
              }
              if (__any(cnt > KBUF - 4)) compact();
            }
          }
        }
      }
    };
    select();
    __syncthreads();
    buf ^= 1;
    t = RUNS ? (has_next ? tn : t1) : tn;
    tn = tnn;
    has_next = has_nn;
    ++it;
  }
  compact();
  {
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
}

template <int NKB, int KP, int CAT = 0>
static int launch_tile_bf16(const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, bool seed) {
  constexpr int NSUB = bf16_nsub(NKB, KP);
  constexpr int BR = 32 * NSUB;
  constexpr int ROWB = 4 * 16 * NKB + 16;
  const size_t shm = (size_t)2 * BR * ROWB + (size_t)2 * BR * 4 + (size_t)(bf16_reglists(NKB, KP) ? KBUF : KP + KBUF) * 256 * 8;
  GLX_CHECK(shm <= 160 * 1024, GLX_EUNSUPPORTED, "glx_knn_bruteforce: bf16 filter needs %zu bytes of LDS", shm);
  const dim3 grid((unsigned)((q1 - q0 + BQ - 1) / BQ), (unsigned)(seed ? 1 : nsplit));
  if (b.runs) {
    GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, true>), grid, dim3(256), shm, st, (const unsigned short*)b.Xb, (const unsigned short*)(CAT ? b.Xq : b.Xb),
                       (const float*)b.nrm, n, q0, q1, nsplit, seed ? b.pre_d : b.cand_d, seed ? b.pre_i : b.cand_i, b.gtau, (const int*)b.runs,
                       (const int*)b.nruns, b.maxruns);
  } else {
    GLX_HIP(hipFuncSetAttribute((const void*)knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((knn_tile_bf16_kernel<NKB, KP, NSUB, CAT, false>), grid, dim3(256), shm, st, (const unsigned short*)b.Xb, (const unsigned short*)(CAT ? b.Xq : b.Xb),
                       (const float*)b.nrm, n, q0, q1, nsplit, seed ? b.pre_d : b.cand_d, seed ? b.pre_i : b.cand_i, b.gtau, (const int*)nullptr,
                       (const int*)nullptr, 0);
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <int KP>
static int launch_tile_bf16_kp(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed) {
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
