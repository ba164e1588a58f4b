// The reference-order column reductions of utils.conjgrad (graphlearning/utils.py:524,527: `np.sum(p * Ap, axis=0)`,
// `np.sum(r ** 2, axis=0)`), bit for bit, without walking the n dependent additions one by one: seqsum_exact.h has the
// arithmetic (a running sum inside one binade advances by integer steps; blocks of rows are summed as integers from a guessed
// exponent and accepted only when the exact state confirms the guess).  Three launches per reduction:
//   ss_sum_kernel    plain sums of runs of 16 rows and of groups of 4 blocks (a block = 256 rows)      (the approximate prefix)
//   ss_quant_kernel  per run of 16 rows: the guess from that prefix and the integer record; the 16 records of a block merged in a
//                    tree inside one wavefront
//   ss_walk_kernel   per column one wavefront walks the chunks with the exact state (three more fetch records ahead into LDS): 64
//                    blocks are checked at once, the first one that is not a plain same-binade block is taken through its record
//                    (splits = single fp64 additions) or, if the record does not fit the exact state, row by row; then
//                    alpha / beta / rsold as cg.hip's first reducer.
// The product array is the one cg.hip's producers write: [column block of 4][row in the caller's order][4].
#define GLX_HD __host__ __device__ static inline
#include "cg_internal.h"
#include "seqsum_exact.h"
#include <mutex>

#define SS_MAX_CHUNKS 1024       // flags of one column in LDS (8 KB): n <= 1024 * 64 * SS_BLOCK = 16.7 M rows (cg.hip falls back to the chain above)
static const int SS_PF = 3;     // blocks the walk may have to add row by row whose rows are fetched one chunk ahead

struct SsSoA {                  // [field][column][chunk][64 blocks]
  int32_t* E[SS_MAXSPLIT + 1];
  int32_t* nsplit;
  int64_t *R[SS_MAXSPLIT + 1], *lo[SS_MAXSPLIT + 1], *hi[SS_MAXSPLIT + 1];
  double* xs[SS_MAXSPLIT];
};

size_t glx_seqsum_rec_bytes(int ncols, int nchunks) {
  const size_t N = (size_t)ncols * nchunks * 64;
  return N * (4 * (SS_MAXSPLIT + 2) + 8 * (3 * (SS_MAXSPLIT + 1) + SS_MAXSPLIT));
}

static SsSoA ss_carve(char* base, int ncols, int nchunks) {
  const size_t N = (size_t)ncols * nchunks * 64;
  SsSoA s;
  char* p = base;
  for (int j = 0; j <= SS_MAXSPLIT; ++j) { s.R[j] = (int64_t*)p; p += N * 8; }
  for (int j = 0; j <= SS_MAXSPLIT; ++j) { s.lo[j] = (int64_t*)p; p += N * 8; }
  for (int j = 0; j <= SS_MAXSPLIT; ++j) { s.hi[j] = (int64_t*)p; p += N * 8; }
  for (int j = 0; j < SS_MAXSPLIT; ++j) { s.xs[j] = (double*)p; p += N * 8; }
  for (int j = 0; j <= SS_MAXSPLIT; ++j) { s.E[j] = (int32_t*)p; p += N * 4; }
  s.nsplit = (int32_t*)p;
  return s;
}

// does this launch have work for column block cb?  (the gates of cg.hip's first reducer)
template <int MODE>
__device__ __forceinline__ bool ss_block_live(const CgScalars& sc, int it, double tol, int cb) {
  if (MODE == 2) return true;
  if (!cg_any_active(sc, it, tol)) return false;
  bool any = false;
  for (int c = cb * 4; c < cb * 4 + 4; ++c) any = any || cg_col_active(sc, it, tol, c);
  return any;
}

// Geometry of the two passes in front of the walk: a workgroup of 256 threads = a GROUP of 4 blocks of one column block, one
// wavefront per SIMD (the passes are instruction-bound: four wavefronts on one SIMD take turns); wavefront w = block 4 g + w;
// lane = (run of SS_SUB rows `subw` of that block, column cc) = subw * 4 + cc.  16 groups make a chunk of the walk.
static_assert(SS_SUB == 16 && SS_Q == 16, "the kernels below lay one block out over one wavefront: 16 runs of 16 rows x 4 columns");
#define SS_GB 4                  // blocks (wavefronts) per group

__device__ __forceinline__ double ss_col_sum(double v) {        // sum over the 16 runs of a wavefront, per column (lanes 4 apart)
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <int MODE>
__global__ __launch_bounds__(64 * SS_GB) void ss_sum_kernel(const double* __restrict__ prod, int64_t n, int ngroups, CgScalars sc, int it,
                                                            double tol, double* __restrict__ ssum, double* __restrict__ gsum) {
  const int cb = blockIdx.y, g = blockIdx.x;
  if (!ss_block_live<MODE>(sc, it, tol, cb)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, subw = lane >> 2, cc = lane & 3;
  const double* __restrict__ src = prod + (size_t)cb * n * 4 + cc;
  const int64_t r0 = (((int64_t)g * SS_GB + wave) * SS_Q + subw) * SS_SUB;
  double xv[SS_SUB];                       // every load issued before the first is used
#pragma unroll
  for (int i = 0; i < SS_SUB; ++i) xv[i] = r0 + i < n ? src[(r0 + i) * 4] : 0.0;
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < SS_SUB; ++i) t += xv[i];
  ssum[(((size_t)cb * ngroups + g) * (SS_GB * 16) + wave * 16 + subw) * 4 + cc] = t;
  const double w = ss_col_sum(t);
  __shared__ double sh[SS_GB * 4];
  if (lane < 4) sh[wave * 4 + lane] = w;
  __syncthreads();
  if (threadIdx.x < 4) {
    double c = 0.0;
#pragma unroll
    for (int q = 0; q < SS_GB; ++q) c += sh[q * 4 + threadIdx.x];
    gsum[((size_t)cb * ngroups + g) * 4 + threadIdx.x] = c;
  }
}

__device__ __forceinline__ SsRec ss_shfl_down_rec(const SsRec& r, int delta) {
  SsRec o;
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    o.E[j] = __shfl_down(r.E[j], delta);
    o.R[j] = __shfl_down((long long)r.R[j], delta);
    o.lo[j] = __shfl_down((long long)r.lo[j], delta);
    o.hi[j] = __shfl_down((long long)r.hi[j], delta);
  }
#pragma unroll
  for (int j = 0; j < SS_MAXSPLIT; ++j) o.xs[j] = __shfl_down(r.xs[j], delta);
  o.nsplit = __shfl_down(r.nsplit, delta);
  return o;
}

// badmask: one byte per group (bit w: block 4 g + w goes row by row); the walk packs the 16 bytes of a chunk into its 64-bit word
template <int MODE>
__global__ __launch_bounds__(64 * SS_GB) void ss_quant_kernel(const double* __restrict__ prod, int64_t n, int nchunks, CgScalars sc, int it,
                                                              double tol, const double* __restrict__ ssum, const double* __restrict__ gsum,
                                                              const double* __restrict__ gpre, SsSoA soa, unsigned char* __restrict__ badmask) {
#pragma clang fp contract(off)
  const int cb = blockIdx.y, g = blockIdx.x, ngroups = nchunks * (64 / SS_GB);
  if (!ss_block_live<MODE>(sc, it, tol, cb)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, subw = lane >> 2, cc = lane & 3;
  __shared__ double sh[SS_GB * 4], sh2[SS_GB * 4];
  __shared__ unsigned shm[4];
  const int64_t r0 = (((int64_t)g * SS_GB + wave) * SS_Q + subw) * SS_SUB;
  double xv[SS_SUB];                       // this thread's rows: every load issued here, consumed from registers below
  {
    const double* __restrict__ src = prod + (size_t)cb * n * 4 + cc;
#pragma unroll
    for (int i = 0; i < SS_SUB; ++i) xv[i] = r0 + i < n ? src[(r0 + i) * 4] : 0.0;
  }
  // the approximate state in front of this group (sums of the groups before it) ...
  // (every workgroup adding the sums of ALL groups in front of it is quadratic in the number of groups -- fine for the 69 groups of
  // 70 000 rows, 1.3e8 loads per column block at 16.7 M rows: from SS_SCAN_FROM groups on ss_scan_kernel has left the exclusive prefix)
  double part = 0.0;
  if (gpre) {
    part = (wave == 0 && subw == 0) ? gpre[((size_t)cb * ngroups + g) * 4 + cc] : 0.0;
  } else {
    for (int g2 = wave * 16 + subw; g2 < g; g2 += SS_GB * 16) part += gsum[((size_t)cb * ngroups + g2) * 4 + cc];
  }
  part = ss_col_sum(part);
  // ... and in front of this thread's rows inside it (inclusive scan over the 16 runs of the wavefront, wavefront totals in LDS)
  const double mine = ssum[(((size_t)cb * ngroups + g) * (SS_GB * 16) + wave * 16 + subw) * 4 + cc];
  double incl = mine;
#pragma unroll
  for (int d = 1; d < 16; d *= 2) {
    const double v = __shfl_up(incl, 4 * d);
    if (subw >= d) incl += v;
  }
  if (lane < 4) sh[wave * 4 + lane] = part;
  if (lane >= 60) sh2[wave * 4 + cc] = incl;          // run 15: the wavefront's total
  if (threadIdx.x < 4) shm[threadIdx.x] = 0u;
  __syncthreads();
  double pre = incl - mine;
#pragma unroll
  for (int q = 0; q < SS_GB; ++q) pre += sh[q * 4 + cc] + (q < wave ? sh2[q * 4 + cc] : 0.0);
  const int64_t left = n - r0;
  const int len = left <= 0 ? 0 : (left < SS_SUB ? (int)left : SS_SUB);
  SsRec rec;
  ss_sub_record(xv, len, pre, &rec);
  // the 16 records of the block, merged in a tree: run q takes in run q + d
#pragma unroll
  for (int d = 1; d < 16; d *= 2) {
    const SsRec other = ss_shfl_down_rec(rec, 4 * d);
    if ((subw & (2 * d - 1)) == 0) ss_merge_record(&rec, &other);
  }
  // lanes 0..3 hold the block's record for the four columns
  if (lane < 4 && rec.E[0] == SS_E_BAD) atomicOr(&shm[lane], 1u << wave);
  __syncthreads();
  if (lane < 4) {
    const int col = cb * 4 + lane;
    const size_t o = ((size_t)col * nchunks + g / (64 / SS_GB)) * 64 + (g % (64 / SS_GB)) * SS_GB + wave;
#pragma unroll
    for (int j = 0; j <= SS_MAXSPLIT; ++j) {
      soa.E[j][o] = rec.E[j];
      soa.R[j][o] = rec.R[j];
      soa.lo[j][o] = rec.lo[j];
      soa.hi[j][o] = rec.hi[j];
    }
#pragma unroll
    for (int j = 0; j < SS_MAXSPLIT; ++j) soa.xs[j][o] = rec.xs[j];
    soa.nsplit[o] = rec.nsplit;
    if (wave == 0) badmask[(size_t)col * ngroups + g] = (unsigned char)shm[lane];
  }
}

// exclusive prefix of the group sums of one column block (one workgroup of 256 threads per column block: 64 threads per column walk
// strides of the groups, a wavefront scan carries the running total) -- only launched for many groups (SS_SCAN_FROM)
#define SS_SCAN_FROM 1024
template <int MODE>
__global__ __launch_bounds__(256) void ss_scan_kernel(int ngroups, CgScalars sc, int it, double tol, const double* __restrict__ gsum,
                                                      double* __restrict__ gpre) {
  const int cb = blockIdx.x;
  if (!ss_block_live<MODE>(sc, it, tol, cb)) return;
  const int cc = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double carry = 0.0;
  for (int g0 = 0; g0 < ngroups; g0 += 64) {
    const int g = g0 + lane;
    const double v = g < ngroups ? gsum[((size_t)cb * ngroups + g) * 4 + cc] : 0.0;
    double incl = v;
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
      const double t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (g < ngroups) gpre[((size_t)cb * ngroups + g) * 4 + cc] = carry + (incl - v);
    carry += __shfl(incl, 63);
  }
}

struct SsLane {                 // one block's record, one lane
  int32_t E[SS_MAXSPLIT + 1], nsplit;
  int64_t R[SS_MAXSPLIT + 1], lo[SS_MAXSPLIT + 1], hi[SS_MAXSPLIT + 1];
  double xs[SS_MAXSPLIT];
  unsigned long long excl;      // sum of R[0] over the plain blocks in front of this one in its chunk (wrapping arithmetic): the walk's own scan
};

__device__ __forceinline__ SsLane ss_load_lane(const SsSoA& soa, size_t o) {
  SsLane L;
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    L.E[j] = soa.E[j][o];
    L.R[j] = soa.R[j][o];
    L.lo[j] = soa.lo[j][o];
    L.hi[j] = soa.hi[j][o];
  }
#pragma unroll
  for (int j = 0; j < SS_MAXSPLIT; ++j) L.xs[j] = soa.xs[j][o];
  L.nsplit = soa.nsplit[o];
  L.excl = 0ull;
  return L;
}

__device__ __forceinline__ int ss_rl32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ long long ss_rl64(long long v, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), lane);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ double ss_rlf(double v, int lane) { return __longlong_as_double(ss_rl64(__double_as_longlong(v), lane)); }
// a value every lane holds, moved to scalar registers: what follows from it is scalar arithmetic (one wavefront alone on its
// SIMD pays 4-16 cycles per dependent vector instruction; the walk below is a long dependent chain of integer steps)
__device__ __forceinline__ long long ss_uni(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ double ss_unif(double v) { return __longlong_as_double(ss_uni(__double_as_longlong(v))); }

#define SS_RPL (SS_BLOCK / 64)    // rows of one block per lane: row 64 q + lane in register q
struct SsRows { double v[SS_RPL]; };
// s += the rows of a block, row after row (rows past n were loaded as +0): the chain of cg.hip's first reducer.  Register q of lane l
// holds row 64 q + l; `v_fmac_f64_dpp tot, x, 1.0 row_newbcast:k` adds the value lane k of a DPP row of 16 lanes holds to every lane of
// that row (x * 1 + tot, fused = the correctly rounded sum), so the 16 rows 64 q + 16 r .. + 15 are first replicated into all four
// DPP rows (two ds_bpermute per run of 16, independent of the chain and issued ahead of it) and then added by 16 dependent
// instructions with no cross-lane move in between: 2.4 ns per row like the chain kernel, instead of the 6.7 of a readlane pair + add
// per row (rounds 5: 1.7 us per block).  Every lane ends with the same sum.
#define SS_DPP_L(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ double ss_add_rows(double s, const SsRows& x) {
  const int lane = (int)(threadIdx.x & 63);
  double rep[SS_RPL * 4];
#pragma unroll
  for (int q = 0; q < SS_RPL; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) rep[q * 4 + r] = __shfl(x.v[q], 16 * r + (lane & 15));
  }
  double tot = s;
  const double one = 1.0;
#pragma unroll
  for (int g = 0; g < SS_RPL * 4; ++g) {
    // (s_nop 1: a VALU write of the DPP source right in front of the block needs two wait states before a DPP read; the assembler block
    // is opaque to the hazard recogniser)
    asm volatile("s_nop 1\n\t" SS_DPP_L(0) SS_DPP_L(1) SS_DPP_L(2) SS_DPP_L(3) SS_DPP_L(4) SS_DPP_L(5) SS_DPP_L(6) SS_DPP_L(7) SS_DPP_L(8)
                     SS_DPP_L(9) SS_DPP_L(10) SS_DPP_L(11) SS_DPP_L(12) SS_DPP_L(13) SS_DPP_L(14) SS_DPP_L(15)
                 : "+v"(tot)
                 : "v"(rep[g]), "v"(one));
  }
  return ss_unif(tot);
}

// ss_apply_record (seqsum_exact.h) on the record lane f holds, its fields fetched as they are needed
__device__ __forceinline__ bool ss_apply_lane(double* s, const SsLane& cur, int f) {
#pragma clang fp contract(off)
  const int E0 = ss_rl32(cur.E[0], f);
  if (E0 == SS_E_ANY) return true;
  if (E0 == SS_E_BAD || !ss_valid(*s)) return false;
  int E = ss_expo(*s);
  int64_t K = ss_mant(*s);
  const int nsplit = ss_rl32(cur.nsplit, f);
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    if (j > nsplit) break;
    if (ss_rl32(cur.E[j], f) != E || !ss_range_ok(K, ss_rl64(cur.lo[j], f), ss_rl64(cur.hi[j], f))) return false;
    K += ss_rl64(cur.R[j], f);
    if (j < SS_MAXSPLIT && j < nsplit) {
      const double sn = ss_unif(ss_compose(E, K) + ss_rlf(cur.xs[j < SS_MAXSPLIT ? j : 0], f));
      if (!ss_valid(sn)) return false;
      E = ss_expo(sn);
      K = ss_mant(sn);
    }
  }
  *s = ss_compose(E, K);
  return true;
}

// LDS of the walk: two halves of four chunk slots of records ([field][64 lanes]) and the column's row-by-row flags
#define SS_RING64 (3 * (SS_MAXSPLIT + 1) + SS_MAXSPLIT)           // R, lo, hi per segment, xs per split
#define SS_RING32 (SS_MAXSPLIT + 2)                                // E per segment, nsplit
static size_t ss_walk_lds_bytes(int nchunks) {
  return (size_t)2 * 4 * 64 * (SS_RING64 * 8 + SS_RING32 * 4) + (size_t)nchunks * 8;
}

__device__ __forceinline__ void ss_ring_store(unsigned long long* r64, int* r32, int lane, const SsLane& L) {
  int k = 0;
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    r64[(k++) * 64 + lane] = (unsigned long long)L.R[j];
    r64[(k++) * 64 + lane] = (unsigned long long)L.lo[j];
    r64[(k++) * 64 + lane] = (unsigned long long)L.hi[j];
    r32[j * 64 + lane] = L.E[j];
  }
#pragma unroll
  for (int j = 0; j < SS_MAXSPLIT; ++j) r64[(k++) * 64 + lane] = (unsigned long long)__double_as_longlong(L.xs[j]);
  r32[(SS_MAXSPLIT + 1) * 64 + lane] = L.nsplit;
}
__device__ __forceinline__ SsLane ss_ring_fetch(const unsigned long long* r64, const int* r32, int lane) {
  SsLane L;
  int k = 0;
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    L.R[j] = (int64_t)r64[(k++) * 64 + lane];
    L.lo[j] = (int64_t)r64[(k++) * 64 + lane];
    L.hi[j] = (int64_t)r64[(k++) * 64 + lane];
    L.E[j] = r32[j * 64 + lane];
  }
#pragma unroll
  for (int j = 0; j < SS_MAXSPLIT; ++j) L.xs[j] = __longlong_as_double((long long)r64[(k++) * 64 + lane]);
  L.excl = 0ull;
  L.nsplit = r32[(SS_MAXSPLIT + 1) * 64 + lane];
  return L;
}

// One workgroup of four wavefronts per column.  All four fetch block records (wavefront w the chunks 4 j + w) one PHASE of four
// chunks ahead into the other half of the LDS ring -- a lone wavefront that fetched its own next chunk waited a full memory
// round trip per chunk (1.3 us, measured) --; wavefront 0 walks the four chunks of the current phase from LDS with the exact state.
template <int MODE>
__global__ __launch_bounds__(256) void ss_walk_kernel(const double* __restrict__ prod, int64_t n, int nchunks, int ncols_all, int C,
                                                      CgScalars sc, int it, double tol, SsSoA soa,
                                                      const unsigned long long* __restrict__ badmask, unsigned long long* __restrict__ stats) {
#pragma clang fp contract(off)
  extern __shared__ unsigned long long ss_lds[];
  if (MODE != 2 && !cg_any_active(sc, it, tol)) return;
  const int col = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (col >= C) {                          // a padding column of the last column block: its products are zeros
    if (col < ncols_all && threadIdx.x == 0) {
      if (MODE == 0) sc.alpha[col] = 0.0;
      if (MODE == 1) sc.beta[col] = 0.0;
      if (MODE != 0) sc.rsold[col] = 0.0;
    }
    return;
  }
  const bool live = MODE == 2 || cg_col_active(sc, it, tol, col);
  if (!live) return;
  unsigned long long* ring64 = ss_lds;                                            // [2][4][SS_RING64][64]
  int* ring32 = (int*)(ss_lds + (size_t)2 * 4 * SS_RING64 * 64);                  // [2][4][SS_RING32][64]
  unsigned long long* shmask = (unsigned long long*)(ring32 + (size_t)2 * 4 * SS_RING32 * 64);   // [nchunks]
  auto r64_of = [&](int half, int slot) { return ring64 + (size_t)(half * 4 + slot) * SS_RING64 * 64; };
  auto r32_of = [&](int half, int slot) { return ring32 + (size_t)(half * 4 + slot) * SS_RING32 * 64; };
  const double* __restrict__ src = prod + (size_t)(col >> 2) * n * 4 + (col & 3);
  auto load_rows = [&](int64_t blk) -> SsRows {
    SsRows x;
#pragma unroll
    for (int q = 0; q < SS_RPL; ++q) {
      const int64_t row = blk * SS_BLOCK + q * 64 + lane;
      x.v[q] = row < n ? src[row * 4] : 0.0;
    }
    return x;
  };
  const size_t obase = (size_t)col * nchunks * 64;
  for (int c = threadIdx.x; c < nchunks; c += 256) {       // a byte per group of 4 blocks -> the chunk's 64 flags
    const unsigned char* bm = (const unsigned char*)badmask + ((size_t)col * nchunks + c) * 16;
    unsigned long long m = 0ull;
#pragma unroll
    for (int q = 0; q < 16; ++q) m |= (unsigned long long)(bm[q] & 0xfu) << (4 * q);
    shmask[c] = m;
  }
  if (wave < nchunks) ss_ring_store(r64_of(0, wave), r32_of(0, wave), lane, ss_load_lane(soa, obase + (size_t)wave * 64 + lane));
  __syncthreads();
  // rows of the blocks of a chunk flagged "row by row", in flagged order, for the first SS_PF of them
  auto prefetch = [&](int chunk, SsRows* pf) {
    unsigned long long m = chunk < nchunks ? (unsigned long long)ss_uni((long long)shmask[chunk]) : 0ull;
#pragma unroll
    for (int q = 0; q < SS_PF; ++q) {
#pragma unroll
      for (int k = 0; k < SS_RPL; ++k) pf[q].v[k] = 0.0;
    }
    if (m) {                               // (most chunks have none)
#pragma unroll
      for (int q = 0; q < SS_PF; ++q) {
        if (m) {
          const int f = __ffsll((long long)m) - 1;
          m &= m - 1;
          pf[q] = load_rows((int64_t)chunk * 64 + f);
        }
      }
    }
  };
  double s = 0.0;                          // the exact state: the same in every lane, kept in scalar registers
  int n_plain = 0, n_rec = 0, n_rows = 0;  // blocks taken as plain integers / through their record / row by row
  SsRows pf_nxt[SS_PF];
  if (wave == 0) prefetch(0, pf_nxt);
  const int nphases = (nchunks + 3) / 4;
  for (int phase = 0; phase < nphases; ++phase) {
    const int half = phase & 1;
    const int mine = (phase + 1) * 4 + wave;             // the chunk this wavefront fetches for the next phase
    SsLane Lnext;
    if (mine < nchunks) Lnext = ss_load_lane(soa, obase + (size_t)mine * 64 + lane);
    if (wave == 0) {
      for (int slot = 0; slot < 4; ++slot) {
        const int chunk = phase * 4 + slot;
        if (chunk >= nchunks) break;
        SsLane cur = ss_ring_fetch(r64_of(half, slot), r32_of(half, slot), lane);
        SsRows pf[SS_PF];
#pragma unroll
        for (int q = 0; q < SS_PF; ++q) pf[q] = pf_nxt[q];
        prefetch(chunk + 1, pf_nxt);
        const unsigned long long bad = (unsigned long long)ss_uni((long long)shmask[chunk]);
        const bool plain = cur.nsplit == 0 && cur.E[0] >= 0;
        // the chunk-local exclusive prefix of the plain blocks' totals (wrapping arithmetic)
        unsigned long long total;
        {
          const unsigned long long contrib = plain ? (unsigned long long)cur.R[0] : 0ull;
          unsigned long long v = contrib;
#pragma unroll
          for (int d = 1; d < 64; d *= 2) {
            const unsigned long long t = (unsigned long long)__shfl_up((long long)v, d);
            if (lane >= d) v += t;
          }
          cur.excl = v - contrib;
          total = (unsigned long long)ss_rl64((long long)v, 63);
        }
        const bool any = cur.E[0] == SS_E_ANY;
        int start = 0;
        while (start < 64) {
          const bool valid = ss_valid(s);
          const int E = ss_expo(s);
          const int64_t K = ss_mant(s);
          const unsigned long long ex0 = (unsigned long long)ss_rl64((long long)cur.excl, start);
          const int64_t Kl = (int64_t)((unsigned long long)K + (cur.excl - ex0));      // the state in front of this lane's block
          const bool ok = lane < start || any || (plain && valid && cur.E[0] == E && ss_range_ok(Kl, cur.lo[0], cur.hi[0]));
          const unsigned long long fm = __ballot(!ok);
          const int f = fm ? __ffsll((long long)fm) - 1 : 64;
          if (f > start) {         // blocks start .. f-1 are plain (or empty): one integer addition
            const unsigned long long upto = f < 64 ? (unsigned long long)ss_rl64((long long)cur.excl, f) : total;
            const unsigned long long d = upto - ex0;
            if (valid && d) s = ss_compose(E, (int64_t)((unsigned long long)K + d));
            n_plain += f - start;
          }
          if (f == 64) break;
          // block f: through its record, else row by row
          if (ss_apply_lane(&s, cur, f)) {
            ++n_rec;
          } else {
            SsRows x;
            const bool flagged = ss_rl32(cur.E[0], f) == SS_E_BAD;
            const int slot_pf = flagged ? __popcll(bad & ((1ull << f) - 1ull)) : SS_PF;   // its place among the flagged ones
            if (slot_pf < SS_PF) {
              x = pf[0];
#pragma unroll
              for (int q = 1; q < SS_PF; ++q) {
#pragma unroll
                for (int k = 0; k < SS_RPL; ++k) x.v[k] = slot_pf == q ? pf[q].v[k] : x.v[k];
              }
            } else {
              x = load_rows((int64_t)chunk * 64 + f);
            }
            s = ss_add_rows(s, x);
            ++n_rows;
          }
          start = f + 1;
        }
      }
    }
    if (mine < nchunks) ss_ring_store(r64_of(half ^ 1, wave), r32_of(half ^ 1, wave), lane, Lnext);
    __syncthreads();
  }
  const double tot = s;
  if (threadIdx.x == 0) {
    if (stats) {
      atomicAdd(&stats[0], (unsigned long long)n_plain);
      atomicAdd(&stats[1], (unsigned long long)n_rec);
      atomicAdd(&stats[2], (unsigned long long)n_rows);
    }
    const int gc = col;
    if (MODE == 0) {
      sc.alpha[gc] = gc < C ? sc.rsold[gc] / tot : 0.0;
    } else if (MODE == 1) {
      sc.beta[gc] = gc < C ? tot / sc.rsold[gc] : 0.0;
      sc.rsold[gc] = tot;
    } else {
      sc.rsold[gc] = tot;
    }
  }
}

template <int MODE>
static int ss_launch(const double* prod, int64_t n, int ncols_all, int C, const CgScalars& sc, int it, double tol, const SsWork& w,
                     hipStream_t st) {
  const SsSoA soa = ss_carve(w.rec, ncols_all, w.nchunks);
  const int ngroups = w.nchunks * (64 / SS_GB);
  const dim3 grid((unsigned)ngroups, (unsigned)(ncols_all / 4));             // groups of 4 blocks x column blocks
  hipLaunchKernelGGL(ss_sum_kernel<MODE>, grid, dim3(64 * SS_GB), 0, st, prod, n, ngroups, sc, it, tol, w.bsum, w.csum);
  GLX_HIP(hipGetLastError());
  const double* gpre = nullptr;
  if (ngroups >= SS_SCAN_FROM) {            // the prefix lives behind the group sums in the same buffer (glx_seqsum_sum_doubles)
    double* pre = w.csum + (size_t)(ncols_all / 4) * ngroups * 4;
    hipLaunchKernelGGL(ss_scan_kernel<MODE>, dim3((unsigned)(ncols_all / 4)), dim3(256), 0, st, ngroups, sc, it, tol, (const double*)w.csum, pre);
    GLX_HIP(hipGetLastError());
    gpre = pre;
  }
  hipLaunchKernelGGL(ss_quant_kernel<MODE>, grid, dim3(64 * SS_GB), 0, st, prod, n, w.nchunks, sc, it, tol, (const double*)w.bsum,
                     (const double*)w.csum, gpre, soa, (unsigned char*)w.mask);
  GLX_HIP(hipGetLastError());
  const size_t lds = ss_walk_lds_bytes(w.nchunks);
  hipLaunchKernelGGL(ss_walk_kernel<MODE>, dim3((unsigned)ncols_all), dim3(256), lds, st, prod, n, w.nchunks, ncols_all, C, sc, it, tol, soa,
                     (const unsigned long long*)w.mask, w.stats ? w.stats + 4 * MODE : nullptr);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

int glx_seqsum_run(int mode, const double* prod, int64_t n, int ncols_all, int C, const CgScalars& sc, int it, double tol, const SsWork& w,
                   hipStream_t st) {
  return mode == 0 ? ss_launch<0>(prod, n, ncols_all, C, sc, it, tol, w, st)
       : mode == 1 ? ss_launch<1>(prod, n, ncols_all, C, sc, it, tol, w, st)
                   : ss_launch<2>(prod, n, ncols_all, C, sc, it, tol, w, st);
}

// The walk's dynamic LDS bound: the same whatever the solve (concurrent solves on other operators set it too, and none may lower it under
// a launch in flight), set once per process and device -- before any launch sequence is captured (cg.hip), not inside one.
int glx_seqsum_prepare() {
  static std::mutex mu;
  static std::vector<int> done;
  int dev = 0;
  GLX_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (std::find(done.begin(), done.end(), dev) != done.end()) return GLX_OK;
  const int lds = (int)ss_walk_lds_bytes(SS_MAX_CHUNKS);
  GLX_HIP(hipFuncSetAttribute((const void*)ss_walk_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  GLX_HIP(hipFuncSetAttribute((const void*)ss_walk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  GLX_HIP(hipFuncSetAttribute((const void*)ss_walk_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  done.push_back(dev);
  return GLX_OK;
}

int glx_seqsum_max_chunks() { return SS_MAX_CHUNKS; }
size_t glx_seqsum_sum_doubles(int ncols, int nchunks, int which) {      // which 0: sums of the runs of 16 rows, 1: of the groups
  return which == 0 ? (size_t)(ncols / 4) * nchunks * 64 * SS_Q * 4 : (size_t)2 * (ncols / 4) * nchunks * (64 / SS_GB) * 4;   // (1: + their exclusive prefix)
}
int glx_seqsum_chunks(int64_t n) { return (int)((n + (int64_t)SS_BLOCK * 64 - 1) / ((int64_t)SS_BLOCK * 64)); }
