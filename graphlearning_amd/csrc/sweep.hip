// Sliced-ELL SpMM over the n x C label matrix: the D^-1 W^T u sweep of Poisson learning
// (reference graphlearning/ssl.py:667-670, :826-827) and the A@p product of utils.conjgrad
// (utils.py:515,523) as one hand-written gfx950 kernel.
//
// Mapping: one wavefront = one slice of R = 64/G rows; the G lanes of a row each own one
// 4-wide column vector of the vertex record (the last used lane owns the fp64 stop value),
// so a neighbour gather is G adjacent lanes reading one contiguous, line-aligned record.
// Entries of a row are accumulated sequentially in stored order with separate multiply
// and add roundings (fp contraction off) -- the order scipy's csr_matvecs uses -- so fp64
// results are bit-identical to the reference's CPU path.  No cross-lane reduction is
// needed for the product itself; wavefront shuffles/DPP serve the entry broadcast, the
// stop-test max and the CG column dots.
#include "glx_internal.h"
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef f32x4 type; };
template <> struct VecOf<double> { typedef f64x4 type; };

struct SpmmParams {
  const int32_t* slot_row;
  const int32_t* slot_len;
  const SliceHdr* slice_hdr;
  const int32_t* col;
  const void* val;
  int64_t head;   // entries of the dense chunk-0 region
  int64_t nslices;
  int64_t nblocks;
  const char* xin;
  char* xout;
  const char* bias;
  const uint8_t* slot_has_bias;
  int rec_bytes;
  int nlanes;  // active lanes per row: nvec + has_w
  int nvec;
  const double* deg;
  const double* vinf;
  const unsigned long long* err_prev;
  unsigned long long* err_next;
  unsigned long long thresh_bits;
  double* dot_partial;
  int dot_ld;
  const double* exit_err;   // CG: skip the launch when !(*exit_err > exit_tol)
  double exit_tol;
  double* prod_out;         // CG reference-order reductions: prod_out[caller_row*dot_ld + c] = xin*xout
  const double* act_row;    // CG column groups: group g still runs iff act_row[g] > exit_tol (null: all)
  int act_cg;               // columns per group
  int act_c;                // total columns
  int prod_sc;              // prod_out is blocked by prod_sc columns: (row, col) at ((col/sc)*n + row)*sc + col%sc
  const int32_t* perm;      // record -> caller row (null: identity)
  int64_t n_rows;
  const int32_t* dup_ptr;   // HAS_DUP (boundary rows of a vertex-partitioned sweep): row r is ALSO stored at records dup_pos[dup_ptr[r] .. dup_ptr[r+1]) of dup_out
  const int32_t* dup_pos;   //   -- the send buffer of the halo exchange, so no pack kernel sits between the SpMM and the transport
  char* dup_out;
  const unsigned* rowmask;  // CG, Dirichlet rows: bit g of rowmask[record] set = A p is held at zero there for system g (null: none)
  CgDev cg;
  // GRP (stacked trials, groups.hip): the columns are `ngroups` groups of `grp_cols`, each group with its own fp64 stop value (group b
  // at byte woff + 8 b of the record, i.e. in the lanes nvec .. nvec + nstop - 1) and its own stop test: err_prev / err_next are rows
  // of [ngroups][GLX_GRP_SHARDS] maxima; groups outside used_mask never run
  int ngroups, grp_cols, nstop;
  unsigned used_mask;
};

// ---- cross-lane helpers -------------------------------------------------------------
template <int T> __device__ __forceinline__ int quad_bcast_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, T * 0x55, 0xf, 0xf, true);
}
template <int T> __device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(quad_bcast_i<T>(__float_as_int(v)));
}
template <int T> __device__ __forceinline__ double quad_bcast(double v) {
  const int lo = quad_bcast_i<T>(__double2loint(v));
  const int hi = quad_bcast_i<T>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_d(double v, int src) {
  const int lo = __shfl(__double2loint(v), src);
  const int hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float shfl_t(float v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double shfl_t(double v, int src) { return shfl_d(v, src); }

// move a value 4 lanes to the right: within a 16-lane row (DPP row_ror:4) or around the wave
__device__ __forceinline__ int row_ror4_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); }
template <typename V> __device__ __forceinline__ V row_ror4(V v) {
  constexpr int NW = sizeof(V) / 4;
  union { V v; int w[NW]; } a, b;
  a.v = v;
#pragma unroll
  for (int i = 0; i < NW; ++i) b.w[i] = row_ror4_i(a.w[i]);
  return b.v;
}
template <int CTRL> __device__ __forceinline__ double row_ror(double v) {      // CTRL = 0x120 + lanes
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <typename V> __device__ __forceinline__ V wave_ror4(V v, int lane) {
  constexpr int NW = sizeof(V) / 4;
  union { V v; int w[NW]; } a, b;
  a.v = v;
  const int src = (lane - 4) & 63;
#pragma unroll
  for (int i = 0; i < NW; ++i) b.w[i] = __shfl(a.w[i], src);
  return b.v;
}

// Wave-wide maxima without LDS round trips (every lane active): two quad permutes and two mirrors leave the maximum of each
// row of 16 lanes in all of its lanes, four v_readlane + scalar maxima finish.  The 64-bit form decides the high words first.
// (The ds_bpermute butterfly this replaces was six dependent LDS round trips -- ~0.3 us in front of every wavefront's first load.)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); v = o > v ? o : v;    // quad_perm [1,0,3,2]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); v = o > v ? o : v;    // quad_perm [2,3,0,1]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); v = o > v ? o : v;   // row_half_mirror
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false); v = o > v ? o : v;   // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)(v & 0xffffffffull);
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

// the maximum of every row of 16 lanes, in all of its lanes (the first four steps of wave_max_u32)
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false); v = o > v ? o : v;
  return v;
}
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)(v & 0xffffffffull);
  const unsigned mh = row16_max_u32(hi);
  const unsigned ml = row16_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int off) {
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xffffffffull), off), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), off);
  return ((unsigned long long)hi << 32) | lo;
}

// acc += v * x for one entry.  Entries past the end of a row carry val = 0 and an unloaded
// x = 0, i.e. a product of exactly 0, and acc (which starts at +0 and therefore can never be
// -0) satisfies acc + 0 == acc bit for bit: no predicate is needed.
template <typename T, bool HAS_W>
__device__ __forceinline__ void accum4(typename VecOf<T>::type& acc, double& accw, T v, const typename VecOf<T>::type& x,
                                       bool is_w) {
#pragma clang fp contract(off)
  typename VecOf<T>::type prod = x * v;
  acc = acc + prod;
  if constexpr (HAS_W && sizeof(T) == 4) {
    // fp32 state: the stop value is an fp64 stored in elements 0..1 of the last vector
    const double xw = __hiloint2double(__float_as_int(x[1]), __float_as_int(x[0]));
    const double pw = (double)v * xw;
    accw = accw + pw;
  }
}

// product of one entry, as it will be added: for the fp32 stop lane the fp64 product travels
// in elements 0..1 of the vector
template <typename T, bool HAS_W>
__device__ __forceinline__ typename VecOf<T>::type product4(T v, const typename VecOf<T>::type& x, bool is_w) {
#pragma clang fp contract(off)
  typename VecOf<T>::type prod = x * v;
  if constexpr (HAS_W && sizeof(T) == 4) {
    if (is_w) {
      const double xw = __hiloint2double(__float_as_int(x[1]), __float_as_int(x[0]));
      const double pw = (double)v * xw;
      prod[0] = __int_as_float(__double2loint(pw));
      prod[1] = __int_as_float(__double2hiint(pw));
    }
  }
  return prod;
}

template <typename T, bool HAS_W>
__device__ __forceinline__ void add_product(typename VecOf<T>::type& acc, double& accw, const typename VecOf<T>::type& pr) {
#pragma clang fp contract(off)
  acc = acc + pr;
  if constexpr (HAS_W && sizeof(T) == 4) {
    const double pw = __hiloint2double(__float_as_int(pr[1]), __float_as_int(pr[0]));
    accw = accw + pw;
  }
}

// One wavefront = one slice of 64/G slots.  A slot is a row (S = 1) or one of the S segments
// a long row is split into (G = 4 only: S = 4 or 16): the segments fetch and multiply their
// entries in parallel and the running sum hops from segment to segment (DPP row rotate /
// wave shuffle), each adding its products in entry order -- long rows stop being a latency
// chain of len/4 dependent memory round trips while the rounding sequence stays that of a
// sequential row sum.
// (Closed experiments -- other loop forms, a two-chunk pipeline, 32-bit offsets, nontemporal loads / stores, a persistent
//  grid -- live as patches under scripts/probes/; EXPERIMENTS.md has their numbers.)
// DOT: 0 none; 1 the column dots p.Ap of the exact CG (cg.hip); 2 the tolerance-mode CG's form (cg_fused.hip)
// GRP (needs HAS_W, G >= 8): column groups with a stop test each -- several training sets of ssl.poisson as ONE sweep (groups.hip)
template <typename T, int G, bool HAS_W, int DOT, bool HAS_DUP = false, bool GRP = false>
__global__ __launch_bounds__(64 * GLX_WPB) void spmm_sell_kernel(const SpmmParams p) {
#pragma clang fp contract(off)
  static_assert(!GRP || (HAS_W && DOT == 0 && !HAS_DUP && G >= 8), "column groups: the stop-column form without extras");
  constexpr bool HAS_DOT = DOT != 0, FUSED = DOT == 2;
  typedef typename VecOf<T>::type V4;
  constexpr int R = 64 / G;
  constexpr int NRED = GLX_WPB * (G == 4 ? 16 : G) * 12 > 256 ? GLX_WPB * (G == 4 ? 16 : G) * 12 : 256;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ double s_red[HAS_DOT ? NRED : 4];

  // the stop values of the previous sweep: the load is issued HERE, the test comes behind the first slice's loads (below), so that
  // the two round trips overlap instead of following each other in front of every wavefront's work
  unsigned long long stop_v = 0;
  unsigned amask = 0;      // GRP: the groups that still run in this sweep
  if constexpr (GRP) {
    // group b runs iff the previous sweep's maximum of its stop value is above 1/n and not NaN (ssl.py:667, per training set): 16
    // shards per group, four groups per load, a row-of-16 maximum and a ballot turn them into a wave-uniform bit mask
    amask = p.used_mask;
    if (p.err_prev) {
      unsigned m = 0;
      for (int i = 0; i * 4 < p.ngroups; ++i) {
        const int grp = i * 4 + (lane >> 4);
        unsigned long long v = grp < p.ngroups ? p.err_prev[grp * GLX_GRP_SHARDS + (lane & 15)] : 0ull;
        v = row16_max_u64(v);
        const unsigned long long bal = __ballot(v > p.thresh_bits && v <= 0x7ff0000000000000ull);
        m |= (unsigned)(((bal & 1ull) | ((bal >> 15) & 2ull) | ((bal >> 30) & 4ull) | ((bal >> 45) & 8ull)) << (i * 4));
      }
      amask &= m;
    }
    if (amask == 0) return;      // every training set has stopped: nothing to do (decided identically by every wavefront)
  } else if constexpr (HAS_W) {
    if (p.err_prev) stop_v = p.err_prev[lane];
  }
  const double* act_row = nullptr;
  if constexpr (HAS_DOT) {
    if constexpr (FUSED) {
      // tolerance-mode CG (cg_fused.hip): the iteration number lives on the device, so that one captured launch sequence serves
      // every iteration; this kernel reads it_a and hands it to the update kernel through it_b
      const int it = *p.cg.it_a;
      if (blockIdx.x == 0 && threadIdx.x == 0) *p.cg.it_b = it;
      if ((int64_t)blockIdx.x == p.nblocks) {   // the extra workgroup: closes iteration it - 1 beside the product
        glx_cg_close_iteration(p.cg, it, p.exit_tol, s_red);
        return;
      }
      if (it > p.cg.max_iter) return;
      // already known to have stopped (a replay behind the last iteration): nothing to do.  A workgroup that does not see the
      // record yet computes a product nobody reads -- the update kernel decides after the launch boundary
      if (it >= 2 && *p.cg.closed >= it - 1 && !(p.cg.err_hist[(size_t)(it - 1) * p.cg.stride + p.cg.ngroups] > p.exit_tol)) return;
      // systems known to have converged neither gather nor store (the record of iteration it - 2: conservative by one iteration)
      act_row = (p.cg.ngroups > 1 && it >= 2) ? p.cg.err_hist + (size_t)(it - 2) * p.cg.stride : nullptr;
    } else {
      if (p.exit_err && !(*p.exit_err > p.exit_tol)) return;
      act_row = p.act_row;
    }
  }
  const int64_t bpx = p.nblocks / 8;
  const int g = lane / G, c = lane % G;
  bool lane_on = c < p.nlanes;
  if constexpr (HAS_DOT) {
    // column groups (CG on several systems): lanes whose 4 columns all belong to converged
    // systems neither gather nor store
    if (act_row && lane_on) {
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = c * 4 + e;
        if (col < p.act_c) any = any || (act_row[col / p.act_cg] > p.exit_tol);
      }
      lane_on = any;
    }
  }
  bool is_w = HAS_W && (c == p.nvec);
  bool ea[4] = {true, true, true, true};     // GRP: which of the lane's four elements belong to a running group
  if constexpr (GRP) {
    is_w = c >= p.nvec && c < p.nlanes;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int grp;
      if (c < p.nvec) {
        const int col = c * 4 + e;
        grp = col < p.ngroups * p.grp_cols ? col / p.grp_cols : 32;
      } else {
        grp = sizeof(T) == 8 ? (c - p.nvec) * 4 + e : (c - p.nvec) * 2 + (e >> 1);
      }
      ea[e] = c < p.nlanes && grp < p.ngroups && ((amask >> grp) & 1u);
    }
    lane_on = ea[0] || ea[1] || ea[2] || ea[3];
  }
  const T* __restrict__ valp = (const T*)p.val;
  const size_t lane_off = (size_t)c * 4 * sizeof(T);

  // XCD-aware block -> slice-group map: the dispatcher places block b on XCD b % 8; each XCD is handed a contiguous range of
  // slices so rows that share neighbours share an L2 (the plan pads every range to nblocks / 8 blocks: a plain transpose)
  const int64_t vb = (int64_t)(blockIdx.x % 8) * bpx + blockIdx.x / 8;
  const int64_t slice = vb * GLX_WPB + wave;
  // header + chunk 0 of the slice (chunk 0 sits at a slice-indexed address: its load is issued together with the header loads)
  int col0 = 0, row = -1, len = 0, nchunks = 0, S = 1, full = 0;
  T val0 = 0;
  int64_t base = 0;
  if (slice < p.nslices) {
    col0 = p.col[slice * 64 + lane];
    val0 = valp[slice * 64 + lane];
    const int64_t slot = slice * R + g;
    row = p.slot_row[slot];
    len = p.slot_len[slot];
    const SliceHdr hd = p.slice_hdr[slice];
    base = p.head + hd.ptr - 64;   // chunk k >= 1 at base + k*64
    nchunks = hd.nchunks;
    S = hd.S & 0xff;
    full = hd.S >> 8;
  }
  // (Round 6 tried asking for what the epilogue needs from memory HERE -- the row's bias flag, degree and stop target -- instead of behind
  // the chunk loop: the phase replays, profiles/r06_sweep_phases.txt, put the epilogue at 1.9 us of the full kernel.  Measured: 12.86 us per
  // launch against 12.4 -- the three early loads per lane compete with the first gathers; not kept.)
  if constexpr (HAS_W && !GRP) {
    if (p.err_prev) {   // stop test of ssl.py:667, decided identically by every wavefront
      // (`while ... np.max(np.absolute(v-vinf)) > 1/n`: a NaN maximum compares False and ends the loop too;
      //  NaN errors are recorded as a bit pattern above +inf, so they dominate the max like numpy's)
      const unsigned long long m = wave_max_u64(stop_v);
      if (m <= p.thresh_bits || m > 0x7ff0000000000000ull) return;
    }
  }
  // tolerance-mode CG: the entries of a long row need not be added in stored order -- its S segments sum their own entries and
  // a fixed tree combines them at the end, instead of the running sum hopping from segment to segment in every chunk
  constexpr bool relaxed = FUSED;
  // chunks [0, full): every lane of every slot has a real entry (plan) and every lane of a row is in use -> no predicates
  full = (p.nlanes == G && !(HAS_DOT && act_row)) ? (full < nchunks ? full : nchunks) : 0;
  const int seg = g & (S - 1);            // S is a power of two
  V4 acc = {0, 0, 0, 0};
  double accw = 0.0;
  double accw1 = 0.0;      // GRP, fp32 state: a stop lane carries two fp64 stop values (elements 0..1 and 2..3)

  if constexpr (G == 4) {
    // Software pipeline: the index / value chunk k+1 travels while the neighbour gathers of chunk k do.
    struct CV { int col; T val; };
    auto issue = [&](const CV& cv, int k, V4 (&x)[4], T (&v)[4]) {
      const int c0 = quad_bcast_i<0>(cv.col), c1 = quad_bcast_i<1>(cv.col), c2 = quad_bcast_i<2>(cv.col), c3 = quad_bcast_i<3>(cv.col);
      v[0] = quad_bcast<0>(cv.val);
      v[1] = quad_bcast<1>(cv.val);
      v[2] = quad_bcast<2>(cv.val);
      v[3] = quad_bcast<3>(cv.val);
      const int j0 = (k * S + seg) * 4;   // first row entry this slot holds in chunk k
      x[0] = V4{0, 0, 0, 0};
      x[1] = V4{0, 0, 0, 0};
      x[2] = V4{0, 0, 0, 0};
      x[3] = V4{0, 0, 0, 0};
      if (lane_on && j0 + 0 < len) x[0] = *(const V4*)(p.xin + (size_t)c0 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 1 < len) x[1] = *(const V4*)(p.xin + (size_t)c1 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 2 < len) x[2] = *(const V4*)(p.xin + (size_t)c2 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 3 < len) x[3] = *(const V4*)(p.xin + (size_t)c3 * p.rec_bytes + lane_off);
    };
    auto consume = [&](int k, const V4 (&x)[4], const T (&v)[4]) {
      if (relaxed || S == 1) {   // relaxed: every segment keeps its own partial sum, added up once behind the loop
        accum4<T, HAS_W>(acc, accw, v[0], x[0], is_w);
        accum4<T, HAS_W>(acc, accw, v[1], x[1], is_w);
        accum4<T, HAS_W>(acc, accw, v[2], x[2], is_w);
        accum4<T, HAS_W>(acc, accw, v[3], x[3], is_w);
      } else {
        // the running sum visits the row's S segments in order: whoever holds it adds its 4
        // products, then it moves 4 lanes on (every lane executes the adds; only the holder's
        // count).  After S hops it is back at segment 0, ready for the next chunk.
        const V4 q0 = product4<T, HAS_W>(v[0], x[0], is_w), q1 = product4<T, HAS_W>(v[1], x[1], is_w);
        const V4 q2 = product4<T, HAS_W>(v[2], x[2], is_w), q3 = product4<T, HAS_W>(v[3], x[3], is_w);
        for (int ph = 0; ph < S; ++ph) {
          add_product<T, HAS_W>(acc, accw, q0);
          add_product<T, HAS_W>(acc, accw, q1);
          add_product<T, HAS_W>(acc, accw, q2);
          add_product<T, HAS_W>(acc, accw, q3);
          if (S == 4) {
            acc = row_ror4(acc);
            if constexpr (HAS_W && sizeof(T) == 4) accw = row_ror4(accw);
          } else {
            acc = wave_ror4(acc, lane);
            if constexpr (HAS_W && sizeof(T) == 4) accw = wave_ror4(accw, lane);
          }
        }
      }
    };
    V4 xA[4];
    T vA[4];
    // The next chunk's index / value load is unconditional (the ADDRESS is selected, not the value: chunk 0 is re-read from its
    // slice-indexed place) and it is issued BEHIND the gathers: the wait in front of the adds is `vmcnt(2)` -- the gathers, not the
    // two youngest loads -- and the next chunk's indices have the whole gather round trip to arrive.  (A conditional load, or one
    // issued in front of the gathers, makes the compiler's wait-counter pass drain everything at the join: two memory round trips
    // per chunk instead of one -- read off the ISA in round 3.)
    auto load_cv = [&](int k) -> CV {
      const int kc = k < nchunks ? k : nchunks - 1;
      const int64_t off = (kc <= 0 ? slice * 64 : base + (int64_t)kc * 64) + lane;
      CV r;
      r.col = p.col[off];
      r.val = valp[off];
      return r;
    };
    // gathers of a full chunk: straight-line, no zero fills, no exec-mask branches (a third of the loop's vector instructions)
    auto issue_full = [&](const CV& cv, V4 (&x)[4], T (&v)[4]) {
      const int c0 = quad_bcast_i<0>(cv.col), c1 = quad_bcast_i<1>(cv.col), c2 = quad_bcast_i<2>(cv.col), c3 = quad_bcast_i<3>(cv.col);
      v[0] = quad_bcast<0>(cv.val);
      v[1] = quad_bcast<1>(cv.val);
      v[2] = quad_bcast<2>(cv.val);
      v[3] = quad_bcast<3>(cv.val);
      x[0] = *(const V4*)(p.xin + (size_t)c0 * p.rec_bytes + lane_off);
      x[1] = *(const V4*)(p.xin + (size_t)c1 * p.rec_bytes + lane_off);
      x[2] = *(const V4*)(p.xin + (size_t)c2 * p.rec_bytes + lane_off);
      x[3] = *(const V4*)(p.xin + (size_t)c3 * p.rec_bytes + lane_off);
    };
    CV qn;
    qn.col = col0;
    qn.val = val0;
    int k = 0;
    if (sizeof(T) == 4)     // fp64: the second loop costs 4 registers and with them a wavefront per SIMD (measured: slower)
      for (; k < full; ++k) {
        const CV qc = qn;
        issue_full(qc, xA, vA);
        qn = load_cv(k + 1);
        consume(k, xA, vA);
      }
    for (; k < nchunks; ++k) {
      const CV qc = qn;
      issue(qc, k, xA, vA);
      qn = load_cv(k + 1);
      consume(k, xA, vA);
    }
  } else {
    // Wide records (G >= 8 lanes per row: many columns, or several training sets stacked -- groups.hip).  A chunk holds G entries of
    // each of the wavefront's 64 / G rows; its index / value pair arrives with ONE coalesced load (the first chunk's together with
    // the slice header above, the next chunk's while the current one is consumed), entry j of a row goes to the row's G lanes by a
    // cross-lane read, and the gathers are issued NB at a time before the first of them is consumed: a row of 16 entries is two
    // memory round trips behind the header's, not one per four entries.  The loop over a chunk's entries ends at the longest row of
    // the wavefront (rows are sorted by length: the rows of a slice are about equally long).
    constexpr int NB = 8;
    const int gbase = lane & ~(G - 1);
    int mlen = len;
#pragma unroll
    for (int off = 32; off >= G; off >>= 1) {
      const int o = __shfl_xor(mlen, off);
      mlen = o > mlen ? o : mlen;
    }
    int colv = col0;
    T valv = val0;
    for (int k = 0; k < nchunks; ++k) {
      const int j0 = k * G;
      // (unconditional: past the last chunk the first one is read again, cf. load_cv of the G = 4 loop)
      const int64_t noff = (k + 1 < nchunks ? base + (int64_t)(k + 1) * 64 : slice * 64) + lane;
      const int coln = p.col[noff];
      const T valn = valp[noff];
      for (int tb = 0; tb < G && j0 + tb < mlen; tb += NB) {
        int cj[NB];
        T vj[NB];
        V4 xj[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          cj[t] = __shfl(colv, gbase + tb + t);
          vj[t] = shfl_t(valv, gbase + tb + t);
          xj[t] = V4{0, 0, 0, 0};
          if (lane_on && (j0 + tb + t < len)) xj[t] = *(const V4*)(p.xin + (size_t)cj[t] * p.rec_bytes + lane_off);
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          accum4<T, HAS_W>(acc, accw, vj[t], xj[t], is_w);
          if constexpr (GRP && sizeof(T) == 4) {
            const double xw = __hiloint2double(__float_as_int(xj[t][3]), __float_as_int(xj[t][2]));
            const double pw = (double)vj[t] * xw;
            accw1 = accw1 + pw;
          }
        }
      }
      colv = coln;
      valv = valn;
    }
  }

  if constexpr (FUSED && G == 4) {
    if (S > 1) {      // the segments' partial sums: two DPP rotations inside the 16-lane row, two exchanges across rows
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double a = (double)acc[e];
        a += row_ror<0x124>(a);
        a += row_ror<0x128>(a);
        if (S == 16) {
          a += shfl_d(a, lane ^ 16);
          a += shfl_d(a, lane ^ 32);
        }
        acc[e] = (T)a;
      }
    }
  }
  // epilogue: u_out[row] = Db[row] + acc   (ssl.py:668: `Db + P*u`; addition commutes bitwise)
  V4 outv = acc;
  const bool store_on = (GRP ? c < p.nlanes : lane_on) && row >= 0 && seg == 0;
  if (store_on) {
    bool hb = p.bias != nullptr;
    if (hb && p.slot_has_bias) hb = p.slot_has_bias[slice * R + g] != 0;
    if (hb) {
      const V4 b = *(const V4*)(p.bias + (size_t)row * p.rec_bytes + lane_off);
      outv = b + acc;
    }
    if constexpr (HAS_W && sizeof(T) == 4) {
      if (is_w) {
        outv[0] = __int_as_float(__double2loint(accw));
        outv[1] = __int_as_float(__double2hiint(accw));
        outv[2] = 0;
        outv[3] = 0;
        if constexpr (GRP) {
          outv[2] = __int_as_float(__double2loint(accw1));
          outv[3] = __int_as_float(__double2hiint(accw1));
        }
      }
    }
    if constexpr (GRP) {
      // a training set that has stopped keeps its iterate: its elements are copied forward from the row's own record, so that
      // both buffers hold u_T of that set from its last sweep on, whatever the other sets still do
      if (!(ea[0] && ea[1] && ea[2] && ea[3])) {
        const V4 own = *(const V4*)(p.xin + (size_t)row * p.rec_bytes + lane_off);
#pragma unroll
        for (int e = 0; e < 4; ++e) outv[e] = ea[e] ? outv[e] : own[e];
      }
    }
    if constexpr (HAS_DOT) {
      // Dirichlet rows (ssl.laplace, ssl.py:1232-1241): A p is held at zero on the labelled rows of each system (bit g of the row's
      // mask), so x, r and p stay zero there and the solve is the one on the sub-matrix (cg.hip)
      if (p.rowmask) {
        const unsigned m = p.rowmask[row];
        if (m) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int col = c * 4 + e;
            if (col < p.act_c && ((m >> (col / p.act_cg)) & 1u)) outv[e] = 0;
          }
        }
      }
    }
    // (stacked state, measured: non-temporal stores of the new iterate -- meant to keep the records being gathered in the L2 -- change
    // nothing, 35.3 vs 35.3 us at 4 trials per record, 76.0 vs 76.1 at 8, three alternating runs on one box: profiles/r05_trials_gd.txt)
    *(V4*)(p.xout + (size_t)row * p.rec_bytes + lane_off) = outv;
    if constexpr (HAS_DUP) {
      // a boundary row leaves for its peers straight from the registers: one more store per destination
      for (int q = p.dup_ptr[row], q1 = p.dup_ptr[row + 1]; q < q1; ++q)
        *(V4*)(p.dup_out + (size_t)p.dup_pos[q] * p.rec_bytes + lane_off) = outv;
    }
  }

  if constexpr (GRP) {
    if (p.err_next) {   // per training set: max_i |v_i - vinf_i| with v = deg * w  (ssl.py:667), the sets that still run only
      unsigned long long ev[4] = {0ull, 0ull, 0ull, 0ull};
      if (store_on && is_w) {
        const double dg = p.deg[row], vi = p.vinf[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (sizeof(T) == 4 && (e & 1)) continue;
          if (!ea[e]) continue;
          double wnew;
          if constexpr (sizeof(T) == 4) wnew = e == 0 ? accw : accw1; else wnew = (double)outv[e];
          double er = fabs(dg * wnew - vi);
          if (er != er) er = __longlong_as_double(0x7ff8000000000000ll);
          ev[e] = (unsigned long long)__double_as_longlong(er);
        }
      }
      // the rows of the wavefront (lanes with the same c), then the wavefronts of the workgroup through LDS, one atomic per group
#pragma unroll
      for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned long long o = shfl_xor_u64(ev[e], off);
          ev[e] = o > ev[e] ? o : ev[e];
        }
      }
      __shared__ unsigned long long s_eg[GLX_WPB][32];
      if (g == 0 && is_w) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (sizeof(T) == 4 && (e & 1)) continue;
          const int grp = sizeof(T) == 8 ? (c - p.nvec) * 4 + e : (c - p.nvec) * 2 + (e >> 1);
          if (grp < p.ngroups) s_eg[wave][grp] = ev[e];
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < p.ngroups) {
        unsigned long long mm = s_eg[0][threadIdx.x];
        for (int w = 1; w < GLX_WPB; ++w) mm = s_eg[w][threadIdx.x] > mm ? s_eg[w][threadIdx.x] : mm;
        if (mm != 0) atomicMax(&p.err_next[threadIdx.x * GLX_GRP_SHARDS + (blockIdx.x & (GLX_GRP_SHARDS - 1))], mm);
      }
    }
  } else if constexpr (HAS_W) {
    if (p.err_next) {   // max_i |v_i - vinf_i| with v = deg * w  (ssl.py:667)
      double e = 0.0;
      if (store_on && is_w) {
        double wnew;
        if constexpr (sizeof(T) == 4) wnew = accw; else wnew = (double)outv[0];
        e = fabs(p.deg[row] * wnew - p.vinf[row]);
        if (e != e) e = __longlong_as_double(0x7ff8000000000000ll);   // canonical NaN: orders above +inf as a bit pattern (np.max propagates NaN)
      }
      const unsigned long long m = wave_max_u64((unsigned long long)__double_as_longlong(e));
      __shared__ unsigned long long s_err[GLX_WPB];
      if (lane == 0) s_err[wave] = m;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long mm = s_err[0];
        for (int w = 1; w < GLX_WPB; ++w) mm = s_err[w] > mm ? s_err[w] : mm;
        if (mm != 0) atomicMax(&p.err_next[blockIdx.x & 63], mm);
      }
    }
  }

  if constexpr (HAS_DOT) {
    // column dots over the rows (utils.py:524 `np.sum(p*Ap,axis=0)`): fixed-order tree inside the block, one partial row per
    // block, reduced by the consumer.  ND = 1: p.Ap;  tolerance-mode CG, ND = 3: p.Ap, r.Ap, Ap.Ap (cg_fused.hip)
    double d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (store_on) {
      const V4 own = *(const V4*)(p.xin + (size_t)row * p.rec_bytes + lane_off);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = (double)own[e] * (double)outv[e];
      if constexpr (FUSED) {
        const V4 rv = *(const V4*)(p.cg.r + (size_t)row * p.rec_bytes + lane_off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d[4 + e] = (double)rv[e] * (double)outv[e];
          d[8 + e] = (double)outv[e] * (double)outv[e];
        }
      }
      if (p.prod_out && c < p.nvec) {   // elementwise p*Ap in the array dtype, row-major (dot_ld columns) in the caller's row order
        const V4 pr = own * outv;
        const int64_t orow = p.perm ? p.perm[row] : row;
        f64x4 pd;
#pragma unroll
        for (int e = 0; e < 4; ++e) pd[e] = (double)pr[e];
        *(f64x4*)(p.prod_out + ((size_t)((c * 4) / p.prod_sc) * p.n_rows + orow) * p.prod_sc + (c * 4) % p.prod_sc) = pd;
      }
    }
    if constexpr (!FUSED) {
#pragma unroll
      for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] += shfl_d(d[e], lane ^ off);
      }
      if (lane < G) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s_red[(wave * G + lane) * 4 + e] = d[e];
      }
      __syncthreads();
      if (threadIdx.x < G && threadIdx.x < p.nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double s = s_red[(0 * G + threadIdx.x) * 4 + e];
          for (int w = 1; w < GLX_WPB; ++w) s += s_red[(w * G + threadIdx.x) * 4 + e];
          p.dot_partial[(size_t)vb * p.dot_ld + threadIdx.x * 4 + e] = s;
        }
      }
    } else {
      // rows of the wavefront: two DPP row rotations add the four slots of every 16-lane row (G = 4), the rest goes through LDS
      constexpr int NSUB = G == 4 ? 4 : 1;      // sub-sums per wavefront handed to LDS
      if constexpr (G == 4) {
#pragma unroll
        for (int e = 0; e < 12; ++e) {
          d[e] += row_ror<0x124>(d[e]);
          d[e] += row_ror<0x128>(d[e]);
        }
        if ((lane & 15) < 4) {
#pragma unroll
          for (int e = 0; e < 12; ++e) s_red[((wave * 4 + (lane >> 4)) * 4 + (lane & 3)) * 12 + e] = d[e];
        }
      } else {
#pragma unroll
        for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
          for (int e = 0; e < 12; ++e) d[e] += shfl_d(d[e], lane ^ off);
        }
        if (lane < G) {
#pragma unroll
          for (int e = 0; e < 12; ++e) s_red[(wave * G + lane) * 12 + e] = d[e];
        }
      }
      __syncthreads();
      // the workgroup's sums leave with agent-scope stores; the last arriver of a group of p.cg.grp workgroups adds the group's
      // rows (fixed order: deterministic) -- the update kernel adds the groups (cg_fused.hip)
      const int ncols = p.dot_ld, nq = 3 * ncols;
      for (int q = threadIdx.x; q < nq; q += 64 * GLX_WPB) {
        const int dot = q / ncols, col = q % ncols, cc = col / 4, e = col % 4;
        double s = 0.0;
        if (cc < p.nvec) {
          for (int w = 0; w < GLX_WPB * NSUB; ++w) s += s_red[(w * G + cc) * 12 + dot * 4 + e];
        }
        glx_agent_store(p.cg.part1 + (size_t)vb * nq + q, s);
      }
      const int64_t grp = vb / p.cg.grp;
      const int64_t g0 = grp * p.cg.grp, g1 = g0 + p.cg.grp < p.nblocks ? g0 + p.cg.grp : p.nblocks;
      if (!glx_arrive_last(p.cg.tick1 + grp, (unsigned)(g1 - g0), s_red)) return;
      glx_reduce_rows<true>(p.cg.part1 + (size_t)g0 * nq, g1 - g0, nq, s_red, [&](int q, double tot) {
        p.cg.part1g[(size_t)grp * nq + q] = tot;
      });
    }
  }
}

int64_t glx_spmm_blocks(const SellPlan* plan) { return (plan->nslices + GLX_WPB - 1) / GLX_WPB; }

template <typename T, int G>
static int launch_g(const SweepArgs& a, const SpmmParams& p, hipStream_t stream) {
  const dim3 grid((unsigned)p.nblocks), block(64 * GLX_WPB);
  if (a.ngroups > 1) {
    if constexpr (G >= 8) hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, 0, false, true>), grid, block, 0, stream, p);
  } else if (a.cg) {      // one workgroup more: it closes the previous iteration beside the product (cg_fused.hip)
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, 2>), dim3((unsigned)p.nblocks + 1), block, 0, stream, p);
  } else if (a.dot_partial) {
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, 1>), grid, block, 0, stream, p);
  } else if (a.has_w && a.dup_ptr) {
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, 0, true>), grid, block, 0, stream, p);
  } else if (a.has_w) {
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, 0>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, 0>), grid, block, 0, stream, p);
  }
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

template <typename T>
static int launch_t(const SweepArgs& a, const SpmmParams& p, hipStream_t stream) {
  switch (a.plan->G) {
    case 4: return launch_g<T, 4>(a, p, stream);
    case 8: return launch_g<T, 8>(a, p, stream);
    case 16: return launch_g<T, 16>(a, p, stream);
    case 32: return launch_g<T, 32>(a, p, stream);
    case 64: return launch_g<T, 64>(a, p, stream);
  }
  glx_set_error("spmm: unsupported lanes-per-row %d", a.plan->G);
  return GLX_EUNSUPPORTED;
}

int glx_launch_spmm(const SweepArgs& a, hipStream_t stream) {
  GLX_CHECK(!((a.dot_partial || a.cg) && a.has_w), GLX_EINVAL, "spmm: dot and stop column are exclusive");
  GLX_CHECK(!a.dup_ptr || (a.has_w && a.dup_pos && a.dup_out), GLX_EINVAL, "spmm: the send-buffer scatter needs the stop column form and all three arrays");
  GLX_CHECK(a.plan->G == a.L.G, GLX_EINVAL, "spmm: plan G=%d but layout G=%d", a.plan->G, a.L.G);
  if (a.plan->nslices == 0) return GLX_OK;
  SpmmParams p;
  memset(&p, 0, sizeof(p));
  p.slot_row = a.plan->d_slot_row;
  p.slot_len = a.plan->d_slot_len;
  p.slice_hdr = a.plan->d_slice_hdr;
  p.col = a.plan->d_col;
  p.val = a.plan->d_val;
  p.head = a.plan->head;
  p.nslices = a.plan->nslices;
  p.nblocks = glx_spmm_blocks(a.plan);
  p.xin = (const char*)a.xin;
  p.xout = (char*)a.xout;
  p.bias = (const char*)a.bias;
  p.slot_has_bias = a.slot_has_bias;
  p.rec_bytes = a.L.ld * a.L.esize;
  p.nvec = a.L.nvec;
  p.nlanes = a.L.nvec + (a.has_w ? 1 : 0);
  if (a.ngroups > 1) {
    GLX_CHECK(a.has_w && !a.dot_partial && !a.cg && !a.dup_ptr && a.L.G >= 8 && a.ngroups <= 32 && a.L.ngroups == a.ngroups, GLX_EINVAL,
              "spmm: column groups need the stop-column form, a grouped layout and at most 32 groups");
    p.nlanes = a.L.nvec + a.L.nstop;
    p.ngroups = a.ngroups;
    p.grp_cols = a.L.C / a.ngroups;
    p.nstop = a.L.nstop;
    p.used_mask = a.used_mask;
  }
  p.deg = a.deg;
  p.vinf = a.vinf;
  p.err_prev = a.err_prev;
  p.err_next = a.err_next;
  union { double d; unsigned long long u; } cv;
  cv.d = a.thresh;
  p.thresh_bits = cv.u;
  p.dot_partial = a.dot_partial;
  p.dot_ld = a.L.nvec * 4;
  p.exit_err = a.exit_err;
  p.exit_tol = a.exit_tol;
  p.prod_out = a.prod_out;
  p.act_row = a.act_row;
  p.act_cg = a.act_cg;
  p.act_c = a.act_c;
  p.prod_sc = a.prod_sc > 0 ? a.prod_sc : 4;
  p.perm = a.perm;
  p.n_rows = a.n_rows;
  p.dup_ptr = a.dup_ptr;
  p.dup_pos = a.dup_pos;
  p.dup_out = (char*)a.dup_out;
  p.rowmask = a.rowmask;
  if (a.cg) p.cg = *a.cg;
  return a.dtype == GLX_F32 ? launch_t<float>(a, p, stream) : launch_t<double>(a, p, stream);
}

// ---- dense (n,C) <-> vertex records ---------------------------------------------------
template <typename T>
__global__ void pack_kernel(const T* __restrict__ dense, T* __restrict__ rec, int64_t n, int C, int ld,
                            const int32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld) return;
  const int64_t row = i / ld;
  const int c = (int)(i % ld);
  T v = 0;
  if (c < C && dense) v = dense[(perm ? (int64_t)perm[row] : row) * C + c];
  rec[i] = v;
}

template <typename T>
__global__ void unpack_kernel(const T* __restrict__ rec, T* __restrict__ dense, int64_t n, int C, int ld,
                              const int32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  const int64_t row = i / C;
  const int c = (int)(i % C);
  dense[(perm ? (int64_t)perm[row] : row) * C + c] = rec[row * ld + c];
}

__global__ void write_w_kernel(char* __restrict__ rec, int64_t n, int rec_bytes, int woff, const double* __restrict__ w,
                               const int32_t* __restrict__ perm) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < n) *(double*)(rec + (size_t)row * rec_bytes + woff) = w[perm ? perm[row] : row];
}

// dense (n,C) [or zeros when dense == nullptr] -> records; then the fp64 stop values, if any
int glx_pack_records(const void* dense, void* rec, int64_t n, const RecLayout& L, int dtype, const double* w, hipStream_t s,
                     const int32_t* perm) {
  const int64_t total = n * L.ld;
  if (total == 0) return GLX_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dense, (float*)rec, n, L.C, L.ld, perm);
  else
    hipLaunchKernelGGL(pack_kernel<double>, dim3(grid), dim3(256), 0, s, (const double*)dense, (double*)rec, n, L.C, L.ld, perm);
  GLX_HIP(hipGetLastError());
  if (L.woff >= 0 && w) {
    hipLaunchKernelGGL(write_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (char*)rec, n, L.ld * L.esize, L.woff, w, perm);
    GLX_HIP(hipGetLastError());
  }
  return GLX_OK;
}

int glx_unpack_records(const void* rec, void* dense, int64_t n, const RecLayout& L, int dtype, hipStream_t s,
                       const int32_t* perm) {
  const int64_t total = n * L.C;
  if (total == 0) return GLX_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(unpack_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)rec, (float*)dense, n, L.C, L.ld, perm);
  else
    hipLaunchKernelGGL(unpack_kernel<double>, dim3(grid), dim3(256), 0, s, (const double*)rec, (double*)dense, n, L.C, L.ld, perm);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}
