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
#ifndef GLX_PIPE_DEPTH
#define GLX_PIPE_DEPTH 1
#endif
#ifndef GLX_NT_STREAM
#define GLX_NT_STREAM 0   // measured: nontemporal operator loads 15.5 vs 13.4 us (the image is re-read from the Infinity Cache every sweep)
#endif
#ifndef GLX_NT_STORE
#define GLX_NT_STORE 0
#endif
#ifndef GLX_LOOP_FORM
#define GLX_LOOP_FORM 1   // 1: branch-free chunk prefetch issued BEHIND the chunk's gathers (see the chunk loop); 0: the round-1 order
#endif
#ifndef GLX_FULL_CHUNKS
#define GLX_FULL_CHUNKS 1   // chunks in which every slot of the slice has four real entries are gathered without predicates
#endif
#ifndef GLX_OFF32
#define GLX_OFF32 0   // experiment: 32-bit record offsets from a uniform base (state < 4 GB) instead of 64-bit address arithmetic
#endif
#ifndef GLX_PERSIST_DEFAULT
#define GLX_PERSIST_DEFAULT 1   // blocks per workgroup of the sweep kernel (see the persistent form in spmm_sell_kernel)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef f32x4 type; };
template <> struct VecOf<double> { typedef f64x4 type; };

// 256 bytes of zeros: what the lanes of an entry past the end of its row gather in the branch-free loop forms (0 * 0 = 0 exactly,
// whatever the state holds -- a product with a real record could be 0 * inf)
__device__ double g_zero_rec[32];

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
  int nt;                   // bit 0: nontemporal operator-stream loads, bit 1: nontemporal stores (GLX_NT / large operators)
  const int32_t* dup_ptr;   // HAS_DUP (boundary rows of a vertex-partitioned sweep): row r is ALSO stored at records dup_pos[dup_ptr[r] .. dup_ptr[r+1]) of dup_out
  const int32_t* dup_pos;   //   -- the send buffer of the halo exchange, so no pack kernel sits between the SpMM and the transport
  char* dup_out;
  int ablate;               // developer probe (GLX_ABLATE): 1 no chunk loop, 2 gathers hit one hot record, 4 no stores, 8 no XCD remap
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

// XCD-aware block -> slice-group map: the dispatcher places block b on XCD b % 8; hand
// each XCD a contiguous range of slices so rows that share neighbours share an L2.
// (the plan pads every range to nb/8 blocks, so the map is a plain transpose)
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb) {
  return (b % 8) * (nb / 8) + b / 8;
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
#ifndef GLX_FORCE_OCC6
#define GLX_FORCE_OCC6 0
#endif
#if GLX_FORCE_OCC6
#define GLX_SPMM_OCC __attribute__((amdgpu_waves_per_eu((G == 4 && !PERSIST && !HAS_DOT && sizeof(T) == 8) ? 6 : 1)))
#else
#define GLX_SPMM_OCC
#endif
template <typename T, int G, bool HAS_W, bool HAS_DOT, bool HAS_DUP = false, bool PERSIST = false>
__global__ __launch_bounds__(64 * GLX_WPB) GLX_SPMM_OCC void spmm_sell_kernel(const SpmmParams p) {
#pragma clang fp contract(off)
  typedef typename VecOf<T>::type V4;
  constexpr int R = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ double s_red[HAS_DOT ? GLX_WPB * 64 * 4 : 4];

  // the stop values of the previous sweep: the load is issued HERE, the test comes behind the first slice's loads (below), so that
  // the two round trips overlap instead of following each other in front of every wavefront's work
  unsigned long long stop_v = 0;
  if constexpr (HAS_W) {
    if (p.err_prev) stop_v = p.err_prev[lane];
  }
  if constexpr (HAS_DOT) {
    if (p.exit_err && !(*p.exit_err > p.exit_tol)) return;
  }
  // Persistent form (GLX_PERSIST = k: grid = nblocks / k workgroups): a workgroup walks the blocks j, j + gridDim/8, ... of ITS
  // XCD's range, so the header / first-chunk loads of its next block travel while the current one gathers, and the waves of a
  // CU are in different phases (one stores while another gathers) instead of all starting and all finishing together.
  // gridDim.x == nblocks is the one-block-per-workgroup form.
  const int64_t bpx = p.nblocks / 8, gpx = gridDim.x / 8;
  const bool flat = (p.ablate & 8) != 0;
  const int g = lane / G, c = lane % G;
  bool lane_on = c < p.nlanes;
  if constexpr (HAS_DOT) {
    // column groups (CG on several systems): lanes whose 4 columns all belong to converged
    // systems neither gather nor store
    if (p.act_row && lane_on) {
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = c * 4 + e;
        if (col < p.act_c) any = any || (p.act_row[col / p.act_cg] > p.exit_tol);
      }
      lane_on = any;
    }
  }
  const bool is_w = HAS_W && (c == p.nvec);
  const T* __restrict__ valp = (const T*)p.val;
  const size_t lane_off = (size_t)c * 4 * sizeof(T);
  unsigned long long err_run = 0;     // running max of this lane's stop values over the blocks it serves (bit patterns)

  // header + chunk 0 of one slice (chunk 0 sits at a slice-indexed address: its load is issued together with the header loads)
  struct SliceIn { int col0; T val0; int row, len, nchunks, S, full; int64_t base; };
  auto load_slice = [&](int64_t slice) -> SliceIn {
    SliceIn in;
    in.col0 = 0; in.val0 = 0; in.row = -1; in.len = 0; in.nchunks = 0; in.S = 1; in.full = 0; in.base = 0;
    if (slice < p.nslices) {
      if (p.nt & 1) {
        in.col0 = __builtin_nontemporal_load(&p.col[slice * 64 + lane]);
        in.val0 = __builtin_nontemporal_load(&valp[slice * 64 + lane]);
      } else {
        in.col0 = p.col[slice * 64 + lane];
        in.val0 = valp[slice * 64 + lane];
      }
      const int64_t slot = slice * R + g;
      in.row = p.slot_row[slot];
      in.len = p.slot_len[slot];
      const SliceHdr hd = p.slice_hdr[slice];
      in.base = p.head + hd.ptr - 64;   // chunk k >= 1 at base + k*64
      in.nchunks = (p.ablate & 1) ? 0 : hd.nchunks;
      in.S = hd.S & 0xff;
      in.full = hd.S >> 8;
    }
    return in;
  };
  auto vb_of = [&](int64_t it) -> int64_t {     // it-th block of this workgroup, or -1
    if (flat) { const int64_t v = (int64_t)blockIdx.x + it * gridDim.x; return v < p.nblocks ? v : -1; }
    const int64_t j = (int64_t)(blockIdx.x / 8) + it * gpx;
    return j < bpx ? (int64_t)(blockIdx.x % 8) * bpx + j : -1;
  };
  int64_t vb = vb_of(0);
  SliceIn nxt = load_slice(vb >= 0 ? vb * GLX_WPB + wave : p.nslices);
  if constexpr (HAS_W) {
    if (p.err_prev) {   // stop test of ssl.py:667, decided identically by every wavefront
      // (`while ... np.max(np.absolute(v-vinf)) > 1/n`: a NaN maximum compares False and ends the loop too;
      //  NaN errors are recorded as a bit pattern above +inf, so they dominate the max like numpy's)
      const unsigned long long m = wave_max_u64(stop_v);
      if (m <= p.thresh_bits || m > 0x7ff0000000000000ull) return;
    }
  }
  for (int64_t it = 0; vb >= 0; ++it) {
  const int64_t slice = vb * GLX_WPB + wave;
  SliceIn cur;
  int64_t vb_next = -1;
  if constexpr (PERSIST) {
    cur = nxt;
    vb_next = vb_of(it + 1);
    if (vb_next >= 0) nxt = load_slice(vb_next * GLX_WPB + wave);    // in flight during this block's chunk loop
  } else {
    cur = it == 0 ? nxt : load_slice(slice);                           // one block per workgroup: nothing to look ahead to
  }
  const int col0 = cur.col0;
  const T val0 = cur.val0;
  const int row = cur.row, len = cur.len, nchunks = cur.nchunks, S = cur.S;
  // chunks [0, full): every lane of every slot has a real entry (plan) and every lane of a row is in use -> no predicates
  const int full = (p.nlanes == G && !(HAS_DOT && p.act_row)) ? (cur.full < nchunks ? cur.full : nchunks) : 0;
  const int64_t base = cur.base;
  const int seg = g & (S - 1);            // S is a power of two
  V4 acc = {0, 0, 0, 0};
  double accw = 0.0;

  if constexpr (G == 4) {
    // Software pipeline: index/value chunks travel 3 chunks ahead of their use, the
    // neighbour gathers of chunk k+1 are in flight while chunk k is being added up.
    struct CV { int col; T val; };
    // unconditional (index clamped to the slice's last chunk): a conditional load would make the
    // compiler wait for it at the branch join, i.e. before this chunk's gathers are even issued
    auto load_cv = [&](int k) -> CV {
      CV r;
      const int kc = k < nchunks ? k : nchunks - 1;
      if (kc <= 0) {          // (also the clamp target of a 1-chunk slice)
        r.col = col0;
        r.val = val0;
      } else if (p.nt & 1) {
        r.col = __builtin_nontemporal_load(&p.col[base + (int64_t)kc * 64 + lane]);
        r.val = __builtin_nontemporal_load(&valp[base + (int64_t)kc * 64 + lane]);
      } else {
        r.col = p.col[base + (int64_t)kc * 64 + lane];
        r.val = valp[base + (int64_t)kc * 64 + lane];
      }
      return r;
    };
    auto issue = [&](const CV& cv, int k, V4 (&x)[4], T (&v)[4]) {
      int c0 = quad_bcast_i<0>(cv.col), c1 = quad_bcast_i<1>(cv.col), c2 = quad_bcast_i<2>(cv.col), c3 = quad_bcast_i<3>(cv.col);
#ifdef GLX_ABLATE_BUILD      // developer probe (gathers hit 16 hot records): compile-time only, a branch here splits the chunk loop's blocks
      if (p.ablate & 2) { c0 &= 15; c1 &= 15; c2 &= 15; c3 &= 15; }
#endif
      v[0] = quad_bcast<0>(cv.val);
      v[1] = quad_bcast<1>(cv.val);
      v[2] = quad_bcast<2>(cv.val);
      v[3] = quad_bcast<3>(cv.val);
      const int j0 = (k * S + seg) * 4;   // first row entry this slot holds in chunk k
      x[0] = V4{0, 0, 0, 0};
      x[1] = V4{0, 0, 0, 0};
      x[2] = V4{0, 0, 0, 0};
      x[3] = V4{0, 0, 0, 0};
#if GLX_OFF32
      const unsigned lo32 = (unsigned)lane_off, rb = (unsigned)p.rec_bytes;
      if (lane_on && j0 + 0 < len) x[0] = *(const V4*)(p.xin + ((unsigned)c0 * rb + lo32));
      if (lane_on && j0 + 1 < len) x[1] = *(const V4*)(p.xin + ((unsigned)c1 * rb + lo32));
      if (lane_on && j0 + 2 < len) x[2] = *(const V4*)(p.xin + ((unsigned)c2 * rb + lo32));
      if (lane_on && j0 + 3 < len) x[3] = *(const V4*)(p.xin + ((unsigned)c3 * rb + lo32));
#else
      if (lane_on && j0 + 0 < len) x[0] = *(const V4*)(p.xin + (size_t)c0 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 1 < len) x[1] = *(const V4*)(p.xin + (size_t)c1 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 2 < len) x[2] = *(const V4*)(p.xin + (size_t)c2 * p.rec_bytes + lane_off);
      if (lane_on && j0 + 3 < len) x[3] = *(const V4*)(p.xin + (size_t)c3 * p.rec_bytes + lane_off);
#endif
    };
    auto consume = [&](int k, const V4 (&x)[4], const T (&v)[4]) {
      if (S == 1) {
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
#if GLX_PIPE_DEPTH >= 2
    V4 xA[4], xB[4];
    T vA[4], vB[4];
    CV q0, q1, q2, q3;
    q0.col = q1.col = q2.col = q3.col = 0;
    q0.val = q1.val = q2.val = q3.val = 0;
    if (nchunks > 0) { q0 = load_cv(0); q1 = load_cv(1); q2 = load_cv(2); issue(q0, 0, xA, vA); }
    int k = 0;
    while (k < nchunks) {
      q3 = load_cv(k + 3); issue(q1, k + 1, xB, vB); consume(k, xA, vA); if (++k >= nchunks) break;
      q0 = load_cv(k + 3); issue(q2, k + 1, xA, vA); consume(k, xB, vB); if (++k >= nchunks) break;
      q1 = load_cv(k + 3); issue(q3, k + 1, xB, vB); consume(k, xA, vA); if (++k >= nchunks) break;
      q2 = load_cv(k + 3); issue(q0, k + 1, xA, vA); consume(k, xB, vB); ++k;
    }
#else
    V4 xA[4];
    T vA[4];
#if GLX_LOOP_FORM == 2 || GLX_LOOP_FORM == 3
    // Experimental (2: two chunks of gathers in flight; 3: one, like form 1, but without exec-mask branches).  Every load of the loop is unconditional -- lanes whose entry lies past the end
    // of the row gather the zero record instead of being masked off -- so that the compiler's wait-counter pass can count the
    // loads behind the ones it waits for (a load under an exec-mask branch makes it fall back to "all but the unconditional ones").
    auto load_cv2 = [&](int k) -> CV {
      const int kc = k < nchunks ? k : nchunks - 1;
      const int64_t off = (kc <= 0 ? slice * 64 : base + (int64_t)kc * 64) + lane;
      CV r;
      r.col = p.col[off];
      r.val = valp[off];
      return r;
    };
    const char* zrec = (const char*)g_zero_rec + lane_off;
    auto issue_u = [&](const CV& cv, int k, V4 (&x)[4], T (&v)[4]) {
      const int c0 = quad_bcast_i<0>(cv.col), c1 = quad_bcast_i<1>(cv.col), c2 = quad_bcast_i<2>(cv.col), c3 = quad_bcast_i<3>(cv.col);
      v[0] = quad_bcast<0>(cv.val);
      v[1] = quad_bcast<1>(cv.val);
      v[2] = quad_bcast<2>(cv.val);
      v[3] = quad_bcast<3>(cv.val);
      const int j0 = (k * S + seg) * 4;
      const char* a0 = (lane_on && j0 + 0 < len) ? p.xin + (size_t)c0 * p.rec_bytes + lane_off : zrec;
      const char* a1 = (lane_on && j0 + 1 < len) ? p.xin + (size_t)c1 * p.rec_bytes + lane_off : zrec;
      const char* a2 = (lane_on && j0 + 2 < len) ? p.xin + (size_t)c2 * p.rec_bytes + lane_off : zrec;
      const char* a3 = (lane_on && j0 + 3 < len) ? p.xin + (size_t)c3 * p.rec_bytes + lane_off : zrec;
      x[0] = *(const V4*)a0;
      x[1] = *(const V4*)a1;
      x[2] = *(const V4*)a2;
      x[3] = *(const V4*)a3;
    };
#if GLX_LOOP_FORM == 3
    {
      CV qn;
      qn.col = col0;
      qn.val = val0;
      for (int k = 0; k < nchunks; ++k) {
        const CV qc = qn;
        issue_u(qc, k, xA, vA);
        qn = load_cv2(k + 1);
        consume(k, xA, vA);
      }
    }
#else
    V4 xB[4];
    T vB[4];
    if (nchunks > 0) {
      CV c0v;
      c0v.col = col0;
      c0v.val = val0;
      CV c1v = load_cv2(1);            // older than the gathers of chunk 0: it arrives first
      CV c2v;
      issue_u(c0v, 0, xA, vA);
      int k = 0;
      while (true) {
        if (k + 1 >= nchunks) { consume(k, xA, vA); break; }
        c2v = load_cv2(k + 2);
        issue_u(c1v, k + 1, xB, vB);
        consume(k, xA, vA);
        ++k;
        if (k + 1 >= nchunks) { consume(k, xB, vB); break; }
        c1v = load_cv2(k + 2);
        issue_u(c2v, k + 1, xA, vA);
        consume(k, xB, vB);
        ++k;
      }
    }
#endif
#elif GLX_LOOP_FORM
    // Round 3.  The round-1 order issued the next chunk's index / value load FIRST and the gathers behind it; the load sat in a
    // branch (chunk 0 comes from registers, later chunks from memory), and the compiler's wait-counter pass answers a load in a
    // branch with `s_waitcnt vmcnt(0)` at the join -- so every chunk waited for the NEXT chunk's indices before its own gathers
    // were even issued: two memory round trips per chunk instead of one (read off the ISA).  Now the address is selected, not the
    // value (chunk 0 is re-read from its slice-indexed place), the load is unconditional and it is issued BEHIND the gathers: the
    // wait in front of the adds is `vmcnt(2)` -- the gathers, not the two youngest loads -- and the next chunk's indices have the
    // whole gather round trip to arrive.
    auto load_cv2 = [&](int k) -> CV {
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
#if GLX_OFF32
      const unsigned lo32 = (unsigned)lane_off, rb = (unsigned)p.rec_bytes;
      x[0] = *(const V4*)(p.xin + ((unsigned)c0 * rb + lo32));
      x[1] = *(const V4*)(p.xin + ((unsigned)c1 * rb + lo32));
      x[2] = *(const V4*)(p.xin + ((unsigned)c2 * rb + lo32));
      x[3] = *(const V4*)(p.xin + ((unsigned)c3 * rb + lo32));
#else
      x[0] = *(const V4*)(p.xin + (size_t)c0 * p.rec_bytes + lane_off);
      x[1] = *(const V4*)(p.xin + (size_t)c1 * p.rec_bytes + lane_off);
      x[2] = *(const V4*)(p.xin + (size_t)c2 * p.rec_bytes + lane_off);
      x[3] = *(const V4*)(p.xin + (size_t)c3 * p.rec_bytes + lane_off);
#endif
    };
    CV qn;
    qn.col = col0;
    qn.val = val0;
    int k = 0;
#if GLX_FULL_CHUNKS == 2
    for (; k < nchunks; ++k) {          // one loop, the gather form picked per chunk (uniform branch)
      const CV qc = qn;
      if (k < full) issue_full(qc, xA, vA); else issue(qc, k, xA, vA);
      qn = load_cv2(k + 1);
      consume(k, xA, vA);
    }
#else
#if GLX_FULL_CHUNKS
    if (sizeof(T) == 4 || GLX_FULL_CHUNKS == 3)     // fp64: the second loop costs 4 registers and with them a wavefront per SIMD (measured: slower)
      for (; k < full; ++k) {
        const CV qc = qn;
        issue_full(qc, xA, vA);
        qn = load_cv2(k + 1);
        consume(k, xA, vA);
      }
#endif
    for (; k < nchunks; ++k) {
      const CV qc = qn;
      issue(qc, k, xA, vA);
      qn = load_cv2(k + 1);
      consume(k, xA, vA);
    }
#endif
#else
    CV qn;
    qn.col = 0;
    qn.val = 0;
    if (nchunks > 0) qn = load_cv(0);
    for (int k = 0; k < nchunks; ++k) {
      const CV qc = qn;
      qn = load_cv(k + 1);      // next chunk's indices/values travel while this chunk's gathers do
      issue(qc, k, xA, vA);
      consume(k, xA, vA);
    }
#endif
#endif
  } else {
    const int gbase = lane & ~(G - 1);
    for (int k = 0; k < nchunks; ++k) {
      const int colv = k == 0 ? col0 : p.col[base + (int64_t)k * 64 + lane];
      const T valv = k == 0 ? val0 : valp[base + (int64_t)k * 64 + lane];
      const int j0 = k * G;
      for (int tb = 0; tb < G; tb += 4) {
        int cj[4];
        T vj[4];
        bool aj[4];
        V4 xj[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          cj[t] = __shfl(colv, gbase + tb + t);
          vj[t] = shfl_t(valv, gbase + tb + t);
          aj[t] = lane_on && (j0 + tb + t < len);
          xj[t] = V4{0, 0, 0, 0};
          if (aj[t]) xj[t] = *(const V4*)(p.xin + (size_t)cj[t] * p.rec_bytes + lane_off);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) accum4<T, HAS_W>(acc, accw, vj[t], xj[t], is_w);
      }
    }
  }

  // epilogue: u_out[row] = Db[row] + acc   (ssl.py:668: `Db + P*u`; addition commutes bitwise)
  V4 outv = acc;
  const bool store_on = lane_on && row >= 0 && seg == 0;
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
      }
    }
    if (!(p.ablate & 4)) {
      if (p.nt & 2)
        __builtin_nontemporal_store(outv, (V4*)(p.xout + (size_t)row * p.rec_bytes + lane_off));
      else
        *(V4*)(p.xout + (size_t)row * p.rec_bytes + lane_off) = outv;
    }
    if constexpr (HAS_DUP) {
      // a boundary row leaves for its peers straight from the registers: one more store per destination
      for (int q = p.dup_ptr[row], q1 = p.dup_ptr[row + 1]; q < q1; ++q)
        *(V4*)(p.dup_out + (size_t)p.dup_pos[q] * p.rec_bytes + lane_off) = outv;
    }
  }

  if constexpr (HAS_W) {
    if (p.err_next) {   // max_i |v_i - vinf_i| with v = deg * w  (ssl.py:667)
      double e = 0.0;
      if (store_on && is_w) {
        double wnew;
        if constexpr (sizeof(T) == 4) wnew = accw; else wnew = (double)outv[0];
        e = fabs(p.deg[row] * wnew - p.vinf[row]);
        if (e != e) e = __longlong_as_double(0x7ff8000000000000ll);   // canonical NaN: orders above +inf as a bit pattern (np.max propagates NaN)
      }
      const unsigned long long eb = (unsigned long long)__double_as_longlong(e);
      err_run = eb > err_run ? eb : err_run;
    }
  }

  if constexpr (HAS_DOT) {
    // column dots sum_rows xin[row,c] * xout[row,c] (utils.py:524 `np.sum(p*Ap,axis=0)`):
    // fixed-order tree inside the block, one partial row per block, reduced by the consumer.
    double d[4] = {0, 0, 0, 0};
    if (store_on) {
      const V4 own = *(const V4*)(p.xin + (size_t)row * p.rec_bytes + lane_off);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = (double)own[e] * (double)outv[e];
      if (p.prod_out && c < p.nvec) {   // elementwise p*Ap in the array dtype, row-major (dot_ld columns) in the caller's row order
        const V4 pr = own * outv;
        const int64_t orow = p.perm ? p.perm[row] : row;
        f64x4 pd;
#pragma unroll
        for (int e = 0; e < 4; ++e) pd[e] = (double)pr[e];
        *(f64x4*)(p.prod_out + ((size_t)((c * 4) / p.prod_sc) * p.n_rows + orow) * p.prod_sc + (c * 4) % p.prod_sc) = pd;
      }
    }
#pragma unroll
    for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] += shfl_d(d[e], lane ^ off);
    }
    if (lane < G) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s_red[(wave * 64 + lane) * 4 + e] = d[e];
    }
    __syncthreads();
    if (threadIdx.x < G && threadIdx.x < p.nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double s = s_red[(0 * 64 + threadIdx.x) * 4 + e];
        for (int w = 1; w < GLX_WPB; ++w) s += s_red[(w * 64 + threadIdx.x) * 4 + e];
        p.dot_partial[(size_t)vb * p.dot_ld + threadIdx.x * 4 + e] = s;
      }
    }
    __syncthreads();   // s_red is reused by the workgroup's next block
  }
  vb = vb_next;
  }   // blocks of this workgroup

  if constexpr (HAS_W) {
    if (p.err_next) {
      const unsigned long long m = wave_max_u64(err_run);
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
}

int64_t glx_spmm_blocks(const SellPlan* plan) { return (plan->nslices + GLX_WPB - 1) / GLX_WPB; }

template <typename T, int G>
static int launch_g(const SweepArgs& a, const SpmmParams& p, hipStream_t stream) {
  // GLX_PERSIST = k > 1: k blocks per workgroup (persistent form, sweeps only: the CG form writes per-block partials)
  static const int persist = getenv("GLX_PERSIST") ? atoi(getenv("GLX_PERSIST")) : GLX_PERSIST_DEFAULT;
  int64_t nwg = p.nblocks;
  if (persist > 1 && !a.dot_partial && p.nblocks >= 8 * (int64_t)persist) nwg = ((p.nblocks / 8 + persist - 1) / persist) * 8;
  const dim3 grid((unsigned)nwg), block(64 * GLX_WPB);
  const bool persistent = nwg != p.nblocks;
  if (a.dot_partial) {
    hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, true>), grid, block, 0, stream, p);
  } else if (a.has_w && a.dup_ptr) {
    if (persistent) hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, false, true, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, false, true>), grid, block, 0, stream, p);
  } else if (a.has_w) {
    if (persistent) hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, false, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((spmm_sell_kernel<T, G, true, false>), grid, block, 0, stream, p);
  } else {
    if (persistent) hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, false, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((spmm_sell_kernel<T, G, false, false>), grid, block, 0, stream, p);
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
  GLX_CHECK(!(a.dot_partial && a.has_w), GLX_EINVAL, "spmm: dot and stop column are exclusive");
  GLX_CHECK(!a.dup_ptr || (a.has_w && a.dup_pos && a.dup_out), GLX_EINVAL, "spmm: the send-buffer scatter needs the stop column form and all three arrays");
  GLX_CHECK(a.plan->G == a.L.G, GLX_EINVAL, "spmm: plan G=%d but layout G=%d", a.plan->G, a.L.G);
  if (a.plan->nslices == 0) return GLX_OK;
  SpmmParams p;
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
  static const int ablate = getenv("GLX_ABLATE") ? atoi(getenv("GLX_ABLATE")) : 0;
  p.ablate = ablate;
  static const int nt_env = getenv("GLX_NT") ? atoi(getenv("GLX_NT")) : -1;
  p.nt = nt_env >= 0 ? nt_env : 0;
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
