// Multi right-hand-side conjugate gradient on device: utils.conjgrad of the reference
// (graphlearning/utils.py:483-532).  Used by ssl.poisson (default solver, ssl.py:624-629) and
// ssl.laplace (ssl.py:1249).  This file: the REFERENCE-ORDER solve -- every reduction adds in numpy's
// order, so iterates and iteration counts are bit-identical to the reference.  Per iteration: one
// sliced-ELL SpMM that also writes the products p*Ap, one fused x/r update that writes r*r, one p
// update, and two reduction chains (one wavefront per four columns).  The host reads the residual
// history in chunks; kernels of iterations past convergence exit at once (same trick as the sweep's
// stop column).  The tolerance mode (GLX_CG_TREE) lives in cg_fused.hip.
#include "cg_internal.h"
#include <map>
#include <string.h>
#include <algorithm>
#include <vector>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename T> struct V4Of;
template <> struct V4Of<float> { typedef f32x4 type; };
template <> struct V4Of<double> { typedef f64x4 type; };

static const int CG_CHUNK = 8;
static const int UPD_ROWS_PER_BLOCK = 256;

// x += alpha p ; r -= alpha Ap ; partial[b][c] = sum_rows r^2           (utils.py:525-527)
// MODE 0: that update.  MODE 1 (init): r = p = b given in r; partial = sum r^2 (utils.py:514-517)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void cg_update_kernel(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p,
                                                        const T* __restrict__ Ap, const double* __restrict__ alpha,
                                                        double* __restrict__ partial, int64_t n, int ld, int nvec,
                                                        const CgScalars sc, int it, double tol,
                                                        double* __restrict__ prod_out, int prod_sc,
                                                        const int32_t* __restrict__ perm) {
#pragma clang fp contract(off)
  typedef typename V4Of<T>::type V4;
  if (MODE == 0 && !cg_any_active(sc, it, tol)) return;
  __shared__ double s_part[256 * 4];
  const int nvq = ld / 4;
  const int rows_pass = 256 / nvq;
  const int cv = threadIdx.x % nvq, rs = threadIdx.x / nvq;
  const bool on = rs < rows_pass && cv < nvec;
  double acc[4] = {0, 0, 0, 0};
  V4 a = {0, 0, 0, 0};
  bool act[4] = {true, true, true, true};
  bool on_any = on;
  if (MODE == 0 && on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (T)alpha[cv * 4 + e];
      act[e] = cg_col_active(sc, it, tol, cv * 4 + e);
    }
    on_any = act[0] || act[1] || act[2] || act[3];   // all four columns converged: nothing to read or write
  }
  const int64_t r0 = (int64_t)blockIdx.x * UPD_ROWS_PER_BLOCK;
  const int64_t r1 = min(n, r0 + UPD_ROWS_PER_BLOCK);
  if (on_any) {
    for (int64_t row = r0 + rs; row < r1; row += rows_pass) {
      const size_t o = (size_t)row * ld + cv * 4;
      V4 rv = *(const V4*)(r + o);
      if (MODE == 0) {
        const V4 pv = *(const V4*)(p + o);
        const V4 apv = *(const V4*)(Ap + o);
        V4 xv = *(const V4*)(x + o);
        const V4 t1 = a * pv;
        const V4 xn = xv + t1;
        const V4 t2 = a * apv;
        const V4 rn = rv - t2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // columns of a converged group keep their values
          xv[e] = act[e] ? xn[e] : xv[e];
          rv[e] = act[e] ? rn[e] : rv[e];
        }
        *(V4*)(x + o) = xv;
        *(V4*)(r + o) = rv;
      }
      if (prod_out) {   // r*r in the array dtype, row-major (nvec*4 columns) in the caller's row order
        const V4 sq = rv * rv;
        const int64_t orow = perm ? perm[row] : row;
        f64x4 sd;
#pragma unroll
        for (int e = 0; e < 4; ++e) sd[e] = (double)sq[e];
        *(f64x4*)(prod_out + ((size_t)((cv * 4) / prod_sc) * n + orow) * prod_sc + (cv * 4) % prod_sc) = sd;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double q = (double)rv[e] * (double)rv[e];
        acc[e] = acc[e] + q;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) s_part[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < nvec * 4) {
    const int c = threadIdx.x, ccv = c / 4, ce = c % 4;
    double s = 0.0;
    for (int q = 0; q < rows_pass; ++q) s += s_part[(q * nvq + ccv) * 4 + ce];
    partial[(size_t)blockIdx.x * (nvec * 4) + c] = s;
  }
}

// err_g = np.sqrt(np.sum(rsnew_g)) for every group still running (utils.py:528), then the
// maximum over those groups (what the kernels of the next iteration test).  np.sum over a
// contiguous 1-D float64 array is numpy's pairwise_sum: 8 accumulators below 128 elements, a plain
// loop below 8 (Cg <= 128 here).  One thread per group.
__device__ __forceinline__ void cg_group_err_body(const CgScalars& sc, int it, double tol, double* s_e) {
#pragma clang fp contract(off)
  // (the rows of the residual history are a ring of CG_CHUNK + 1 entries reused chunk after chunk: an iteration that does not run
  // writes the zeros -- "stopped" -- that a fresh buffer would hold, so that everything behind it stays off as well)
  const bool any = cg_any_active(sc, it, tol);
  const int g = threadIdx.x;
  double mine = 0.0;
  if (any && g < sc.ngroups && cg_col_active(sc, it, tol, g * sc.Cg)) {
    const double* a = sc.rsold + (size_t)g * sc.Cg;
    const int C = sc.Cg;
    double e;
    if (C < 8) {
      e = 0.0;
      for (int q = 0; q < C; ++q) e = e + a[q];
    } else {
      double r8[8];
      for (int q = 0; q < 8; ++q) r8[q] = a[q];
      int i = 8;
      for (; i < C - (C % 8); i += 8)
        for (int q = 0; q < 8; ++q) r8[q] = r8[q] + a[i + q];
      e = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
      for (; i < C; ++i) e = e + a[i];
    }
    mine = sqrt(e);
  }
  __syncthreads();        // (every thread has read row it - 1 before row `it` is written: with a one-row ring they could be the same)
  if (g < sc.ngroups) sc.err_hist[(size_t)it * sc.stride + g] = mine;
  s_e[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    // a NaN error stops its own group (`nan > tol` is false) and must not keep the others alive
    double m = 0.0;
    for (int q = 0; q < sc.ngroups; ++q)
      if (s_e[q] > m) m = s_e[q];
    sc.err_hist[(size_t)it * sc.stride + sc.ngroups] = m;
  }
}
// p = r + beta p                                                          (utils.py:529)
template <typename T>
__global__ __launch_bounds__(256) void cg_pupdate_kernel(const T* __restrict__ r, T* __restrict__ p,
                                                         const double* __restrict__ beta, int64_t n, int ld, int nvec,
                                                         const CgScalars sc, int it, double tol, int with_err) {
#pragma clang fp contract(off)
  typedef typename V4Of<T>::type V4;
  // with_err == 2: the launch has one workgroup more than the rows need, and that one forms the iteration's residual norms (the row of
  // the history the NEXT iteration tests; this kernel tests the previous row like everything else of iteration `it`) -- one launch of a
  // single workgroup less per iteration of the reference-order solve (4.6 us of its ~140)
  if (with_err == 2 && blockIdx.x == gridDim.x - 1) {
    __shared__ double s_e[256];
    cg_group_err_body(sc, it, tol, s_e);
    return;
  }
  // this kernel belongs to iteration `it`: it runs iff the iteration ran
  if (!cg_any_active(sc, it, tol)) return;
  const int nvq = ld / 4;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = v / nvq;
  const int cv = (int)(v % nvq);
  if (row >= n || cv >= nvec) return;
  bool any = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) any = any || cg_col_active(sc, it, tol, cv * 4 + e);
  if (!any) return;
  V4 b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = (T)beta[cv * 4 + e];
  const size_t o = (size_t)row * ld + cv * 4;
  const V4 rv = *(const V4*)(r + o);
  const V4 pv = *(const V4*)(p + o);
  const V4 t = b * pv;
  V4 pn = rv + t;
#pragma unroll
  for (int e = 0; e < 4; ++e) pn[e] = cg_col_active(sc, it, tol, cv * 4 + e) ? pn[e] : pv[e];
  *(V4*)(p + o) = pn;
}

// the ring of the residual history turns: row 0 (what iteration 1 of a chunk tests) <- row `from` (what the previous chunk's last iteration left)
__global__ void cg_roll_hist_kernel(double* err_hist, int stride, int from) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < stride) err_hist[i] = err_hist[(size_t)from * stride + i];
}

// Dirichlet rows of every system as bits of rowmask[record] (zeroed before): the SpMM holds A p at zero there (sweep.hip)
__global__ __launch_bounds__(256) void cg_rowmask_kernel(unsigned* __restrict__ rowmask, const int32_t* __restrict__ mask_rows,
                                                         const int32_t* __restrict__ mask_ptr) {
  const int g = blockIdx.y;
  const int cnt = mask_ptr[g + 1] - mask_ptr[g];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < cnt) atomicOr(&rowmask[mask_rows[mask_ptr[g] + idx]], 1u << g);
}

// ---- reference-order reductions ---------------------------------------------------------
// numpy's `np.sum(a*b, axis=0)` on a C-contiguous (n,k) array accumulates row after row
// (out[c] += a[i,c]*b[i,c], i ascending), a strictly sequential rounding chain per column.
// Reproducing it makes the whole CG bit-identical to the reference (iteration count
// included) -- necessary because the Poisson system is singular and 100+ CG iterations
// amplify any reordering far beyond 1e-5.  The elementwise products are formed in parallel
// by the producing kernels (SpMM epilogue / r-update) into an array in the caller's row order,
// blocked by 4 columns; here one wavefront per block adds up its columns row after row -- only
// the add chain is serial.
// MODE 0: tot = sum p*Ap ; alpha = rsold / tot                                (utils.py:524)
// MODE 1: tot = sum r*r  ; beta = tot / rsold ; rsold = tot ; err = sqrt(np.sum(tot)) (:527-530)
// MODE 2: rsold = sum r*r                                                     (utils.py:517)
// A wavefront issues one instruction every 4 cycles and a dependent fp64 add every ~6, so every
// instruction that is not an add stretches the chain (an LDS-staged version with one 16-byte read
// per 2 rows ran at 4.1 ns per row; this one at 2.4, the latency of the add).  One wavefront owns 4
// columns, one per DPP row of 16 lanes: lane k of row r loads element (16 g + k, column r) straight
// from global memory (the product array is blocked by 4 columns, so a wavefront load is 512
// contiguous bytes), and `v_fmac_f64_dpp tot, x, 1.0 row_newbcast:k` adds the value held by lane k
// to the running sum of every lane of its row: ONE load instruction per 16 rows, 16 dependent
// fused multiply-adds x*1+tot (= the correctly rounded sum, bit for bit the add).  48 loads stay in
// flight to cover the memory latency; no LDS, no barriers.
#define GLX_DPP_L(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
// (one asm block: the compiler pads separate asm statements with s_nop, each an issue slot of the chain)
// (s_nop 1: a VALU write of the DPP source right in front of the block -- a compiler-made copy of
// `x` -- needs two wait states before a DPP read; the assembler block is opaque to the hazard recogniser)
#define GLX_DPP_ADD16()                                                                                                        \
  asm volatile("s_nop 1\n\t" GLX_DPP_L(0) GLX_DPP_L(1) GLX_DPP_L(2) GLX_DPP_L(3) GLX_DPP_L(4) GLX_DPP_L(5) GLX_DPP_L(6) GLX_DPP_L(7) GLX_DPP_L(8) \
                   GLX_DPP_L(9) GLX_DPP_L(10) GLX_DPP_L(11) GLX_DPP_L(12) GLX_DPP_L(13) GLX_DPP_L(14) GLX_DPP_L(15)              \
               : "+v"(tot)                                                                                                     \
               : "v"(x), "v"(one))

template <int MODE>
__global__ __launch_bounds__(64) void cg_seqsum_dpp_kernel(const double* __restrict__ prod_all, int64_t n, int ncols_all, int C,
                                                           CgScalars sc, int it, double tol) {
  if (MODE != 2 && !cg_any_active(sc, it, tol)) return;
  const int lane = threadIdx.x, r = lane >> 4, k = lane & 15;
  const int col0 = blockIdx.x * 4;
  if (MODE != 2) {   // nothing to do if every group with a column here has converged
    bool any = false;
    for (int cc = col0; cc < col0 + 4; ++cc) any = any || cg_col_active(sc, it, tol, cc);
    if (!any) return;
  }
  // element (row 16 g + k, column r) of this block at src[g * 64]
  const double* __restrict__ src = prod_all + (size_t)blockIdx.x * n * 4 + (size_t)k * 4 + r;
  const int64_t ngroups = (n + 15) / 16;   // the last one may be partial
  const int64_t nfull = n / 16;
  constexpr int DEPTH = 48;
  double buf[DEPTH];
  // rows past n read as +0: tot (which starts at +0 and can never become -0) + 0 == tot bit for bit
  auto load_checked = [&](int64_t g) -> double { return g * 16 + k < n ? src[g * 64] : 0.0; };
  double tot = 0.0;
  const double one = 1.0;
  int64_t g0 = 0;
  if (2 * DEPTH <= nfull) {
    // steady state: group g0+d is added, then the load of group g0+d+DEPTH is issued into the
    // register it just freed (no bounds check: a full group).  The loads and their waits are
    // written by hand: loads return in order and exactly DEPTH are outstanding whenever a group is
    // about to be added, so `s_waitcnt vmcnt(DEPTH-1)` is the exact wait for the oldest -- the
    // compiler's own counter analysis gives up at the loop back-edge and drains all of them
    // (vmcnt(0)) at the top of every iteration, exposing a full memory latency each time.
    static_assert(DEPTH == 48, "the vmcnt immediate below is DEPTH-1");
    const double* sbase = prod_all + (size_t)blockIdx.x * n * 4;            // uniform: lives in SGPRs
    unsigned voff = (unsigned)((k * 4 + r) * 8);                            // this lane's byte offset, group 0
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(buf[d]) : "v"(voff), "s"(sbase));
      voff += 512;
    }
    for (; g0 + 2 * DEPTH <= nfull; g0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        asm volatile("s_waitcnt vmcnt(47)\n\t" GLX_DPP_L(0) GLX_DPP_L(1) GLX_DPP_L(2) GLX_DPP_L(3) GLX_DPP_L(4) GLX_DPP_L(5) GLX_DPP_L(6)
                         GLX_DPP_L(7) GLX_DPP_L(8) GLX_DPP_L(9) GLX_DPP_L(10) GLX_DPP_L(11) GLX_DPP_L(12) GLX_DPP_L(13) GLX_DPP_L(14)
                             GLX_DPP_L(15)
                     : "+v"(tot)
                     : "v"(buf[d]), "v"(one));
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(buf[d]) : "v"(voff), "s"(sbase));
        voff += 512;
      }
    }
    // hand the buffers back to compiler-managed code: the operand ties every later use of buf[d]
    // to this wait (a plain copy of the register could otherwise be scheduled in front of it)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[d]));
  } else {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = load_checked(d);
  }
  // last rounds: the same with checked loads
  for (; g0 + DEPTH <= ngroups; g0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const double x = buf[d];
      GLX_DPP_ADD16();
      __builtin_amdgcn_sched_barrier(0);
      buf[d] = load_checked(g0 + d + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    if (g0 + d < ngroups) {
      const double x = buf[d];
      GLX_DPP_ADD16();
    }
  }
  if (k == 0) {
    const int gc = col0 + r;
    const bool live = gc < ncols_all && (MODE == 2 || gc >= C || cg_col_active(sc, it, tol, gc));
    if (!live) {
    } else if (MODE == 0) {
      sc.alpha[gc] = gc < C ? sc.rsold[gc] / tot : 0.0;
    } else if (MODE == 1) {
      sc.beta[gc] = gc < C ? tot / sc.rsold[gc] : 0.0;
      sc.rsold[gc] = tot;
    } else {
      sc.rsold[gc] = tot;
    }
  }
}

// numpy reduces a contiguous 1-D float64 array with pairwise summation (umath
// DOUBLE_pairwise_sum: blocks of <= 128 elements summed with 8 strided accumulators combined as a
// fixed tree, halves split at a multiple of 8), applied to pieces of 8192 elements (its buffer size)
// whose sums are added one after another.  utils.conjgrad called with a 1-D right-hand side
// (graph.reweight, graph.py:429) takes that path, so its device twin reproduces the same tree.
// (`a` holds the elements `st` doubles apart: column 0 of the row-major product array)
// The recursion tree is fixed by n alone, so it is parallel: the host lists the leaves (blocks of
// at most 128 elements) and the internal nodes by height; one thread sums a leaf exactly as numpy
// does (8 strided accumulators, fixed combination, tail), then one workgroup adds the nodes level
// by level -- the same additions in the same association as the recursive routine.
__device__ __forceinline__ double np_leaf_sum(const double* __restrict__ a, int64_t n, int st) {
#pragma clang fp contract(off)
  if (n < 8) {
    double res = -0.0;
    for (int64_t i = 0; i < n; ++i) res = res + a[i * st];
    return res;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j * st];
  int64_t i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = r[j] + a[(i + j) * st];
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res = res + a[i * st];
  return res;
}

__global__ __launch_bounds__(256) void cg_pw_leaf_kernel(const double* __restrict__ prod, int st, PwPlan pw, CgScalars sc, int it,
                                                         double tol, int mode) {
  if (mode != 2 && !cg_any_active(sc, it, tol)) return;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < pw.nleaves) pw.vals[q] = np_leaf_sum(prod + pw.leaf_off[q] * st, pw.leaf_len[q], st);
}

template <int MODE>
__global__ __launch_bounds__(256) void cg_pw_tree_kernel(PwPlan pw, CgScalars sc, int it, double tol) {
#pragma clang fp contract(off)
  if (MODE != 2 && !cg_any_active(sc, it, tol)) return;
  for (int h = 0; h < pw.nlevels; ++h) {
    for (int q = pw.level_start[h] + threadIdx.x; q < pw.level_start[h + 1]; q += 256)
      pw.vals[pw.nleaves + q] = pw.vals[pw.node_l[q]] + pw.vals[pw.node_r[q]];
    __threadfence_block();
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double tot = 0.0 + pw.vals[pw.nleaves + pw.ninternal - 1];   // the root is the last value (a lone leaf if n <= 128)
  if (MODE == 0) {
    sc.alpha[0] = sc.rsold[0] / tot;
  } else if (MODE == 1) {
    sc.beta[0] = tot / sc.rsold[0];
    sc.rsold[0] = tot;
  } else {
    sc.rsold[0] = tot;
  }
  for (int c = 1; c < 4; ++c) {   // padded columns of the single 4-wide vector stay inert
    if (MODE == 0) sc.alpha[c] = 0.0;
    if (MODE == 1) sc.beta[c] = 0.0;
  }
}

// host side of the plan: the recursion of DOUBLE_pairwise_sum on (offset, n)
struct PwHost {
  std::vector<int64_t> leaf_off;
  std::vector<int32_t> leaf_len, node_l, node_r, node_h, level_start;
  // returns a provisional id: leaves >= 0, internal nodes as -(index + 1)
  int build(int64_t off, int64_t n, int* height) {
    if (n <= 128) {
      leaf_off.push_back(off);
      leaf_len.push_back((int32_t)n);
      *height = 0;
      return (int)leaf_off.size() - 1;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    int hl, hr;
    const int l = build(off, n2, &hl);
    const int r = build(off + n2, n - n2, &hr);
    *height = std::max(hl, hr) + 1;
    node_l.push_back(l);
    node_r.push_back(r);
    node_h.push_back(*height);
    return -(int)node_l.size();
  }
  // numpy hands a contiguous 1-D reduction to the pairwise routine in pieces of its buffer size
  // (8192 elements) and adds the pieces' sums one after another: ((c0 + c1) + c2) + ...
  void build_chunked(int64_t n) {
    const int64_t NPY_BUFSIZE = 8192;
    int h, hacc = 0;
    int acc = build(0, std::min(n, NPY_BUFSIZE), &hacc);
    for (int64_t off = NPY_BUFSIZE; off < n; off += NPY_BUFSIZE) {
      const int c = build(off, std::min(NPY_BUFSIZE, n - off), &h);
      hacc = std::max(hacc, h) + 1;
      node_l.push_back(acc);
      node_r.push_back(c);
      node_h.push_back(hacc);
      acc = -(int)node_l.size();
    }
  }
  void finish() {   // order the internal nodes by height (stable: children always precede parents) and renumber
    const int ni = (int)node_l.size(), nl = (int)leaf_off.size();
    std::vector<int> order(ni);
    for (int q = 0; q < ni; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return node_h[x] < node_h[y]; });
    std::vector<int> newpos(ni);
    for (int q = 0; q < ni; ++q) newpos[order[q]] = q;
    auto remap = [&](int id) { return id >= 0 ? id : nl + newpos[-id - 1]; };
    std::vector<int32_t> l2(ni), r2(ni);
    int maxh = 0;
    for (int q = 0; q < ni; ++q) {
      l2[q] = remap(node_l[order[q]]);
      r2[q] = remap(node_r[order[q]]);
      maxh = std::max(maxh, (int)node_h[order[q]]);
    }
    level_start.assign(maxh + 1, 0);
    for (int q = 0; q < ni; ++q) level_start[node_h[order[q]]]++;   // counts per height h >= 1 at index h
    // prefix: level_start[h-1] = first internal node of height h
    std::vector<int32_t> ls(maxh + 1, 0);
    int acc = 0;
    for (int h = 1; h <= maxh; ++h) { ls[h - 1] = acc; acc += level_start[h]; }
    ls[maxh] = acc;
    level_start = ls;
    node_l = l2;
    node_r = r2;
  }
};

// the ring of CG_CHUNK + 1 history rows: row 0 and the last row = 1 (utils.py:519: err = 1 in front of the loop; every chunk starts by
// copying the last row to row 0), the others 0 (= stopped)
__global__ void cg_set_err0(double* err_hist, int64_t n, int stride, int last_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) err_hist[i] = (i < stride || i / stride == last_row) ? 1.0 : 0.0;
}

// Dirichlet rows (ssl.laplace._fit, ssl.py:1232-1241, solves on the sub-matrix of the unlabelled
// vertices): on the FULL operator the same solve is obtained by keeping x, r, p at zero on the
// labelled rows -- their columns then multiply zeros, which adds exact zeros to every row sum
// and to every reduction chain -- and that only needs A p forced to zero there after each SpMM.
// mask_rows holds record indices, group g owns mask_rows[mask_ptr[g] .. mask_ptr[g+1]).
template <typename T>
__global__ __launch_bounds__(256) void cg_zero_rows_kernel(T* __restrict__ ap, int ld, const int32_t* __restrict__ mask_rows,
                                                           const int32_t* __restrict__ mask_ptr, int Cg, CgScalars sc, int it,
                                                           double tol) {
  if (!cg_any_active(sc, it, tol)) return;
  const int g = blockIdx.y;
  const int cnt = mask_ptr[g + 1] - mask_ptr[g];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)cnt * Cg) return;
  const int row = mask_rows[mask_ptr[g] + idx / Cg];
  ap[(size_t)row * ld + g * Cg + idx % Cg] = (T)0;
}

void glx_cg_ws_destroy(void* ws) { delete (CgBufs*)ws; }

// right-hand side rows given one by one (all other rows zero): r[rec(b_rows[q])][:] = b_vals[q][:]
template <typename T>
__global__ __launch_bounds__(256) void cg_scatter_rows_kernel(T* __restrict__ rec, int ld, int C, const int32_t* __restrict__ rows,
                                                              const T* __restrict__ vals, int64_t nb, const int32_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * C) return;
  const int64_t q = i / C;
  const int c = (int)(i % C);
  const int32_t row = rows[q];
  rec[(size_t)(inv ? inv[row] : row) * ld + c] = vals[i];
}

template <typename T>
static int cg_run(glx_graph* A, const void* B, void* X, int C, int Cg, double tol, int64_t max_iter, int* iters_out,
                  double* err_out, int flags, const int32_t* mask_rows, const int32_t* mask_ptr, const CgRhsRows& rr) {
  if (flags & GLX_CG_TREE)     // tolerance mode: cg_fused.hip (deterministic, not bit-identical to numpy's chains)
    return glx_cg_run_fused(A, B, X, C, Cg, tol, max_iter, iters_out, err_out, flags, mask_rows, mask_ptr, rr);
  const bool np1d = (flags & 1) && C == 1;   // caller passed a 1-D right-hand side: numpy's pairwise reductions
  const int ngroups = C / Cg;
  const int stride = ngroups + 1;
  const int64_t n = A->n_rows;
  const int dtype = A->dtype;
  RecLayout L;
  int rc = glx_make_layout(C, dtype, false, &L);
  if (rc) return rc;
  SellPlan* plan = nullptr;
  rc = glx_graph_plan(A, L.G, &plan);
  if (rc) return rc;
  const size_t es = L.esize;
  const int ncols = L.nvec * 4;
  GLX_CHECK(ncols <= 256, GLX_EUNSUPPORTED, "glx_cg_multi: C=%d too wide for the column reducer", C);
  GLX_CHECK(Cg <= 128, GLX_EUNSUPPORTED, "glx_cg_multi: %d columns per system too wide for the reference-order reducer", Cg);
  GLX_CHECK(256 / (L.ld / 4) >= 1, GLX_EUNSUPPORTED, "glx_cg_multi: record too wide");
  const int64_t nb_spmm = std::max<int64_t>(glx_spmm_blocks(plan), 1);
  const int64_t nb_upd = std::max<int64_t>((n + UPD_ROWS_PER_BLOCK - 1) / UPD_ROWS_PER_BLOCK, 1);
  const int64_t hist_cap = CG_CHUNK + 1;       // a ring: row 0 = what the chunk's first iteration tests, rows 1 .. CG_CHUNK = what its iterations leave
  GLX_CHECK(max_iter < (1ll << 24), GLX_EUNSUPPORTED, "glx_cg_multi: max_iter %lld exceeds the supported 2^24-1", (long long)max_iter);
  GLX_CHECK(n < (1ll << 27), GLX_EUNSUPPORTED, "glx_cg_multi: %lld rows exceed the reference-order reducer's 32-bit offsets", (long long)n);

  // reference-order reducer: one wavefront per 4 columns; the product array is blocked the same way
  const int prod_sc = 4;
  const unsigned seq_grid = (unsigned)(ncols / 4);
  if (!A->cg_ws) A->cg_ws = new CgBufs();
  CgBufs& b = *(CgBufs*)A->cg_ws;
  const size_t recb = std::max<size_t>((size_t)n * L.ld * es, 64);
  if (!b.stream) GLX_HIP(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
  CG_NEED(b.x, recb);
  CG_NEED(b.r, recb);
  CG_NEED(b.p, recb);
  CG_NEED(b.ap, recb);
  CG_NEED(b.dense, (size_t)n * C * es);
  CG_NEED(b.part_dot, (size_t)nb_spmm * ncols * 8);
  CG_NEED(b.part_rs, (size_t)nb_upd * ncols * 8);
  CG_NEED(b.scal, (size_t)3 * ncols * 8);
  CG_NEED(b.err_hist, (size_t)hist_cap * stride * 8);
  CG_NEED(b.prod, (size_t)ncols * n * 8);
  // the two reference-order chains per iteration: walked row by row (cg_seqsum_dpp_kernel, 2.4 ns per row) or in block form
  // (cg_seqsum.hip: integer block sums confirmed by the exact state; same bits) -- the block form from 8192 rows on
  SsWork ssw;
  memset(&ssw, 0, sizeof(ssw));
  ssw.nchunks = glx_seqsum_chunks(n);
  const bool ss_blocks = !np1d && !(flags & GLX_CG_CHAIN) && ssw.nchunks <= glx_seqsum_max_chunks() && n >= 1 &&
                         glx_seqsum_rec_bytes(ncols, ssw.nchunks) <= ((size_t)1 << 30) &&      // (hundreds of columns x millions of rows: the chain)
                         ((flags & GLX_CG_BLOCKS) || n >= 8192);
  if (ss_blocks) {
    rc = glx_seqsum_prepare();
    if (rc) return rc;
    CG_NEED(b.ss_bsum, glx_seqsum_sum_doubles(ncols, ssw.nchunks, 0) * 8);
    CG_NEED(b.ss_csum, glx_seqsum_sum_doubles(ncols, ssw.nchunks, 1) * 8);
    CG_NEED(b.ss_rec, glx_seqsum_rec_bytes(ncols, ssw.nchunks));
    CG_NEED(b.ss_mask, (size_t)ncols * ssw.nchunks * 16);      // a byte per group of 4 blocks
    CG_NEED(b.ss_stats, 128);
    ssw.rec = b.ss_rec;
    ssw.bsum = b.ss_bsum;
    ssw.csum = b.ss_csum;
    ssw.mask = b.ss_mask;
    ssw.stats = b.ss_stats;
    { int rc_ = b.need_host(&b.h_ss, 256); if (rc_) return rc_; }
    memset(b.h_ss, 0, 256);
  }
  // Per kind of reduction (0: p.Ap, 1: r.r) the form is re-decided whenever the host looks at the residual history: the block
  // form costs about a microsecond per block that is not a plain same-binade one (a lone wavefront issues an instruction every
  // four cycles), the chain 2.4 ns per row -- products that cancel (the singular Poisson system on separate clusters: p.Ap
  // changes sign from row to row) are cheaper row by row.  Same bits either way, so switching in mid-solve changes nothing but time.
  bool blocks_now[2] = {ss_blocks, ss_blocks};
  const bool blocks_forced = (flags & GLX_CG_BLOCKS) != 0;
  unsigned long long ss_seen[2][3] = {{0, 0, 0}, {0, 0, 0}};
  PwPlan pw;
  memset(&pw, 0, sizeof(pw));
  unsigned pw_grid = 1;
  if (np1d) {   // numpy's pairwise-summation tree for n elements (depends on n only: kept with the buffers)
    if (b.pw_n != n) {
      PwHost ph;
      ph.build_chunked(n);
      ph.finish();
      const size_t nl = ph.leaf_off.size(), ni = ph.node_l.size();
      CG_NEED(b.pw_off, nl * 8);
      CG_NEED(b.pw_len, nl * 4);
      CG_NEED(b.pw_l, std::max<size_t>(ni, 1) * 4);
      CG_NEED(b.pw_r, std::max<size_t>(ni, 1) * 4);
      CG_NEED(b.pw_ls, ph.level_start.size() * 4 + 4);
      CG_NEED(b.pw_vals, (nl + ni) * 8);
      GLX_UP(glx_upload_sync(b.pw_off, ph.leaf_off.data(), nl * 8, __func__));
      GLX_UP(glx_upload_sync(b.pw_len, ph.leaf_len.data(), nl * 4, __func__));
      if (ni) {
        GLX_UP(glx_upload_sync(b.pw_l, ph.node_l.data(), ni * 4, __func__));
        GLX_UP(glx_upload_sync(b.pw_r, ph.node_r.data(), ni * 4, __func__));
      }
      if (!ph.level_start.empty()) GLX_UP(glx_upload_sync(b.pw_ls, ph.level_start.data(), ph.level_start.size() * 4, __func__));
      b.pw.leaf_off = b.pw_off;
      b.pw.leaf_len = b.pw_len;
      b.pw.node_l = b.pw_l;
      b.pw.node_r = b.pw_r;
      b.pw.level_start = b.pw_ls;
      b.pw.vals = b.pw_vals;
      b.pw.nleaves = (int)nl;
      b.pw.ninternal = (int)ni;
      b.pw.nlevels = ph.level_start.empty() ? 0 : (int)ph.level_start.size() - 1;
      b.pw_grid = (unsigned)((nl + 255) / 256);
      b.pw_n = n;
    }
    pw = b.pw;
    pw_grid = b.pw_grid;
  }
  { int rc_ = b.need_host(&b.h_err, (size_t)2 * (CG_CHUNK + 1) * stride * 8); if (rc_) return rc_; }      // two chunks in flight
  CgScalars sc;
  sc.rsold = b.scal;
  sc.alpha = b.scal + ncols;
  sc.beta = b.scal + 2 * ncols;
  sc.err_hist = b.err_hist;
  sc.stride = stride;
  sc.ngroups = ngroups;
  sc.Cg = Cg;
  sc.C = C;
  hipStream_t st = b.stream;
  const dim3 blk(256);
  T* x = (T*)b.x;
  T* r = (T*)b.r;
  T* p = (T*)b.p;
  T* ap = (T*)b.ap;

  hipLaunchKernelGGL(cg_set_err0, dim3((unsigned)((hist_cap * stride + 255) / 256)), blk, 0, st, b.err_hist, hist_cap * stride, stride, CG_CHUNK);
  GLX_HIP(hipGetLastError());
  if (flags & GLX_CG_X0) {   // X holds x0 on entry; B is the caller's r0 = b - A@x0 (utils.py:510-514)
    GLX_UP(glx_upload(b.dense, X, (size_t)n * C * es, st, __func__));
    rc = glx_pack_records(b.dense, b.x, n, L, dtype, nullptr, st, A->d_perm);
    if (rc) return rc;
  } else {
    GLX_HIP(hipMemsetAsync(b.x, 0, recb, st));
  }
  GLX_HIP(hipMemsetAsync(b.ap, 0, recb, st));
  GLX_HIP(hipMemsetAsync(b.part_dot, 0, nb_spmm * ncols * 8, st));
  GLX_HIP(hipMemsetAsync(b.scal, 0, 3 * ncols * 8, st));
  if (rr.rows) {            // the nonzero rows only (ssl.laplace: the neighbours of the labelled vertices)
    GLX_HIP(hipMemsetAsync(b.r, 0, recb, st));
    if (rr.nb > 0) {
      CG_NEED(b.rhs_rows, (size_t)rr.nb * 4);
      GLX_UP(glx_upload(b.rhs_rows, rr.rows, (size_t)rr.nb * 4, st, __func__));
      GLX_UP(glx_upload(b.dense, rr.vals, (size_t)rr.nb * C * es, st, __func__));
      hipLaunchKernelGGL((cg_scatter_rows_kernel<T>), dim3((unsigned)((rr.nb * C + 255) / 256)), dim3(256), 0, st, (T*)b.r, L.ld, C,
                         (const int32_t*)b.rhs_rows, (const T*)b.dense, rr.nb, (const int32_t*)A->d_inv);
      GLX_HIP(hipGetLastError());
    }
  } else {
    GLX_UP(glx_upload(b.dense, B, (size_t)n * C * es, st, __func__));
    rc = glx_pack_records(b.dense, b.r, n, L, dtype, nullptr, st, A->d_perm);   // r = b - A@0 = b (utils.py:514)
    if (rc) return rc;
  }
  if (rr.out_scale) {
    CG_NEED(b.out_scale, (size_t)n * 8);
    GLX_UP(glx_upload(b.out_scale, rr.out_scale, (size_t)n * 8, st, __func__));
  }
  GLX_HIP(hipMemcpyAsync(b.p, b.r, recb, hipMemcpyDeviceToDevice, st));   // p = r.copy() (utils.py:516)
  hipLaunchKernelGGL((cg_update_kernel<T, 1>), dim3((unsigned)nb_upd), blk, 0, st, x, r, (const T*)p, (const T*)ap,
                     (const double*)sc.alpha, b.part_rs, n, L.ld, L.nvec, sc, 1, tol, b.prod, prod_sc, (const int32_t*)A->d_perm);
  GLX_HIP(hipGetLastError());
  if (np1d)
    {
      hipLaunchKernelGGL(cg_pw_leaf_kernel, dim3(pw_grid), dim3(256), 0, st, (const double*)b.prod, prod_sc, pw, sc, 0, tol, 2);
      hipLaunchKernelGGL(cg_pw_tree_kernel<2>, dim3(1), dim3(256), 0, st, pw, sc, 0, tol);
    }
  else if (ss_blocks) {
    GLX_HIP(hipMemsetAsync(b.ss_stats, 0, 128, st));
    rc = glx_seqsum_run(2, b.prod, n, ncols, C, sc, 0, tol, ssw, st);
    if (rc) return rc;
  } else
    hipLaunchKernelGGL(cg_seqsum_dpp_kernel<2>, dim3(seq_grid), dim3(64), 0, st, (const double*)b.prod, n, ncols, C, sc, 0, tol);
  GLX_HIP(hipGetLastError());

  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.plan = plan;
  a.L = L;
  a.dtype = dtype;
  a.xin = b.p;
  a.xout = b.ap;
  a.dot_partial = b.part_dot;   // the fused dot also carries the kernel's early-exit hook
  a.n_rows = n;
  a.exit_tol = tol;
  a.prod_out = b.prod;
  a.perm = A->d_perm;
  a.act_cg = Cg;
  a.act_c = C;
  a.prod_sc = prod_sc;
  // Dirichlet rows per group (caller row numbers -> record indices)
  unsigned mask_grid = 0;
  if (mask_rows && mask_ptr && mask_ptr[ngroups] > 0) {
    const int total = mask_ptr[ngroups];
    std::vector<int32_t> rec(total);
    int most = 0;
    for (int g = 0; g < ngroups; ++g) most = std::max(most, mask_ptr[g + 1] - mask_ptr[g]);
    for (int q = 0; q < total; ++q) {
      GLX_CHECK(mask_rows[q] >= 0 && mask_rows[q] < n, GLX_EINVAL, "glx_cg_groups_masked: row %d out of range", mask_rows[q]);
      rec[q] = A->order_ready && !A->h_inv.empty() ? A->h_inv[mask_rows[q]] : mask_rows[q];
    }
    CG_NEED(b.mask_rows, (size_t)total * 4);
    CG_NEED(b.mask_ptr, (size_t)(ngroups + 1) * 4);
    GLX_UP(glx_upload_sync(b.mask_rows, rec.data(), (size_t)total * 4, __func__));
    GLX_UP(glx_upload_sync(b.mask_ptr, mask_ptr, (size_t)(ngroups + 1) * 4, __func__));
    mask_grid = (unsigned)(((int64_t)most * Cg + 255) / 256);
    if (ngroups <= 32) {
      // the SpMM itself holds A p at zero on these rows (a bit per system in rowmask[record]): no kernel behind it.  (The products
      // p * Ap of such a row are 0 * 0 = +0 instead of 0 * (A p) = +-0: a zero of either sign leaves every running sum as it is.)
      CG_NEED(b.e_rowmask, (size_t)n * 4);
      GLX_HIP(hipMemsetAsync(b.e_rowmask, 0, (size_t)n * 4, st));
      hipLaunchKernelGGL(cg_rowmask_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)ngroups), blk, 0, st, b.e_rowmask,
                         (const int32_t*)b.mask_rows, (const int32_t*)b.mask_ptr);
      GLX_HIP(hipGetLastError());
      a.rowmask = b.e_rowmask;
      mask_grid = 0;
    }
  }
  const unsigned pgrid = (unsigned)std::max<int64_t>(((int64_t)n * (L.ld / 4) + 255) / 256, 1);

  std::vector<int64_t> iters(ngroups, 0);       // iterations that ran, per group (utils.py:522 `i`)
  std::vector<double> err(ngroups, 1.0);        // utils.py:519
  std::vector<char> done(ngroups, !(1.0 > tol));
  int running = 0;
  for (int g = 0; g < ngroups; ++g) running += !done[g];
  // rows [it0 + 1, it0 + cnt] of the residual history have arrived in `h`: which systems stopped where
  auto read_history = [&](const double* h, int64_t it0, int64_t cnt) {
    for (int64_t q = 0; q < cnt && running > 0; ++q) {
      // iteration it0+q+1 ran for every group whose previous err was > tol; its err decides the next one
      for (int g = 0; g < ngroups; ++g) {
        if (done[g]) continue;
        iters[g] = it0 + q + 1;
        err[g] = h[(size_t)q * stride + g];
        if (!(err[g] > tol)) { done[g] = 1; --running; }
      }
    }
  };
  // One iteration of utils.conjgrad's loop (utils.py:521-530) as it is enqueued: `i` = its row in the ring of the residual history
  // (1 .. CG_CHUNK; the kernels test row i - 1 and leave row i), `bl0` / `bl1` = the form of its two reduction chains.
  auto enqueue_iteration = [&](int i, bool bl0, bool bl1) -> int {
    a.exit_err = b.err_hist + (size_t)(i - 1) * stride + ngroups;   // all groups converged: exit at once
    a.act_row = ngroups > 1 ? b.err_hist + (size_t)(i - 1) * stride : nullptr;
    int rc2 = glx_launch_spmm(a, st);                                               // Ap = A@p, p.Ap partials
    if (rc2) return rc2;
    if (mask_grid) {
      hipLaunchKernelGGL((cg_zero_rows_kernel<T>), dim3(mask_grid, (unsigned)ngroups), blk, 0, st, ap, L.ld, (const int32_t*)b.mask_rows,
                         (const int32_t*)b.mask_ptr, Cg, sc, i, tol);
      GLX_HIP(hipGetLastError());
    }
    if (np1d) {
      hipLaunchKernelGGL(cg_pw_leaf_kernel, dim3(pw_grid), dim3(256), 0, st, (const double*)b.prod, prod_sc, pw, sc, i, tol, 0);
      hipLaunchKernelGGL(cg_pw_tree_kernel<0>, dim3(1), dim3(256), 0, st, pw, sc, i, tol);
    } else if (bl0) {
      rc2 = glx_seqsum_run(0, b.prod, n, ncols, C, sc, i, tol, ssw, st);
      if (rc2) return rc2;
    } else {
      hipLaunchKernelGGL(cg_seqsum_dpp_kernel<0>, dim3(seq_grid), dim3(64), 0, st, (const double*)b.prod, n, ncols, C, sc, i, tol);
    }
    GLX_HIP(hipGetLastError());
    hipLaunchKernelGGL((cg_update_kernel<T, 0>), dim3((unsigned)nb_upd), blk, 0, st, x, r, (const T*)p, (const T*)ap,
                       (const double*)sc.alpha, b.part_rs, n, L.ld, L.nvec, sc, i, tol, b.prod, prod_sc, (const int32_t*)A->d_perm);
    GLX_HIP(hipGetLastError());
    if (np1d) {
      hipLaunchKernelGGL(cg_pw_leaf_kernel, dim3(pw_grid), dim3(256), 0, st, (const double*)b.prod, prod_sc, pw, sc, i, tol, 1);
      hipLaunchKernelGGL(cg_pw_tree_kernel<1>, dim3(1), dim3(256), 0, st, pw, sc, i, tol);
    } else if (bl1) {
      rc2 = glx_seqsum_run(1, b.prod, n, ncols, C, sc, i, tol, ssw, st);
      if (rc2) return rc2;
    } else {
      hipLaunchKernelGGL(cg_seqsum_dpp_kernel<1>, dim3(seq_grid), dim3(64), 0, st, (const double*)b.prod, n, ncols, C, sc, i, tol);
    }
    GLX_HIP(hipGetLastError());
    hipLaunchKernelGGL((cg_pupdate_kernel<T>), dim3(pgrid + 1), blk, 0, st, (const T*)r, p, (const double*)sc.beta, n, L.ld, L.nvec,
                       sc, i, tol, 2);      // (+ the workgroup of the residual norms: cg_group_err_body)
    GLX_HIP(hipGetLastError());
    return GLX_OK;
  };
  // a chunk: the ring turns (row 0 <- the last row), then `cnt` iterations
  auto enqueue_chunk = [&](int cnt, bool bl0, bool bl1) -> int {
    hipLaunchKernelGGL(cg_roll_hist_kernel, dim3((unsigned)((stride + 255) / 256)), blk, 0, st, b.err_hist, stride, CG_CHUNK);
    GLX_HIP(hipGetLastError());
    for (int i = 1; i <= cnt; ++i) {
      int rc2 = enqueue_iteration(i, bl0, bl1);
      if (rc2) return rc2;
    }
    return GLX_OK;
  };
  // Full chunks are replayed from a captured launch sequence (one per combination of reduction forms): the same thirteen-odd launches
  // per iteration, without the host between them.  The sequences belong to the solve parameters they were captured for.
  const bool use_graphs = !(flags & GLX_CG_EAGER);
  if (use_graphs) {
    const std::vector<unsigned long long> key = {
        (unsigned long long)n, (unsigned long long)C, (unsigned long long)Cg, (unsigned long long)__builtin_bit_cast(unsigned long long, tol),
        (unsigned long long)dtype, (unsigned long long)flags, (unsigned long long)(uintptr_t)plan, (unsigned long long)(uintptr_t)b.x,
        (unsigned long long)(uintptr_t)b.r, (unsigned long long)(uintptr_t)b.p, (unsigned long long)(uintptr_t)b.ap,
        (unsigned long long)(uintptr_t)b.prod, (unsigned long long)(uintptr_t)b.part_dot, (unsigned long long)(uintptr_t)b.part_rs,
        (unsigned long long)(uintptr_t)b.scal, (unsigned long long)(uintptr_t)b.err_hist, (unsigned long long)(uintptr_t)b.ss_bsum,
        (unsigned long long)(uintptr_t)b.ss_csum, (unsigned long long)(uintptr_t)b.ss_rec, (unsigned long long)(uintptr_t)b.ss_mask,
        (unsigned long long)(uintptr_t)b.ss_stats, (unsigned long long)(uintptr_t)b.mask_rows, (unsigned long long)(uintptr_t)b.mask_ptr,
        (unsigned long long)(uintptr_t)a.rowmask, (unsigned long long)mask_grid, (unsigned long long)(uintptr_t)A->d_perm,
        (unsigned long long)(uintptr_t)b.pw_vals, (unsigned long long)b.pw_n, (unsigned long long)(uintptr_t)st};
    if (key != b.e_key) {
      for (int q = 0; q < 8; ++q)
        if (b.e_exec[q / 2][q % 2]) { hipGraphExecDestroy(b.e_exec[q / 2][q % 2]); b.e_exec[q / 2][q % 2] = nullptr; }
      b.e_key = key;
    }
  }
  for (int q = 0; q < 2; ++q)
    if (!b.e_ev[q]) GLX_HIP(hipEventCreateWithFlags(&b.e_ev[q], hipEventDisableTiming));
  // `inst`: the chunk's slot (0 / 1).  Two chunks are in flight at a time and each slot has its OWN instance of the captured sequence:
  // launching an instance that is still running makes the runtime wait for it on the host -- a sleeping wait, ~0.95 ms of idle device per
  // chunk in the first version of this loop (profiles/r06_cg_blocks_kernel_stats.csv has the trace before and after)
  auto launch_chunk = [&](int cnt, bool bl0, bool bl1, int inst) -> int {
    if (!use_graphs || cnt < CG_CHUNK) return enqueue_chunk(cnt, bl0, bl1);
    const int v = (bl0 ? 1 : 0) + (bl1 ? 2 : 0);
    if (!b.e_exec[v][inst]) {
      hipGraph_t graph = nullptr;
      GLX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc2 = enqueue_chunk(CG_CHUNK, bl0, bl1);
      const hipError_t e = hipStreamEndCapture(st, &graph);
      if (rc2) { if (graph) hipGraphDestroy(graph); return rc2; }
      GLX_HIP(e);
      const hipError_t e2 = hipGraphInstantiate(&b.e_exec[v][inst], graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      GLX_HIP(e2);
    }
    GLX_HIP(hipGraphLaunch(b.e_exec[v][inst], st));
    return GLX_OK;
  };
  {
  // The GPU does not wait for the host: chunk k + 1 is launched before the history of chunk k is looked at (the kernels of iterations
  // past convergence exit at once and leave "stopped" behind them); two chunks in flight, each with its own page-locked slot.
  struct Flight { int64_t it0; int cnt; int slot; };
  Flight fl[2];
  int nfl = 0, slot = 0;
  int64_t it = 0;                               // iterations launched
  const size_t slot_doubles = (size_t)(CG_CHUNK + 1) * stride;
  // A chunk launched ahead runs in the forms decided one look earlier.  While a kind of reduction in block form is anywhere near the price
  // of the chain (products that cancel: its walk can cost twice the chain), a late switch costs more than the bubble of waiting for the
  // look: one chunk in flight until the block form has shown itself cheap (under half the chain) or is no longer in use.
  bool ahead = !(ss_blocks && !blocks_forced);
  while (running > 0 && (it < max_iter || nfl > 0)) {
    if (it < max_iter && nfl < (ahead ? 2 : 1)) {
      const int cnt = (int)std::min<int64_t>(CG_CHUNK, max_iter - it);
      rc = launch_chunk(cnt, blocks_now[0], blocks_now[1], slot);
      if (rc) return rc;
      GLX_HIP(hipMemcpyAsync(b.h_err + slot * slot_doubles, b.err_hist + stride, (size_t)cnt * stride * 8, hipMemcpyDeviceToHost, st));
      if (ss_blocks) GLX_HIP(hipMemcpyAsync(b.h_ss + slot * 16, b.ss_stats, 128, hipMemcpyDeviceToHost, st));
      GLX_HIP(hipEventRecord(b.e_ev[slot], st));
      fl[nfl].it0 = it;
      fl[nfl].cnt = cnt;
      fl[nfl].slot = slot;
      ++nfl;
      it += cnt;
      slot ^= 1;
      if (nfl < (ahead ? 2 : 1) && it < max_iter) continue;
    }
    const Flight f = fl[0];
    fl[0] = fl[1];
    --nfl;
    GLX_HIP(hipEventSynchronize(b.e_ev[f.slot]));
    read_history(b.h_err + f.slot * slot_doubles, f.it0, f.cnt);
    if (ss_blocks && !blocks_forced) {
      const unsigned long long* hs = b.h_ss + f.slot * 16;
      bool cheap = true;
      for (int m = 0; m < 2; ++m) {
        if (!blocks_now[m]) continue;
        const double by_rec = (double)(hs[4 * m + 1] - ss_seen[m][1]), by_rows = (double)(hs[4 * m + 2] - ss_seen[m][2]);
        for (int q = 0; q < 3; ++q) ss_seen[m][q] = hs[4 * m + q];
        const double chains = (double)f.cnt * ncols;
        // measured on one MI355X (profiles/r06_cg_blocks_kernel_stats.csv): 0.7 us per block taken through its record, 1.5 per 256-row
        // block taken row by row (0.65 of additions -- the chain's 2.4 ns per row --, the rest its rows arriving: three are fetched
        // ahead per chunk; 399 us for a walk with every one of its 274 blocks row by row), 0.6 per chunk of the walk, 14.5 for the two
        // passes in front; the chain 2.4 ns per row
        const double blocks_us = (by_rec * 0.7 + by_rows * 1.5) / chains + ssw.nchunks * 0.6 + 14.5;
        const double chain_us = (double)n * 0.0024;
        if (blocks_us > chain_us) blocks_now[m] = false;
        else if (blocks_us > 0.5 * chain_us) cheap = false;
      }
      ahead = cheap;
    }
  }
  }
  if (rr.out_scale) {
    rc = glx_cg_unpack_scaled(dtype, b.x, b.dense, n, L, A->d_perm, b.out_scale, st);
    if (rc) return rc;
  } else {
    rc = glx_unpack_records(b.x, b.dense, n, L, dtype, st, A->d_perm);
    if (rc) return rc;
  }
  GLX_UP(glx_download(X, b.dense, (size_t)n * C * es, st, __func__));
  if (ss_blocks) GLX_HIP(hipMemcpyAsync(b.h_ss, b.ss_stats, 128, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  for (int q = 0; q < 3; ++q)     // (the entry point hands out ints: clamped, never negative -- -1 is the chain form's mark)
    b.ss_last[q] = ss_blocks ? (int)std::min<unsigned long long>(b.h_ss[q] + b.h_ss[4 + q] + b.h_ss[8 + q], 0x7fffffffull) : -1;
  b.ss_last[3] = ss_blocks ? (blocks_now[0] ? 1 : 0) + (blocks_now[1] ? 2 : 0) : -1;
  for (int g = 0; g < ngroups; ++g) {
    if (iters_out) iters_out[g] = (int)iters[g];
    if (err_out) err_out[g] = err[g];
  }
  return GLX_OK;
}

static int cg_entry(glx_graph* A, const void* B, void* X, int C, int group_cols, const int32_t* mask_rows, const int32_t* mask_ptr,
                    double tol, int64_t max_iter, int flags, int* iters_out, double* err_out, const CgRhsRows& rr) {
  GLX_CHECK(A && (B || rr.rows) && X, GLX_EINVAL, "glx_cg_multi: null argument");
  GLX_CHECK(A->n_rows == A->n_cols, GLX_EINVAL, "glx_cg_multi: operator must be square");
  GLX_CHECK(max_iter >= 0, GLX_EINVAL, "glx_cg_multi: negative max_iter");
  GLX_CHECK(C >= 1 && group_cols >= 1 && C % group_cols == 0, GLX_EINVAL,
            "glx_cg_groups: %d columns do not split into systems of %d columns", C, group_cols);
  GLX_CHECK((mask_rows == nullptr) == (mask_ptr == nullptr) || (mask_ptr && mask_ptr[C / group_cols] == 0), GLX_EINVAL,
            "glx_cg_groups_masked: mask_rows and mask_ptr go together");
  std::lock_guard<std::mutex> one_solve(A->solve_mu);   // the operator's work buffers are shared by its solves (include/glx.h: threading)
  GLX_HIP(hipSetDevice(A->device));
  return A->dtype == GLX_F32 ? cg_run<float>(A, B, X, C, group_cols, tol, max_iter, iters_out, err_out, flags, mask_rows, mask_ptr, rr)
                             : cg_run<double>(A, B, X, C, group_cols, tol, max_iter, iters_out, err_out, flags, mask_rows, mask_ptr, rr);
}

extern "C" int glx_cg_groups_masked(glx_graph* A, const void* B, void* X, int C, int group_cols, const int32_t* mask_rows,
                                    const int32_t* mask_ptr, double tol, int64_t max_iter, int flags, int* iters_out,
                                    double* err_out) {
  return cg_entry(A, B, X, C, group_cols, mask_rows, mask_ptr, tol, max_iter, flags, iters_out, err_out, CgRhsRows());
}

extern "C" int glx_cg_groups_rows(glx_graph* A, int64_t nb, const int32_t* b_rows, const void* b_vals, const double* out_scale, void* X,
                                  int C, int group_cols, const int32_t* mask_rows, const int32_t* mask_ptr, double tol, int64_t max_iter,
                                  int flags, int* iters_out, double* err_out) {
  GLX_CHECK(nb >= 0 && b_rows && (nb == 0 || b_vals), GLX_EINVAL, "glx_cg_groups_rows: null right-hand side");
  GLX_CHECK(!(flags & (GLX_CG_X0 | GLX_CG_NP1D)), GLX_EINVAL, "glx_cg_groups_rows: flags X0 / NP1D need the dense form");
  if (A) {
    GLX_CHECK(nb <= A->n_rows, GLX_EINVAL, "glx_cg_groups_rows: %lld rows given, the operator has %lld", (long long)nb, (long long)A->n_rows);
    for (int64_t q = 0; q < nb; ++q)
      GLX_CHECK(b_rows[q] >= 0 && b_rows[q] < A->n_rows, GLX_EINVAL, "glx_cg_groups_rows: row %d out of range", b_rows[q]);
  }
  CgRhsRows rr;
  rr.nb = nb;
  rr.rows = b_rows;
  rr.vals = b_vals;
  rr.out_scale = out_scale;
  return cg_entry(A, nullptr, X, C, group_cols, mask_rows, mask_ptr, tol, max_iter, flags, iters_out, err_out, rr);
}

// How the LAST reference-order solve on this operator took its reduction chains: blocks applied as plain integer sums / through
// their record (splits) / row by row, and which kinds of reduction were still in block form at its end (bit 0: p.Ap, bit 1: r.r);
// all -1: the chain form (one dependent addition per row) or no solve yet.
extern "C" int glx_cg_last_block_stats(glx_graph* A, int* out3) {
  GLX_CHECK(A && out3, GLX_EINVAL, "glx_cg_last_block_stats: null argument");
  for (int q = 0; q < 4; ++q) out3[q] = A->cg_ws ? ((CgBufs*)A->cg_ws)->ss_last[q] : -1;
  return GLX_OK;
}

extern "C" int glx_cg_solve(glx_graph* A, const void* B, void* X, int C, double tol, int64_t max_iter, int flags,
                            int* iters_out, double* err_out) {
  return glx_cg_groups_masked(A, B, X, C, C, nullptr, nullptr, tol, max_iter, flags, iters_out, err_out);
}

extern "C" int glx_cg_multi(glx_graph* A, const void* B, void* X, int C, double tol, int64_t max_iter, int* iters_out,
                            double* err_out) {
  return glx_cg_solve(A, B, X, C, tol, max_iter, 0, iters_out, err_out);
}
