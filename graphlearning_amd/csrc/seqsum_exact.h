// A sequential fp64 rounding chain, evaluated in parallel and still bit for bit.
//
// numpy's `np.sum(a*b, axis=0)` on a C-contiguous (n,k) array (utils.conjgrad, graphlearning/utils.py:524,527) is, per column,
//     s_0 = +0 ;  s_{i+1} = RN(s_i + x_i)
// -- n dependent roundings.  cg.hip's first reducer walked that chain at the latency of a dependent add (2.4 ns per row).  This
// header holds the arithmetic of the second one, which rests on one observation: while the running sum stays inside ONE binade,
// s = K * 2^q with 2^52 <= |K| < 2^53, a step is an INTEGER addition.  s + x = (K + X) 2^q with X = x / 2^q (an exact scaling),
// and as long as the result stays inside the binade the rounding grid is 2^q, so RN(s + x) = (K + rnd(X)) 2^q, where rnd is
// round-to-nearest and -- unless X lies exactly half way between two integers -- does not depend on K.  Integer addition is
// associative: a block of rows becomes (R, lo, hi) = (sum of rnd(X_i), minimum and maximum of the partial sums), computed by
// anyone who knows q; the block may be applied to an exact state K iff every intermediate K + P_j stays STRICTLY inside
// (2^52, 2^53) (then |K + X| stays inside [2^52, 2^53): |X - rnd(X)| <= 1/2).  What the integer form cannot express is handled
// by an exact fp64 addition of that one element (a "split"): a tie, a step that leaves the binade (each of the ~log2(n) times a
// growing sum doubles), a non-finite or oversized value.
// Who knows q in advance?  Nobody exactly; but an APPROXIMATE prefix (plain parallel block sums) tells the binade of the running
// sum at every block start unless it sits within rounding noise of a power of two, and predicts which element will cross.  The
// quantising pass works from that guess; the combining pass walks the blocks with the EXACT state and accepts a block only if
// the guess was right (same exponent, range test on lo / hi with the true K) -- otherwise it adds the block's rows one by one in
// fp64.  The result never depends on the guess; only the path taken does.
//
// Everything here is scalar and compiles for the host too (tests/test_seqsum_exact.py builds it with gcc and checks it against
// the plain chain on adversarial inputs); the kernels of cg_seqsum.hip call the same functions.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#ifndef GLX_HD
#define GLX_HD static inline
#endif
#if defined(__clang__)
#define SS_NOFMA _Pragma("clang fp contract(off)")     // every multiply and add below rounds on its own (the host build: -ffp-contract=off)
#else
#define SS_NOFMA
#endif

#ifndef SS_SUB
#define SS_SUB 16               // rows one thread turns into a record ...
#endif
#ifndef SS_Q
#define SS_Q 16                 // ... and records of consecutive rows merged into the record of one BLOCK (a power of two: the device merges in a tree)
#endif
#define SS_BLOCK (SS_SUB * SS_Q)   // rows per block: what the walk takes in one integer step when nothing special happens inside
#define SS_E_BAD (-1)           // a block the quantising pass could not prepare: always taken row by row
#define SS_E_ANY (-2)           // matches any exponent (blocks beyond the last row: nothing to add)
#define SS_EMIN 64              // states below 2^(64-1023) (and zero, subnormals, inf, nan) are not handled in integer form
#define SS_TWO52 4503599627370496LL
#define SS_TWO53 9007199254740992LL

GLX_HD int64_t ss_bits(double s) { int64_t b; memcpy(&b, &s, 8); return b; }
GLX_HD double ss_from_bits(int64_t b) { double s; memcpy(&s, &b, 8); return s; }
GLX_HD int ss_expo(double s) { return (int)((ss_bits(s) >> 52) & 0x7ff); }
// a state the integer form can carry: finite, normal, not tiny
GLX_HD bool ss_valid(double s) { const int E = ss_expo(s); return E >= SS_EMIN && E < 2047; }
// s = K * 2^(E - 1075), 2^52 <= |K| < 2^53
GLX_HD int64_t ss_mant(double s) {
  const int64_t b = ss_bits(s);
  const int64_t m = (b & 0xfffffffffffffLL) | SS_TWO52;
  return b < 0 ? -m : m;
}
GLX_HD double ss_compose(int E, int64_t K) {
  const int64_t a = K < 0 ? -K : K;   // 2^52 <= a < 2^53
  return ss_from_bits((K < 0 ? (int64_t)(1ULL << 63) : 0) | ((int64_t)E << 52) | (a & 0xfffffffffffffLL));
}
// 2^(1075 - E): x * scale = x / (the grid of a state with exponent E); E in [SS_EMIN, 2046] keeps it a normal double
GLX_HD double ss_scale(int E) { return ss_from_bits((int64_t)(1023 + 1075 - E) << 52); }

// rnd(x / grid) if it is an integer step the chain could take blindly: not a tie, not huge, not nan
GLX_HD bool ss_quant(double x, double scale, int64_t* r) {
  SS_NOFMA
  const double X = x * scale;                 // exact (a power of two) unless it underflows, and then |X| < 2^-1000: rnd = 0, no tie
  const double Rd = rint(X);                  // round half to even (the default mode; v_rndne_f64 on the device)
  // the integer in Rd, for |Rd| <= 2^51: Rd + 1.5 * 2^52 is exact and lies in [2^52, 2^53), where consecutive doubles are
  // consecutive bit patterns (a float -> int64 conversion is a dozen instructions on the device; this is two)
  *r = ss_bits(Rd + 6755399441055744.0) - 0x4338000000000000LL;
  // X - Rd is exact; a tie's direction depends on the parity of K.  (nan and inf fail the first test)
  return fabs(X) < 2251799813685248.0 && fabs(X - Rd) != 0.5;
}
// is a (a state in units of K's grid) strictly inside K's binade, on K's side of zero
GLX_HD bool ss_inside(int64_t K, int64_t a) {
  const int64_t m = K > 0 ? a : (int64_t)(0 - (uint64_t)a);
  return (uint64_t)m - (uint64_t)(SS_TWO52 + 1) < (uint64_t)(SS_TWO52 - 1);      // 2^52 < m < 2^53, one unsigned comparison
}
// may a segment with partial sums in [lo, hi] be applied to K?  (every intermediate strictly inside the binade, sign kept)
GLX_HD int64_t ss_wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }   // wrapping (a walk may test lanes it will not use)
GLX_HD bool ss_range_ok(int64_t K, int64_t lo, int64_t hi) { return ss_inside(K, ss_wadd(K, lo)) && ss_inside(K, ss_wadd(K, hi)); }

// What the quantising pass leaves per block: up to SS_MAXSPLIT + 1 integer segments with one row added exactly between each two.
#ifndef SS_MAXSPLIT
#define SS_MAXSPLIT 2
#endif
struct SsRec {
  int32_t E[SS_MAXSPLIT + 1];      // exponent each segment was quantised for (E[0] may be SS_E_BAD / SS_E_ANY)
  int32_t nsplit;                  // rows added in fp64: xs[0 .. nsplit), xs[j] between segments j and j + 1
  int64_t R[SS_MAXSPLIT + 1], lo[SS_MAXSPLIT + 1], hi[SS_MAXSPLIT + 1];
  double xs[SS_MAXSPLIT];
};

// (every index into a record is a compile-time constant after unrolling: on the device the record lives in registers)
GLX_HD void ss_store_seg(SsRec* rec, int seg, int E, int64_t R, int64_t lo, int64_t hi) {
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j)
    if (j == seg) { rec->E[j] = E; rec->R[j] = R; rec->lo[j] = lo; rec->hi[j] = hi; }
}

// SS_SUB rows prepared from the approximate state s_apx at the first of them: rows xv[0 .. len) (the caller has loaded them; on the
// device the loops unroll and xv lives in registers -- a row-by-row loop over memory would wait for every load in turn).
GLX_HD void ss_sub_record(const double* xv, int len, double s_apx, SsRec* rec) {
  SS_NOFMA
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) { rec->E[j] = SS_E_ANY; rec->R[j] = rec->lo[j] = rec->hi[j] = 0; }
#pragma unroll
  for (int j = 0; j < SS_MAXSPLIT; ++j) rec->xs[j] = 0.0;
  rec->nsplit = 0;
  {   // rows that are all +-0 change no state (+0 + -0 = +0, and the chain never holds -0): Dirichlet rows, leading zeros
    bool allzero = true;
#pragma unroll
    for (int i = 0; i < SS_SUB; ++i) allzero = allzero && (i >= len || xv[i] == 0.0);
    if (allzero) return;
  }
  rec->E[0] = SS_E_BAD;
  if (!ss_valid(s_apx)) return;
  int E = ss_expo(s_apx);
  double scale = ss_scale(E);
  int64_t Kt = ss_mant(s_apx);          // predicted state: decides where to split, never what the sum is
  int64_t R = 0, lo = 0, hi = 0;
  int seg = 0;
  bool dead = false;                    // more rows to add exactly than the record holds, or a state the integer form cannot carry
#pragma unroll
  for (int i = 0; i < SS_SUB; ++i) {  // (no early exits: the loop unrolls completely and xv stays in registers)
    if (i < len && !dead) {
      const double xi = xv[i];
      int64_t r = 0;
      const bool q = ss_quant(xi, scale, &r);
      const int64_t Rn = ss_wadd(R, r);
      if (q & ss_inside(Kt, ss_wadd(Kt, Rn))) {        // (&: both sides are cheap and branch-free)
        R = Rn;
        lo = R < lo ? R : lo;
        hi = R > hi ? R : hi;
      } else if (seg == SS_MAXSPLIT) {
        dead = true;                    // one too many: the block goes row by row
      } else {                          // this row is to be added exactly
        ss_store_seg(rec, seg, E, R, lo, hi);
#pragma unroll
        for (int j = 0; j < SS_MAXSPLIT; ++j)
          if (j == seg) rec->xs[j] = xi;
        const double sn = ss_compose(E, Kt + R) + xi;
        if (!ss_valid(sn)) {
          dead = true;
        } else {
          E = ss_expo(sn);
          scale = ss_scale(E);
          Kt = ss_mant(sn);
          R = 0; lo = 0; hi = 0;
          ++seg;
        }
      }
    }
  }
  if (dead) { rec->E[0] = SS_E_BAD; return; }
  ss_store_seg(rec, seg, E, R, lo, hi);
  rec->nsplit = seg;
}
// Record r (rows that follow M's) appended to M: the last segment of M continues with segment 0 of r when both were quantised
// for the same exponent; r's further segments follow.  Whatever does not fit (exponents that disagree, more splits than a record
// holds) leaves M "row by row".  A record is only ever a PROPOSAL -- ss_apply_record checks every segment against the exact state --
// so any way of merging is safe; this one keeps the partial-sum extremes exact.
GLX_HD void ss_merge_record(SsRec* M, const SsRec* r) {
  if (M->E[0] == SS_E_BAD || r->E[0] == SS_E_ANY) return;
  if (r->E[0] == SS_E_BAD) { M->E[0] = SS_E_BAD; return; }
  if (M->E[0] == SS_E_ANY) { *M = *r; return; }
  const int m = M->nsplit;
  int El = 0;
  int64_t Rl = 0, lol = 0, hil = 0;
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j)
    if (j == m) { El = M->E[j]; Rl = M->R[j]; lol = M->lo[j]; hil = M->hi[j]; }
  if (El != r->E[0] || m + r->nsplit > SS_MAXSPLIT) { M->E[0] = SS_E_BAD; return; }
  const int64_t a = ss_wadd(Rl, r->lo[0]), b = ss_wadd(Rl, r->hi[0]);
  ss_store_seg(M, m, El, ss_wadd(Rl, r->R[0]), a < lol ? a : lol, b > hil ? b : hil);
#pragma unroll
  for (int j = 1; j <= SS_MAXSPLIT; ++j) {
    if (j <= r->nsplit) {
      ss_store_seg(M, m + j, r->E[j], r->R[j], r->lo[j], r->hi[j]);
#pragma unroll
      for (int k = 0; k < SS_MAXSPLIT; ++k)
        if (k == m + j - 1) M->xs[k] = r->xs[j - 1];
    }
  }
  M->nsplit = m + r->nsplit;
}

// One block from memory (x[i * stride], `len` rows, s_apx[q] = the approximate state in front of its q-th run of SS_SUB rows),
// merged in the tree order of the device (host tests)
GLX_HD void ss_block_record(const double* x, int64_t stride, int len, const double* s_apx, SsRec* rec) {
  SsRec sub[SS_Q];
  for (int q = 0; q < SS_Q; ++q) {
    double xv[SS_SUB];
    const int l = len - q * SS_SUB < 0 ? 0 : (len - q * SS_SUB > SS_SUB ? SS_SUB : len - q * SS_SUB);
    for (int i = 0; i < SS_SUB; ++i) xv[i] = i < l ? x[(int64_t)(q * SS_SUB + i) * stride] : 0.0;
    ss_sub_record(xv, l, s_apx[q], &sub[q]);
  }
  for (int d = 1; d < SS_Q; d *= 2)
    for (int q = 0; q + d < SS_Q; q += 2 * d) ss_merge_record(&sub[q], &sub[q + d]);
  *rec = sub[0];
}

// Apply a prepared block to the exact state.  false: the guess did not hold (state untouched) -- add the rows one by one.
GLX_HD bool ss_apply_record(double* s, const SsRec* rec) {
  SS_NOFMA
  if (rec->E[0] == SS_E_ANY) return true;
  if (!ss_valid(*s)) return false;
  int E = ss_expo(*s);
  int64_t K = ss_mant(*s);
#pragma unroll
  for (int j = 0; j <= SS_MAXSPLIT; ++j) {
    if (j > rec->nsplit) break;
    if (rec->E[j] != E || !ss_range_ok(K, rec->lo[j], rec->hi[j])) return false;
    K += rec->R[j];
    if (j < SS_MAXSPLIT && j < rec->nsplit) {
      const double sn = ss_compose(E, K) + rec->xs[j < SS_MAXSPLIT ? j : 0];
      if (!ss_valid(sn)) return false;
      E = ss_expo(sn);
      K = ss_mant(sn);
    }
  }
  *s = ss_compose(E, K);
  return true;
}
