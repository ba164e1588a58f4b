// exp(x) for a double x, correctly rounded (round to nearest) over the normal range -- the one operation of weightmatrix.knn
// whose bits are library-defined in the reference (`np.exp`, graphlearning/weightmatrix.py:144-150: glibc's exp on some hosts,
// numpy's own SIMD kernel on others, neither correctly rounded).  With the exact result rounded once, the device weights do not
// depend on the machine, and where they differ from a host's numpy the difference is that host's libm error.
//
// Method: x = N * (ln2/64) + r, N = 64 k + j, |r| <= ln2/128; exp(x) = 2^k * 2^(j/64) * exp(r).  r is formed in double-double
// (ln2/64 in three pieces, the first two products exact), 2^(j/64) comes from a double-double table, exp(r) from the Taylor
// series to r^13/13! in double-double Horner form (truncation < 2^-120, arithmetic ~2^-100 relative).  The rounded high word of
// the product is the correctly rounded result unless the exact value lies within ~2^-100 relative of a rounding boundary -- for
// uniformly spread arguments a chance of ~2^-47 per call.  Results below 2^-1022 are scaled in one step (a second rounding: up
// to 1 ulp of the subnormal there).  Plain C: compiles for the device (assemble.hip) and for the host (tests/test_exp_cr.py).
#pragma once
#include "exp_cr_tables.h"
#include <math.h>

// Every function below must be compiled WITHOUT floating-point contraction: hipcc's default (-ffp-contract=fast) fuses the rounded
// product a.hi*b.hi of expcr_mul into the addition that follows it and counts its rounding error twice -- the result is then
// an ordinary ~0.5-ulp exp (22 % of the values one ulp off; caught by tests/test_exp_cr.py on the device).
#if defined(__clang__)
#define EXPCR_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define EXPCR_NOCONTRACT
#endif
#if defined(__HIPCC__)
#define EXPCR_FN __host__ __device__ static inline
#define EXPCR_CONST static __device__ __constant__ const
#else
#define EXPCR_FN static inline
#define EXPCR_CONST static const
#endif

typedef struct { double hi, lo; } expcr_dd;
#if defined(__HIPCC__)
static __device__ const expcr_dd expcr_table_dev[64] = EXPCR_TABLE;
static __device__ const expcr_dd expcr_poly_dev[12] = EXPCR_POLY;
#endif
static const expcr_dd expcr_table_host[64] = EXPCR_TABLE;
static const expcr_dd expcr_poly_host[12] = EXPCR_POLY;

EXPCR_FN expcr_dd expcr_two_sum(double a, double b) {
  EXPCR_NOCONTRACT
  expcr_dd r;
  r.hi = a + b;
  const double bb = r.hi - a;
  r.lo = (a - (r.hi - bb)) + (b - bb);
  return r;
}
EXPCR_FN expcr_dd expcr_quick_two_sum(double a, double b) {   // |a| >= |b|
  EXPCR_NOCONTRACT
  expcr_dd r;
  r.hi = a + b;
  r.lo = b - (r.hi - a);
  return r;
}
EXPCR_FN expcr_dd expcr_add(expcr_dd a, expcr_dd b) {
  EXPCR_NOCONTRACT
  expcr_dd s = expcr_two_sum(a.hi, b.hi);
  const expcr_dd t = expcr_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = expcr_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return expcr_quick_two_sum(s.hi, s.lo);
}
EXPCR_FN expcr_dd expcr_mul(expcr_dd a, expcr_dd b) {
  EXPCR_NOCONTRACT
  expcr_dd p;
  p.hi = a.hi * b.hi;
  p.lo = fma(a.hi, b.hi, -p.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return expcr_quick_two_sum(p.hi, p.lo);
}

EXPCR_FN double exp_cr(double x) {
  EXPCR_NOCONTRACT
  if (x != x) return x;
  if (x > 709.782712893384) return INFINITY;
  if (x < -745.2) return 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
  const expcr_dd* table = expcr_table_dev;
  const expcr_dd* poly = expcr_poly_dev;
#else
  const expcr_dd* table = expcr_table_host;
  const expcr_dd* poly = expcr_poly_host;
#endif
  const double Nf = nearbyint(x * EXPCR_INV);
  const long long N = (long long)Nf;
  const int j = (int)(N & 63);
  const int k = (int)((N - j) / 64);
  // r = x - N ln2/64 in double-double: N*C1 and N*C2 are exact (32-bit pieces, |N| < 2^17), x - N*C1 is exact (the two agree to
  // within ln2/128 + rounding of the quotient)
  const double t1 = x - Nf * EXPCR_C1;
  expcr_dd r = expcr_two_sum(t1, -(Nf * EXPCR_C2));
  r.lo -= Nf * EXPCR_C3;
  r = expcr_quick_two_sum(r.hi, r.lo);
  // exp(r) = 1 + r + r^2 (1/2! + r (1/3! + ... + r/13!))
  expcr_dd p = poly[11];
  for (int q = 10; q >= 0; --q) p = expcr_add(expcr_mul(p, r), poly[q]);
  p = expcr_mul(expcr_mul(p, r), r);
  p = expcr_add(p, r);
  expcr_dd one;
  one.hi = 1.0;
  one.lo = 0.0;
  p = expcr_add(p, one);
  p = expcr_mul(p, table[j]);
  return ldexp(p.hi, k);
}
