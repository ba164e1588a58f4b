/* csr_ref.c -- plain-C restatement of the two scipy product loops the reference's hot loops
 * execute (TEST INFRASTRUCTURE ONLY; see oracle/gl_oracle.py for the rules of use).
 *
 * The arithmetic of `u = Db + P*u` (reference graphlearning/ssl.py:668), `A@p`
 * (graphlearning/utils.py:515,523) and `v = RW*v` (ssl.py:669) lives in an un-vendored
 * third-party dependency: scipy 1.15.3, scipy/sparse/sparsetools/csr.h `csr_matvecs` and
 * csc.h `csc_matvec`.  Their published algorithm:
 *   csr_matvecs: for each row i, for each stored entry jj of the row in stored order:
 *                y[i, :] += a[jj] * x[col[jj], :]        (axpy over the n_vecs columns)
 *   csc_matvec:  for each column j, for each stored entry ii: y[row[ii]] += a[ii] * x[j]
 * Products and sums are separately rounded (baseline x86-64 builds have no FMA contraction).
 * tests/test_oracle_c.py checks these loops bit-for-bit against scipy on random operators.
 *
 * Build: make -C oracle   ->  oracle/_build/libcsr_ref.so   (gcc -O2 -ffp-contract=off)
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Y (n_row, n_vecs) += A (CSR) * X (n_col, n_vecs); C-contiguous dense operands */
void ref_csr_matvecs(int64_t n_row, int64_t n_vecs, const int32_t* indptr, const int32_t* indices, const double* data,
                     const double* X, double* Y) {
  for (int64_t i = 0; i < n_row; ++i) {
    double* y = Y + (size_t)i * n_vecs;
    for (int32_t jj = indptr[i]; jj < indptr[i + 1]; ++jj) {
      const double a = data[jj];
      const double* x = X + (size_t)indices[jj] * n_vecs;
      for (int64_t c = 0; c < n_vecs; ++c) y[c] += a * x[c];
    }
  }
}

/* y (n_row) += A (CSC) * x (n_col) */
void ref_csc_matvec(int64_t n_col, const int32_t* indptr, const int32_t* indices, const double* data, const double* x,
                    double* y) {
  for (int64_t j = 0; j < n_col; ++j)
    for (int32_t ii = indptr[j]; ii < indptr[j + 1]; ++ii) y[indices[ii]] += data[ii] * x[j];
}

/* T Poisson sweeps u <- Db + P u (ssl.py:668), ping-pong between u and tmp; result in u */
void ref_poisson_sweeps(int64_t n, int64_t C, const int32_t* indptr, const int32_t* indices, const double* data,
                        const double* Db, double* u, double* tmp, int64_t T) {
  for (int64_t t = 0; t < T; ++t) {
    for (int64_t k = 0; k < n * C; ++k) tmp[k] = 0.0;
    ref_csr_matvecs(n, C, indptr, indices, data, u, tmp);
    for (int64_t k = 0; k < n * C; ++k) u[k] = Db[k] + tmp[k];
  }
}

/* The same sweeps with the rows of every sweep spread over OpenMP threads (a many-core CPU
 * figure beside the single-threaded scipy one).  A row's entries are still added one after another in
 * stored order, so the result is bit-identical to ref_poisson_sweeps; the fused stop column
 * w <- P w rides along like on the GPU.  Returns the number of threads used. */
int ref_poisson_sweeps_omp(int64_t n, int64_t C, const int32_t* indptr, const int32_t* indices, const double* data,
                           const double* Db, double* u, double* tmp, double* w, double* wtmp, int64_t T, int nthreads) {
  int threads = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    threads = omp_get_num_threads();
  }
#endif
  for (int64_t t = 0; t < T; ++t) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      double acc[64];                       /* C <= 64 (checked by the caller) */
      for (int64_t v = 0; v < C; ++v) acc[v] = 0.0;
      double yw = 0.0;
      const double* restrict uu = u;
      const double* restrict ww = w;
      for (int32_t jj = indptr[i]; jj < indptr[i + 1]; ++jj) {
        const int32_t j = indices[jj];
        const double a = data[jj];
        const double* restrict x = uu + (int64_t)j * C;
        for (int64_t v = 0; v < C; ++v) acc[v] += a * x[v];
        yw += a * ww[j];
      }
      double* restrict y = tmp + i * C;
      for (int64_t v = 0; v < C; ++v) y[v] = Db[i * C + v] + acc[v];
      wtmp[i] = yw;
    }
    double* s;
    s = u; u = tmp; tmp = s;
    s = w; w = wtmp; wtmp = s;
  }
  return threads;   /* after an odd T the newest iterate is in the caller's tmp / wtmp */
}

/* ---- p-Laplace Jacobi sweep ---------------------------------------------------
 * Restates lp_iterate_main, reference c_code/lp_iterate.cpp:35-125, as called by graph.plaplace
 * with fast=False (graph.py:1262-1276: `cextensions.lp_iterate(uu,ul,self.J,self.I,self.V,...)`,
 * i.e. nbr = the neighbour of each stored entry, row = its vertex, entries sorted by vertex).
 * Kept quirks: err is the maximum of uu-ul over the iterate that was READ; the test
 * `err < tol && it > 10` comes before the pointer swap; the swap exchanges local pointers only, so
 * on return the caller's arrays hold the iterate that last lived in them (U_it if the loop stopped
 * at an even `it`, U_it+1 if odd; after T full iterations U_T if T is even, U_T-1 if odd).
 * Returns the iteration index at which the loop stopped (T if it ran out). */
#define REF_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define REF_MAX(a, b) (((a) > (b)) ? (a) : (b))
int64_t ref_lp_iterate(double* uu_caller, double* ul_caller, const int32_t* nbr, const int32_t* row, const double* W,
                       const int32_t* ind, const double* val, double p, int64_t T, double tol, int64_t n, int64_t M, int64_t m) {
  const double alpha = 1 / p;
  const double delta = 1 - 2 / p;
  double dt = 0.9 / (alpha + 2 * delta);
  int64_t* num = (int64_t*)calloc((size_t)n, sizeof(int64_t));
  int64_t* start = (int64_t*)calloc((size_t)n, sizeof(int64_t));
  double* invdeg = (double*)calloc((size_t)n, sizeof(double));
  double* vu = (double*)calloc((size_t)n, sizeof(double));
  double* vl = (double*)calloc((size_t)n, sizeof(double));
  double *uu = uu_caller, *ul = ul_caller;
  int64_t i, j = 0, it;
  for (i = 0; i < n; i++) {                        /* lp_iterate.cpp:47-58 */
    start[i] = j;
    invdeg[i] = 0;
    while (j < M && row[j] == i) {
      num[i]++;
      invdeg[i] += W[j];
      j++;
    }
    invdeg[i] = alpha / invdeg[i];
  }
  double maxW = 0;                                  /* :61-64 */
  for (i = 0; i < M; i++) maxW = REF_MAX(maxW, W[i]);
  dt = dt / maxW;
  for (it = 0; it < T; it++) {                      /* :74-124 */
    double err = 0;
    for (i = 0; i < n; i++) {
      double minw = 0, maxw = 0, sumw = 0;
      for (j = start[i]; j < start[i] + num[i]; j++) {
        minw = REF_MIN(W[j] * (uu[nbr[j]] - uu[i]), minw);
        maxw = REF_MAX(W[j] * (uu[nbr[j]] - uu[i]), maxw);
        sumw += W[j] * (uu[nbr[j]] - uu[i]);
      }
      vu[i] = uu[i] + dt * (invdeg[i] * sumw + delta * (minw + maxw));
      minw = 0; maxw = 0; sumw = 0;
      for (j = start[i]; j < start[i] + num[i]; j++) {
        minw = REF_MIN(W[j] * (ul[nbr[j]] - ul[i]), minw);
        maxw = REF_MAX(W[j] * (ul[nbr[j]] - ul[i]), maxw);
        sumw += W[j] * (ul[nbr[j]] - ul[i]);
      }
      vl[i] = ul[i] + dt * (invdeg[i] * sumw + delta * (minw + maxw));
      err = REF_MAX(uu[i] - ul[i], err);
    }
    for (j = 0; j < m; j++) {
      i = ind[j];
      vu[i] = val[j];
      vl[i] = val[j];
    }
    if (err < tol && it > 10) break;
    double* t;
    t = uu; uu = vu; vu = t;
    t = ul; ul = vl; vl = t;
  }
  /* the two scratch arrays are whichever pair the caller does not own */
  free(uu == uu_caller ? vu : uu);
  free(ul == ul_caller ? vl : ul);
  free(num); free(start); free(invdeg);
  return it;
}
