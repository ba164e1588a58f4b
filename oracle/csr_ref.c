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
