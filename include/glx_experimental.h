/* glx_experimental.h -- the entry points of libglx.so that are NOT part of the drop-in boundary (include/glx.h): host helpers of the
 * Python boundary, diagnostics and plan overrides for tests and A/B measurements, the device-pointer calls of the torch fallback
 * engine of the multi-GPU path, and the stepwise form of the distributed sweep with which multi-rank jobs are tested on ONE GPU.
 * Same conventions as glx.h (status codes, glx_last_error, borrowed host pointers); no stability promise. */
#ifndef GLX_EXPERIMENTAL_H
#define GLX_EXPERIMENTAL_H
#include "glx.h"
#ifdef __cplusplus
extern "C" {
#endif

int glx_version(void);
int glx_device_synchronize(void);
/* sweep kernels enqueued so far by a prepared sweep (the bench derives a per-launch time from it) */
int glx_sweep_launches(const glx_sweep* s, int64_t* sweep_kernel_launches);
int glx_sweep_groups_launches(const glx_sweep_groups* s, int64_t* sweep_kernel_launches);

/* Host helpers of ssl.poisson's operator set-up (graphlearning/ssl.py:634-635, 642) for a W that is symmetric bit for bit:
 * row sums in stored order (= scipy's W * ones), and the rows of P = D^-1 W^T written down without a transpose -- row i of W
 * scaled by scale[i] with its entries in reverse order, the arrays scipy's `D * W.transpose()` yields.  No device involved. */
int glx_host_row_sums(int64_t n, const int32_t* rowptr, const double* val, double* sum_out);
int glx_host_reverse_scale_rows(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val,
                                const double* scale, int32_t* col_out, double* val_out);
/* Host: the nonzero rows of -L[:, cols] * F from the CSC image (cptr, crow, cval) of a canonical L -- ssl.laplace's right-hand side,
 * reference ssl.py:1236, a row's terms added in ascending column order from 0 as scipy's csr_matvecs does --, without the rows listed
 * in cols, every row times row_scale[row] when given (M*b, ssl.py:1249).  rows_out ascending, vals_out (count, k); cap = room in both.
 * *count_out = -1 (and GLX_OK): cols has duplicates, use the literal expression. */
int glx_host_neg_columns_rows(int64_t n, const int32_t* cptr, const int32_t* crow, const double* cval, int64_t m, const int64_t* cols,
                              const double* F, int k, const double* row_scale, int64_t cap, int32_t* rows_out, double* vals_out,
                              int64_t* count_out);

/* host helpers of the sharded build: the library's locality order (perm_out[new] = old, the breadth-first pass glx_graph uses for
 * square operators) of an n-row pattern restricted to the columns [col_lo, col_lo + n) -- a rank orders its own rows by their links
 * among themselves --, and the rows of a CSR matrix in another order (row i of the result = row perm[i], entry order kept). */
int glx_host_locality_order(int64_t n, const int32_t* rowptr, const int32_t* col, int64_t col_lo, int32_t* perm_out);
int glx_host_permute_rows(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, const int64_t* perm,
                          int32_t* rowptr_out, int32_t* col_out, double* val_out);

/* 128-bit content fingerprint of `bytes` bytes (chunks hashed on a few host threads, then combined): what the learners key
 * their device-resident operators by, so that a weight matrix edited in place between two fits is seen as a new graph. */
int glx_host_fingerprint(const void* data, size_t bytes, uint64_t seed, uint64_t out[2]);

/* info[0]=n_rows,[1]=n_cols,[2]=nnz,[3]=stored entries incl. padding,[4]=slices,[5]=rows per slice,[6]=max row nnz,[7]=1 if renumbered */
int glx_graph_info(const glx_graph* g, int64_t info[8]);
/* the internal vertex order: perm_out[new] = caller's row (n_rows entries; the identity when the operator
 * was not renumbered).  Forces the order to be computed if it has not been yet. */
int glx_graph_order(glx_graph* g, int32_t* perm_out);

/* ---- device-pointer entry points: rank-local sweeps of the vertex-partitioned solver -------
 * Buffers are DEVICE memory in the vertex-record layout (torch tensors' data_ptr()); `stream`
 * is a hipStream_t; nothing synchronises.  Record layout: `ld` elements per vertex, columns
 * 0..C-1, zero padding to a multiple of 4, then (has_w) the fp64 stop value at byte `woff`.
 * out = {ld, woff, record bytes, lanes per row, 4-wide column vectors, element size}. */
int glx_record_layout(int C, int dtype, int has_w, int32_t out[6]);
int glx_graph_slots(glx_graph* P, int C, int has_w, int64_t* nslots);
/* flags[slot] = 1 where the slot's row has a nonzero bias record (sparse Db: ssl.py:620-622) */
int glx_bias_flags_dev(glx_graph* P, int C, int has_w, const void* bias_rec, uint8_t* flags, void* stream);
/* one sweep xout[0:n_rows] = bias + P xin[0:n_cols]; err_next (64 x uint64, caller-zeroed) receives
 * max |deg*w - vinf| as fp64 bit patterns when non-NULL (the rank-local part of ssl.py:667) */
int glx_sweep_step_dev(glx_graph* P, int C, int has_w, const void* xin, void* xout, const void* bias_rec,
                       const uint8_t* slot_flags, const double* deg, const double* vinf, void* err_next,
                       void* stream);
int glx_pack_records_dev(const void* dense, void* rec, int64_t n, int C, int dtype, int has_w, const double* w,
                         void* stream);
int glx_unpack_records_dev(const void* rec, void* dense, int64_t n, int C, int dtype, int has_w, void* stream);

/* Dense vector kernels of utils.conjgrad (graphlearning/utils.py:483-532) on device records, for the vertex-partitioned
 * conjugate-gradient solve (dist.py: cg_distributed): out[c] = sum_i a[i,c] b[i,c] in fp64 with a fixed order inside the
 * rank (partial: device scratch of glx_rec_dots_scratch(n, C) doubles); x += alpha p, r -= alpha Ap; p = r + beta p --
 * alpha, beta: device fp64[C].  The ranks add their column sums with one all-reduce each (tolerance mode). */
int64_t glx_rec_dots_scratch(int64_t n, int C);
int glx_rec_dots_dev(const void* a, const void* b, int64_t n, int C, int dtype, int has_w, double* partial, double* out, void* stream);
int glx_rec_axpy2_dev(void* x, void* r, const void* p, const void* Ap, const double* alpha, int64_t n, int C, int dtype, int has_w,
                      void* stream);
int glx_rec_xpby_dev(void* p, const void* r, const double* beta, int64_t n, int C, int dtype, int has_w, void* stream);

int glx_dist_comm_info(const glx_comm* c, int32_t info[4]);   /* rank, nranks, device, 1 if it has an RCCL communicator */

int glx_dist_sweep_stats(const glx_dist_sweep* s, int64_t out[4]);      /* sweeps run, exchanges enqueued, graphs, 1 if it exchanges */
/* what the object decided: out[0] 1 if it exchanges, [1] exchanging sweeps captured (1) / eager (0) / undecided (-1),
 * [2] self-test 0 not run / 1 passed / 2 failed, [3] exchange on a second stream beside the interior rows, [4] one
 * launch per sweep (GLX_DIST_FUSE), [5] boundary rows scattered into the send buffer by the SpMM (no pack kernel),
 * [6] records sent per sweep, [7] halo records */
int glx_dist_sweep_info(const glx_dist_sweep* s, int64_t out[8]);
/* device microseconds of the rank-local pieces of a sweep, each timed alone over `reps` launches: [0] boundary rows
 * (incl. the scatter), [1] interior rows, [2] the stand-alone pack kernel, [3] boundary + interior back to back */
int glx_dist_sweep_time_parts(glx_dist_sweep* s, int reps, float us_out[4]);

/* the same pieces one at a time with the transport left to the caller (eager, synchronous): multi-rank tests on one
 * GPU move the packed records between ranks through a host-side backend */
int glx_dist_sweep_begin(glx_dist_sweep* s);                                    /* state <- initial records; packs them */
int glx_dist_sweep_boundary(glx_dist_sweep* s, int want_err);                   /* boundary rows of the next iterate; packs them */
int glx_dist_sweep_get_send(glx_dist_sweep* s, void* host_out);                 /* the packed records (sum of send_counts) */
int glx_dist_sweep_put_halo(glx_dist_sweep* s, const void* host_in, int next);  /* received records -> halo of the current / next iterate */
int glx_dist_sweep_interior(glx_dist_sweep* s, int want_err, double* err_local_out);   /* interior rows; next becomes current */

/* all n rows in the caller's order, the cells formed by the library: ncells (<= 4096; 0 / 1 = plain all-pairs search) evenly
 * spaced rows serve as centres, every row joins the nearest, the rows are reordered by cell on the device and searched with the
 * pruning of glx_knn_cells_range.  Indices and output rows are the caller's, equal distances go to the lower caller index: the
 * lists of glx_knn_bruteforce bit for bit.  ncells < -1: the rows reordered by -ncells chained cells on the device, then ALL PAIRS
 * (no pruning: a wavefront's queries share a corner of feature space, which is worth 10-14 % of the search on clustered data
 * below the size where pruning pays, and nothing elsewhere).  glx_knn_search hands that order out with its result. */
int glx_knn_clustered(const double* X, int64_t n, int d, int k, int ncells, int64_t* ind_out, double* dist_out, int device);
/* Plan overrides of the calling thread's searches (tests and A/B measurements; NULL or all-default values = the library decides):
 * filter 0 auto | 1 split-bf16 | 2 fp32 operands; lists 0 auto | 1 short | 2 long (one list holds all k neighbours of a query);
 * nsplit 0 auto | 1..8 ref ranges per query block; concat -1 auto | 0 blocks of 16 features | 1 concatenated split operands
 * (d <= 21) | 2 with the norm folded in (d <= 20).  Every plan returns the same exact lists. */
typedef struct { int filter, lists, nsplit, concat; } glx_knn_options;
int glx_knn_set_options(const glx_knn_options* opt);

/* 0: the device work-buffer pool and the idle work sets (streams, events, staging) are bypassed -- every buffer comes from hipMalloc and
 * goes back with hipFree; 1 (default): size-class free lists in front of the runtime.  Results are identical either way
 * (tests/test_gpu_switches.py); the switch exists for ablation runs of the randomised soak. */
int glx_pool_set_enabled(int enabled);
/* debugging aid: every work buffer the pool hands out is first filled with `byte` (0 .. 255; -1 = off, the default), so that a kernel reading
 * a buffer before anything wrote it computes from the pattern instead of from what an earlier call left there */
int glx_pool_set_poison(int byte);
/* debugging aid of the search: flags bit 0 = after the upload of a search's features read the device copy back (by the copy engine and through a
 * kernel) and compare it with the caller's array; counters: uploads checked, uploads whose engine / kernel read-back differed, bytes differing */
int glx_debug_set(int flags);
/* how uploads of 128 KB or more travel: 0 (default) through the library's page-locked staging area and CHECKED (word sums of the source and of
 * what arrived; a difference is described on stderr and the upload repeated), 1 staged without the check, 2 hipMemcpyAsync straight from the caller's
 * pageable memory as rounds 1-5 did -- the path on which round 6 found holes of 256 zero bytes.  For A/B runs. */
int glx_upload_set_mode(int mode);
/* the checked uploads of this process (csrc/glx_internal.h glx_upload): uploads checked, sums that differed, uploads that arrived intact on
 * a repeat, uploads given up */
int glx_upload_stats(unsigned long long out[4]);
int glx_debug_counters(unsigned long long out[4]);

/* out[i] = exp(x[i]) correctly rounded (csrc/exp_cr.h: the exponential of the Gaussian weights), host arrays; a test hook */
int glx_exp_cr(const double* x, double* out, int64_t n, int device);

int glx_knn_stats(double stats[16]);  /* of the calling thread's last search: [0] tile-kernel ms, [1] re-rank ms, [2] fallback rows, [3] total device ms,
                                        [4] fallback ms, [5] padded feature count, [6] ref ranges, [7] list length (negative: bf16 filter),
                                        [8] rows the short lists could not accept when the search was repeated with long ones (else 0);
                                        [9] concatenated operands (d <= 21): 0 no, 1 yes, 2 with the norm folded in (d <= 20),
                                        [10] tile stride of the sample the seeding pre-pass looked at (0: no pre-pass; tile-kernel ms
                                        include it and the cell passes), [11] share of the (query block, ref tile) pairs visited and
                                        [12] number of cells of a cell-pruned search (0: all pairs); [13..15] reserved */

/* the smallest relative distance from `tol` of the residual norms that decided the stops of the last tolerance-mode (GLX_CG_TREE)
 * solve on this operator; +inf when there was none.  ssl.laplace / ssl.randomwalk (reduce='auto') hand a solve whose stop hung on less
 * than ssl.AUTO_STOP_BAND back to the reference-order reductions. */
int glx_cg_last_stop_margin(glx_graph* A, double* margin_out);
/* how the last reference-order solve on this operator walked numpy's reduction chains (csrc/seqsum_exact.h): out4 = blocks of 256 rows
 * applied as plain integer sums, through their record (rows added exactly between integer segments), row by row -- summed over
 * the solve's reductions --, and which kinds of reduction were still in block form at its end (bit 0: p.Ap, bit 1: r.r; the solve
 * moves a kind whose products cancel to the chain form, see GLX_CG_BLOCKS); all -1: the chain form throughout (GLX_CG_CHAIN, fewer
 * than 8192 rows, a 1-D right-hand side) or no solve yet. */
int glx_cg_last_block_stats(glx_graph* A, int* out4);

#ifdef __cplusplus
}
#endif
#endif
