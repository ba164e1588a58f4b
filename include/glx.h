/* glx.h -- C-ABI of libglx.so: the MI355X (gfx950) device path for GraphLearning's
 * kNN-graph + Poisson/Laplace label-propagation hot path.
 *
 * The reference (jwcalder/GraphLearning v1.7.5) has no FFI on this path; its only
 * device seam is the `if self.use_cuda:` blocks of graphlearning/ssl.py:649-663 and
 * :807-823 (host builds scipy CSR P and dense Db, the device runs the iteration, the
 * result is copied back as numpy).  Its native calling convention elsewhere
 * (c_code/cextensions.cpp:19-60, graph.py:69-84) is "caller pre-allocates C-contiguous
 * numpy buffers with explicit dtypes, callee fills them in place".  libglx keeps that
 * convention and adds status codes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative GLX_E* code on failure;
 *     glx_last_error() returns a thread-local message for the last failure.
 *   - host pointers are borrowed for the duration of the call; outputs are
 *     caller-allocated unless stated (glx_knn_to_csr allocates, glx_free releases).
 *   - dense matrices are C-contiguous (n, C); dtype codes: GLX_F32 = 0, GLX_F64 = 1.
 *   - no global HIP state is created at load time (fork-safe; ssl.py:390-396 forks
 *     workers through joblib): the device is touched on the first call only.
 *   - functions ending in _dev take DEVICE pointers (e.g. torch tensors' data_ptr())
 *     and a hipStream_t passed as void*; they never synchronise.
 *   - threading: different handles (glx_graph, glx_sweep, glx_dist_sweep) may be used from different
 *     threads at the same time.  The conjugate-gradient solves of ONE glx_graph share its work buffers:
 *     the glx_cg_* entry points serialise on the operator (a second thread waits).  A glx_sweep /
 *     glx_dist_sweep object is owned by one thread at a time -- concurrent calls on the same object are
 *     the caller's to exclude.
 */
#ifndef GLX_H
#define GLX_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GLX_F32 0
#define GLX_F64 1

#define GLX_OK 0
#define GLX_EINVAL (-1)   /* bad argument */
#define GLX_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define GLX_ENOMEM (-3)
#define GLX_EUNSUPPORTED (-4)
#define GLX_ERCCL (-5)    /* RCCL could not be loaded / a collective call failed */

typedef struct glx_graph glx_graph;   /* a sparse operator resident in HBM (sliced-ELL + CSR) */
typedef struct glx_sweep glx_sweep;   /* a prepared Poisson / heat sweep: device state + launch plan */
typedef struct glx_sweep_groups glx_sweep_groups;   /* several training sets as column groups of ONE prepared Poisson sweep */
typedef struct glx_cg glx_cg;         /* a prepared multi-RHS conjugate-gradient solve */
typedef struct glx_comm glx_comm;     /* one rank's RCCL communicator, owned by the library */
typedef struct glx_dist_sweep glx_dist_sweep;   /* one rank's share of the vertex-partitioned Poisson sweep */

const char* glx_last_error(void);
int glx_device_count(int* n);
void glx_free(void* p);

/* page-locked host memory for result arrays (a D2H copy into pageable memory is staged and several times slower;
 * the Python boundary recycles these blocks as the backing store of the numpy arrays it returns) */
int glx_host_alloc(size_t bytes, void** out);
int glx_host_free(void* p);

/* ---- sparse operator -------------------------------------------------------------
 * Replaces utils.torch_sparse (graphlearning/utils.py:288-317): scipy CSR -> device.
 * The entry order inside each CSR row is preserved: products accumulate a row's
 * entries sequentially in stored order with separate multiply and add roundings,
 * exactly like scipy's csr_matvecs, so fp64 results are bit-identical to `A * X`.
 * n_cols may exceed n_rows (rank-local operator with halo columns).
 * state_dtype selects the arithmetic/storage type of the dense operand (values are
 * converted from the fp64 input once at upload). */
int glx_graph_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                     const int32_t* col, const double* val, int state_dtype, int device,
                     glx_graph** out);
/* The same operator from CSR arrays that are kept on the DEVICE (the host keeps the row pointers only; a fresh ssl.poisson fit on
 * weightmatrix.knn's matrix: no host copy of the 14 MB operator, reference ssl.py:615-617, 634-635).  rowsum_out (n_rows fp64, may be
 * NULL): `A * ones` as scipy's csr_matvec forms it -- the degree vector of graph.py:108-122 -- computed on the device.
 * glx_graph_set_row_transform (before the first use): row i of the OPERATOR is row i of these arrays with its entries in reverse
 * order (reverse_rows != 0) and multiplied by row_scale[i] (NULL: unscaled): with row_scale = 1/degree and reverse_rows = 1 the
 * operator is P = D^-1 W^T of a symmetric W entry for entry as scipy's `D * W.transpose()` writes it. */
int glx_graph_create_resident(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                              const double* val, int state_dtype, int device, double* rowsum_out, glx_graph** out);
int glx_graph_set_row_transform(glx_graph* g, const double* row_scale, int reverse_rows);
int glx_graph_destroy(glx_graph* g);
/* Square operators are renumbered internally for cache locality (a breadth-first pass; dense operands passed as HOST arrays
 * are translated on the way in and out, results do not change -- a row's entries keep their order).  glx_graph_set_order, right
 * after creation, replaces that pass: perm[new] = caller's row, a permutation of 0..n-1 -- weightmatrix.knn has the features in
 * hand, and the cell order of its search needs no look at the graph --, or perm = NULL: the caller's order as it is (required for
 * the device-pointer (_dev) entry points of glx_experimental.h, whose records are in the caller's order). */
int glx_graph_set_order(glx_graph* g, const int32_t* perm);

/* u_out = Db + A u_in, applied `iters` times (u fed back).  Db may be NULL (no bias).
 * Replaces `ut = torch.sparse.addmm(Dbt, Pt, ut)` (ssl.py:658, :821) / `u = Db + P*u`
 * (ssl.py:668) / `u = P*u + Db` (ssl.py:827).  Host pointers, dtype = state_dtype. */
int glx_spmm_bias(glx_graph* A, const void* Db, const void* u_in, void* u_out, int C, int iters);

/* Poisson learning sweep, ssl.py:631-670: u <- Db + P u from u = 0 for T iterations,
 * T = first T >= min_iter with max_i |deg_i w_T[i] - vinf_i| <= 1/n (w_t = D^-1 v_t obeys
 * w_{t+1} = P w_t, so the reference's stop vector v rides along as one extra fp64
 * column), capped at max_iter.  w0 = D^-1 v0, deg, vinf: (n,) fp64.  u_out (n,C). */
int glx_poisson_sweep(glx_graph* P, const void* Db, const double* w0, const double* deg,
                      const double* vinf, int C, int min_iter, int max_iter, void* u_out,
                      int* T_out);

/* prepared form of the same (used by bench.py and repeated fits on one graph) */
int glx_sweep_create(glx_graph* P, int C, int min_iter, int max_iter, int use_hipgraph, glx_sweep** out);
int glx_sweep_set_problem(glx_sweep* s, const void* Db, const double* w0, const double* deg, const double* vinf);
/* the same problem with what a NEW TRAINING SET on a resident graph really changes: Db = D^-1 b and the initial
 * stop vector v0 are nonzero on the m labelled rows only (ssl.py:620-622, 639-641).  glx_sweep_set_vectors uploads
 * the graph's own vectors once (deg, vinf; (n,) fp64, caller's row order); glx_sweep_set_problem_rows then takes the
 * labelled rows (caller's numbering), their rows of Db ((m, C), state dtype) and of w0 = v0/deg ((m,) fp64), and
 * err0 = max|v0 - vinf| (the stop test before the first sweep; only read when min_iter = 0).  Rows set by the
 * previous call are cleared first. */
int glx_sweep_set_vectors(glx_sweep* s, const double* deg, const double* vinf);
int glx_sweep_set_problem_rows(glx_sweep* s, int64_t m, const int64_t* rows, const void* Db_rows,
                               const double* w0_rows, double err0);
int glx_sweep_run(glx_sweep* s, int* T_out, float* device_ms_out);   /* all iterations on device; HIP-event time.  With use_hipgraph the
                                                                        * FIRST run of the object launches its sweeps one by one and the launch
                                                                        * graph is captured on the second (a model fitted once never earns the
                                                                        * capture back); the iterates do not depend on the form. */
/* The stop values the last glx_sweep_run compared with 1/n (ssl.py:667): vals[i] = max|v_t - v_inf| for t = *first + i,
 * i < *count, the last one being the value that ended the loop (absent when max_iter did).  vals may be NULL to query
 * the counts.  The fused column computes them as deg*(P w), the reference as RW*v (ssl.py:669): equal up to rounding, so
 * a value within a few ulps of 1/n could decide differently; ssl.poisson re-derives T with the reference's recurrence
 * in that case (graphlearning_amd/ssl.py: _exact_stop_iteration). */
int glx_sweep_stop_values(const glx_sweep* s, int64_t cap, double* vals, int* first, int* count);
int glx_sweep_fetch(glx_sweep* s, void* u_out);
int glx_sweep_destroy(glx_sweep* s);

/* Heat/MBO inner loop, ssl.py:826-827: u <- P u + Db, `iters` times, u resident on
 * device between calls (glx_sweep created with min_iter = max_iter = 0 has no stop column).  glx_sweep_iterate ENQUEUES the sweeps and
 * returns: every call that reads or replaces the state (glx_sweep_project_iterate, glx_sweep_fetch, glx_sweep_set_state, another
 * glx_sweep_iterate) is ordered behind them in the sweep's own stream. */
int glx_sweep_set_state(glx_sweep* s, const void* u0, const void* Db);
/* The same with the state given as labels -- u = onehot(labels), labels (n,) int64 in the caller's row order -- and the bias by its m
 * nonzero rows (rows distinct; Db_rows (m, C) in the graph's dtype): PoissonMBO's start (ssl.py:797-805) moves n labels and m rows
 * instead of two dense (n, C) arrays. */
int glx_sweep_set_state_labels(glx_sweep* s, const int64_t* labels, int64_t m, const int64_t* rows, const void* Db_rows);
int glx_sweep_iterate(glx_sweep* s, int iters);

/* ---- stacked training sets: B fits of ssl.poisson(solver='gradient_descent') as ONE sweep ----
 * What ssl.ssl_trials does one `_fit` at a time (reference ssl.py:292-396 over ssl.py:631-670).  Trial b owns the columns
 * b C .. b C + C - 1 of the vertex record and an fp64 stop value of its own; group b runs sweep t iff t < min_iter or its
 * max|deg w_b - vinf| > 1/n held after sweep t - 1 (ssl.py:667); a group that has stopped neither gathers nor changes.  Every
 * trial's iterate and sweep count T are those of its own glx_sweep fit, bit for bit (fp64 state).  2 <= B <= 32 and
 * ceil(B C / 4) + ceil(B / 4) <= 64 (fp32 state: ceil(B / 2)).  Call order: create, set_vectors (the graph's deg and vinf, once),
 * then per batch of trials set_problem_rows for the groups in use, run(used), fetch / project per group. */
int glx_sweep_groups_create(glx_graph* P, int C, int B, int min_iter, int max_iter, glx_sweep_groups** out);
int glx_sweep_groups_set_vectors(glx_sweep_groups* s, const double* deg, const double* vinf);
/* training set of group b: m distinct labelled rows, their rows of Db = D^-1 b ((m, C), the graph's dtype) and of w0 = v_0 / deg,
 * err0 = max|v_0 - vinf| (the test in front of the first sweep when min_iter = 0); replaces the group's previous training set */
int glx_sweep_groups_set_problem_rows(glx_sweep_groups* s, int b, int64_t m, const int64_t* rows, const void* Db_rows,
                                      const double* w0_rows, double err0);
/* groups 0 .. used - 1 run from u = 0; T_out[B]: the sweeps of every group (0 for the idle ones) */
int glx_sweep_groups_run(glx_sweep_groups* s, int used, int* T_out, float* device_ms_out);
int glx_sweep_groups_stop_values(const glx_sweep_groups* s, int b, int64_t cap, double* vals, int* first, int* count);
int glx_sweep_groups_fetch(glx_sweep_groups* s, int b, void* u_out);              /* (n, C) of group b, caller order */
int glx_sweep_groups_project(glx_sweep_groups* s, int b, const double* priors, double* weights_inout, int64_t* labels_out,
                             double* err_out, int* steps_out, int max_steps, int similarity);   /* as glx_sweep_project_iterate with iters = 0, on group b */
int glx_sweep_groups_destroy(glx_sweep_groups* s);


/* ---- vertex-partitioned sweep over RCCL (the reference has no distributed code) -------------
 * One process per GPU: rank 0 calls glx_dist_unique_id, ships the 128 bytes to the other ranks by any means (the
 * launcher's store, torch.distributed, MPI ...), every rank calls glx_dist_init_rank.  glx_dist_init is the
 * one-process form (all GPUs of the node, ncclCommInitAll; out[nranks]).  id = NULL gives a rank identity without a
 * transport: enough for one rank, and for the stepwise form below where the caller moves the records.  RCCL is bound with dlopen on first use (librccl.so.1: the copy a host framework
 * already mapped, else /opt/rocm's). */
int glx_dist_unique_id(char id_out[128]);
int glx_dist_init_rank(int nranks, int rank, const char id[128], int device, glx_comm** out);
int glx_dist_init(int nranks, const int* devices, glx_comm** out);
int glx_dist_destroy(glx_comm* c);
/* This rank's share.  rowptr / col / val: its n_own rows of P in local order -- the n_boundary rows some peer
 * gathers FIRST -- with columns renumbered [0, n_own) owned and [n_own, n_own + n_halo) halo, the halo ordered as
 * the peers' records arrive (by owner rank ascending, each peer's in the order of that peer's send list).  Entry
 * order inside a row is kept, so iterates are bit-identical to the single-GPU sweep.  send_counts / recv_counts
 * [nranks]: records exchanged with every peer per sweep; send_idx: local (boundary) row of every record sent,
 * grouped by destination rank.  n_global: vertices of the whole graph (the stop threshold is 1/n_global).
 * force_exchange: issue the exchange even when this rank has nothing to send or receive.  Set it on EVERY rank as soon as ANY
 * rank has a halo: the collective calls around the exchange (the capture self-test's verdict, the stop test's all-reduce) must be
 * issued by all ranks alike (dist.glx_dist_sweep does; also used by 1-rank tests).
 * flags: GLX_DIST_CAPTURE alone is the default (the library picks the form of a sweep from the sizes of the interior and of the
 * exchange); the others force a form -- every one yields the same iterates, tests/test_gpu_dist.py runs them all. */
enum {
  GLX_DIST_CAPTURE = 1,             /* sweeps run as captured device graphs (else every launch is enqueued eagerly) */
  GLX_DIST_FORM_SPLIT = 2,          /* [boundary rows | exchange beside the interior rows]: two launches per sweep */
  GLX_DIST_FORM_FUSED = 4,          /* one launch for all rows, the exchange in line behind it */
  GLX_DIST_PACK_KERNEL = 8,         /* a pack kernel between SpMM and transport instead of the SpMM scattering into the send buffer */
  GLX_DIST_INLINE = 16,             /* the exchange on the sweep's own stream (no second stream) */
  GLX_DIST_EXCHANGE_CAPTURED = 32,  /* exchanging sweeps are captured without the self-test */
  GLX_DIST_EXCHANGE_EAGER = 64,     /* exchanging sweeps are enqueued eagerly */
  GLX_DIST_EXCHANGE_SELFTEST = 128, /* the first run decides by the self-test (default with real peers) */
  GLX_DIST_FORM_GATHER = 256        /* the exchange is ONE in-place ncclAllGather of the ranks' whole blocks (SURVEY 8e's fallback when a
                                       rank's halo is about all rows: an expander graph).  Another contract for the arguments: columns are
                                       numbered owner * cap + (row within the owner's block), n_halo = (nranks - 1) * cap with
                                       cap >= every rank's n_own, n_boundary = n_own; send / receive lists are ignored (may be NULL).
                                       No send lists, no pack, no send buffer; iterates bit-identical as ever. */
};
int glx_dist_sweep_create(glx_comm* comm, int64_t n_own, int64_t n_halo, int64_t n_boundary, const int32_t* rowptr,
                          const int32_t* col, const double* val, int state_dtype, int C, const int64_t* send_counts,
                          const int32_t* send_idx, const int64_t* recv_counts, int64_t n_global, int force_exchange,
                          int flags, glx_dist_sweep** out);
/* rank-local rows (local order) of Db (n_own, C; may be NULL), w0 = v0/deg, deg, vinf */
int glx_dist_sweep_set_problem(glx_dist_sweep* s, const void* Db_own, const double* w0_own, const double* deg_own,
                               const double* vinf_own);
/* ssl.py:631-670 across the ranks (collective call): every sweep is [boundary rows, which the SpMM also stores into
 * the send buffer | grouped ncclSend/ncclRecv all-to-all-v on a second stream | interior rows], the first min_iter
 * sweeps one captured device graph, later sweeps in chunks of check_every on a ring of state buffers with ONE
 * ncclAllReduce(MAX) of the chunk's per-sweep maxima -- no host round trip per sweep, and T and u_T are exactly the
 * reference's.  With real peers the FIRST call runs a self-test (three exchanging sweeps eagerly, then captured and
 * replayed, compared bit for bit on every rank, under a deadline of GLX_DIST_SELFTEST_TIMEOUT = 30 s): a clean pass
 * selects captured exchanging sweeps, anything else the eager form; GLX_DIST_CAPTURE_EXCHANGE=0/1 skips the test.
 * err0 = max|v0 - vinf| over all vertices (read when min_iter = 0). */
int glx_poisson_sweep_dist(glx_dist_sweep* s, int min_iter, int max_iter, int check_every, double err0, int* T_out,
                           float* device_ms_out);
int glx_dist_sweep_fetch(glx_dist_sweep* s, void* u_own_out);           /* (n_own, C) host, local row order */
int glx_dist_sweep_destroy(glx_dist_sweep* s);

/* ---- affine fixed-point iteration with a sup-norm stop --------------------------------
 * u <- A u + b (b may be NULL) from u0 until max|u_new - u_old| <= tol or max_iter sweeps: the power
 * iteration of graph.page_rank (graphlearning/graph.py:1371-1412) with A = alpha * W^T D^-1 and
 * b = (1-alpha) v, all iterations on the device.  b, u0, u_out: (n, C) host arrays in the operator's
 * state dtype.  iters_out: sweeps done; err_out: the last max|u_new - u_old|. */
int glx_affine_iterate(glx_graph* A, const void* b, const void* u0, void* u_out, int C, double tol,
                       int64_t max_iter, int64_t* iters_out, double* err_out);

/* ---- p-Laplace Jacobi iteration (graph.plaplace, fast=False) --------------------------
 * Replaces lp_iterate_main of the reference's C extension (c_code/lp_iterate.cpp:35-125; bound in
 * c_code/cextensions.cpp:19-60 as lp_iterate(uu, ul, I, J, W, ind, val, p, T, tol, prog)), same
 * argument order and the same in-place convention: uu / ul (n,) fp64 are the upper / lower barrier,
 * overwritten with the iterate the reference leaves in the caller's arrays (its swapped pointers are
 * local: U_it if it stops at an even iteration `it`, U_it+1 if odd).  nbr / row / W (M,): neighbour,
 * vertex (ascending) and weight of every stored entry; ind / val (m,): Dirichlet vertices and values.
 * iters_out: the iteration at which `err < tol && it > 10` held (T if never).  T <= 2^24. */
int glx_lp_iterate(double* uu, double* ul, const int32_t* nbr, const int32_t* row, const double* W,
                   const int32_t* ind, const double* val, double p, int64_t T, double tol,
                   int64_t n, int64_t M, int64_t m, int64_t* iters_out, int device);

/* ---- multi right-hand-side conjugate gradient -------------------------------------
 * Replaces utils.conjgrad (graphlearning/utils.py:483-532): x0 = 0, per-column
 * alpha/beta, global stop sqrt(sum over all columns ||r||^2) <= tol, max_iter cap. */
int glx_cg_multi(glx_graph* A, const void* B, void* X, int C, double tol, int64_t max_iter,
                 int* iters_out, double* err_out);
/* same with flags:
 *   GLX_CG_NP1D  the caller's right-hand side is 1-D (C must be 1): numpy then reduces with pairwise
 *                summation (graph.reweight, graphlearning/graph.py:429), reproduced exactly;
 *   GLX_CG_TREE  tolerance mode: the column reductions are deterministic block trees instead of numpy's
 *                row-after-row chains (about 4x faster per iteration; iterates agree with the reference to
 *                rounding, not bit for bit -- meant for the well-conditioned SPD systems of ssl.laplace /
 *                ssl.randomwalk / graph.reweight, not for the singular Poisson system);
 *   GLX_CG_X0    X holds the initial iterate x0 on entry and B the caller's r0 = b - A@x0
 *                (utils.py:510-514: `x = x0.copy(); r = b - A@x`); x then accumulates from x0 like the
 *                reference's `x += alpha * p`;
 *   GLX_CG_BLOCKS / GLX_CG_CHAIN  how the reference-order mode walks numpy's row-after-row reduction chains: in block form
 *                (integer block sums confirmed by the exact running sum, csrc/seqsum_exact.h) or one dependent addition per row.
 *                Same bits either way; the default takes the block form from 8192 rows on.  For tests and measurements.
 *   GLX_CG_EAGER the reference-order mode enqueues every iteration launch by launch instead of replaying chunks of 8 iterations from
 *                captured launch sequences (the default).  Same bits either way.  For tests and measurements. */
#define GLX_CG_NP1D 1
#define GLX_CG_TREE 2
#define GLX_CG_X0 4
#define GLX_CG_BLOCKS 8
#define GLX_CG_CHAIN 16
#define GLX_CG_EAGER 32
int glx_cg_solve(glx_graph* A, const void* B, void* X, int C, double tol, int64_t max_iter, int flags,
                 int* iters_out, double* err_out);
/* several independent systems on one operator, side by side: the C columns are C/group_cols systems
 * of group_cols columns each -- the trials of ssl.ssl_trials (graphlearning/ssl.py:292-396, one
 * utils.conjgrad call per trial there).  Every system keeps its own residual norm, stop test and
 * iteration count and is frozen once it converges, so column for column the result is identical
 * to solving it alone.  iters_out / err_out: C/group_cols entries.  C <= 252.
 * Dirichlet rows per system (mask_rows / mask_ptr; both NULL: none): the solve on the sub-matrix of the unlabelled vertices that
 * ssl.laplace._fit builds for every training set (graphlearning/ssl.py:1232-1250) is carried out on
 * the FULL operator by holding x, r, p at zero on the labelled rows (B must be zero there): their
 * columns then contribute exact zeros to every row sum and reduction, so the unlabelled rows of X and
 * the iteration counts are those of the sub-matrix solve, while the operator is uploaded once for all
 * training sets.  mask_rows: concatenated row numbers, system g owns [mask_ptr[g], mask_ptr[g+1]). */
int glx_cg_groups_masked(glx_graph* A, const void* B, void* X, int C, int group_cols, const int32_t* mask_rows,
                         const int32_t* mask_ptr, double tol, int64_t max_iter, int flags, int* iters_out,
                         double* err_out);

/* the same with the right-hand side given by its nonzero rows and the result scaled row by row on the way out -- what
 * ssl.laplace._fit needs per training set (ssl.py:1236-1250): b = -L[:,train]*F is nonzero only on the neighbours of the
 * labelled vertices, and `v = M*v` multiplies row i of the solution by M_ii.  b_rows (nb,) distinct rows, b_vals (nb, C) of the
 * operator's dtype; rows not listed are zero.  out_scale (n,) fp64 or NULL: X[i,:] = out_scale[i] * x[i,:] (one rounding, the
 * product numpy forms).  Saves the (n, C) upload and two host passes per fit; results identical to the dense form. */
int glx_cg_groups_rows(glx_graph* A, int64_t nb, const int32_t* b_rows, const void* b_vals, const double* out_scale, void* X,
                       int C, int group_cols, const int32_t* mask_rows, const int32_t* mask_ptr, double tol, int64_t max_iter,
                       int flags, int* iters_out, double* err_out);

/* ---- predict / volume-constrained projection --------------------------------------
 * ssl.predict (ssl.py:230-266) and ssl.volume_label_projection (ssl.py:172-209) on
 * device.  prob (n,C) fp64 host.  weights_inout (C,) fp64: all ones for a fresh model.
 * max_steps = 0 -> plain predict with the given weights.  similarity!=0 -> argmax. */
int glx_argmax_project(const double* prob, int64_t n, int C, const double* priors,
                       double* weights_inout, int64_t* labels_out, double* err_out,
                       int* steps_out, int max_steps, int similarity, int device);
/* the same for prob of either dtype.  GLX_F32 (the float32 `prob` the reference's use_cuda branch produces):
 * the subtraction and the division of ssl.py:256-257 round in float32 as numpy's do on a float32 array; only the
 * product with the fp64 class weights is wider. */
int glx_argmax_project_t(const void* prob, int prob_dtype, int64_t n, int C, const double* priors,
                         double* weights_inout, int64_t* labels_out, double* err_out,
                         int* steps_out, int max_steps, int similarity, int device);
/* the same decision on a sweep's device-resident state (no host round trip).  labels_out may be NULL.
 * to_onehot != 0 then replaces the state by onehot(labels): the hand-over between the heat sweeps and
 * the volume-constrained thresholding of ssl.poisson_mbo._fit (graphlearning/ssl.py:826-832). */
/* then_iterate > 0 (needs to_onehot): that many sweeps are enqueued behind the one-hot state before the call returns
 * (not awaited: the next call on this sweep is ordered behind them): PoissonMBO's thresholding and its next chunk of heat sweeps
 * (ssl.py:826-832) without a host round trip between them. */
int glx_sweep_project_iterate(glx_sweep* s, const double* priors, double* weights_inout, int64_t* labels_out, double* err_out,
                              int* steps_out, int max_steps, int similarity, int to_onehot, int then_iterate);


/* ---- kNN graph construction --------------------------------------------------------
 * weightmatrix.knnsearch (graphlearning/weightmatrix.py:297-429), exact: brute-force
 * tiled pairwise distances (fp32 MFMA candidate filter + fp64 direct-difference re-rank
 * with an exactness check and fp64 fallback).  k counts the self point.  X (n,d) fp64
 * host.  similarity must be 0 (euclidean); angular = euclidean on rows the caller normalised
 * (the Python boundary does that with the reference's own expression).  ind_out (n,k) int64, dist_out (n,k)
 * fp64, rows ascending by (distance, index).  Limits: k <= 60 (self included), d <= 16382; for d > 130 (or k > 28 with
 * d > 34) the candidate filter blocks the feature dimension instead of keeping the query in registers. */
int glx_knn_bruteforce(const double* X, int64_t n, int d, int k, int similarity,
                       int64_t* ind_out, double* dist_out, int device);
/* dist_out[i] = euclidean distance from row i of X (n, d) to the nearest of its rows idx[0 .. m): `cKDTree(X[idx]).query(X)[0]` of
 * graph.reweight(method='properly') (graphlearning/graph.py:455-457), all pairs with cKDTree's accumulation pattern -- the reference's
 * distances bit for bit.  Host arrays. */
int glx_nearest_dist(const double* X, int64_t n, int d, const int64_t* idx, int64_t m, double* dist_out, int device);

/* same search restricted to the query rows [q_begin, q_end) (rank-local share when the queries are
 * sharded across GPUs; every rank holds all of X).  ind_out/dist_out: (q_end - q_begin, k).  For this entry point and for
 * glx_knn_cells_range X may be a DEVICE pointer (features generated and ordered on the GPU: no trip through the host). */
int glx_knn_bruteforce_range(const double* X, int64_t n, int d, int k, int64_t q_begin, int64_t q_end,
                             int64_t* ind_out, double* dist_out, int device);
/* the same search -- the same lists, bit for bit -- for rows that come in a coarse geometric order: cell c = the rows
 * [cell_starts[c], cell_starts[c+1]) (host array of ncells <= 4096 ascending starts, cell_starts[0] = 0, the last cell ends at n).
 * Per block of 128 queries only the cells that can hold one of its k nearest neighbours are visited (centre / radius bounds
 * against the k-th distance within a sample of the block's own cells); what is skipped is strictly farther than the k-th
 * neighbour.  The role of the tree in the reference's search (cKDTree / annoy, graphlearning/weightmatrix.py:297-429). */
int glx_knn_cells_range(const double* X, int64_t n, int d, int k, const int64_t* cell_starts, int ncells, int64_t q_begin,
                        int64_t q_end, int64_t* ind_out, double* dist_out, int device);

/* ---- search results as objects (what weightmatrix.knn uses) ---------------------------------------------------------------------
 * glx_knn_search runs the full search (every row a query, self included; ncells as in glx_knn_clustered: 0 / 1 all pairs,
 * > 1 pruned by library-formed cells, < -1 all pairs on rows reordered by -ncells chained cells) and leaves the lists ON THE DEVICE
 * in a result object.  OWNERSHIP: the caller owns *out and releases it with glx_knn_result_destroy; the other calls borrow it.
 * The object may be used from any thread, by one thread at a time; nothing is handed from one call to the next through hidden
 * state (caller owns all buffers: the convention of the reference's c_code/cextensions.cpp:19-60). */
typedef struct glx_knn_result glx_knn_result;
int glx_knn_search(const double* X, int64_t n, int d, int k, int ncells, int device, glx_knn_result** out);
/* copies of the lists for the host: ind_out / dist_out (n, k), either may be NULL */
int glx_knn_result_lists(const glx_knn_result* res, int64_t* ind_out, double* dist_out);
/* perm_out[position] = caller's row in the cell order the search worked out (GLX_EINVAL: it formed no cells) */
int glx_knn_result_order(const glx_knn_result* res, int32_t* perm_out);
/* the weight matrix of the result (reference weightmatrix.py:134-187), k <= the result's columns, arguments otherwise as
 * glx_knn_to_csr_into; the Gaussian kernels evaluate a correctly rounded exp on the device (csrc/exp_cr.h) */
int glx_knn_result_to_csr(const glx_knn_result* res, int k, int kernel, int sym, const double* weights, int64_t cap,
                          int32_t* rowptr, int32_t* col, double* val, int64_t* nnz_out);
int glx_knn_result_destroy(glx_knn_result* res);

/* weightmatrix.knn given knn data (graphlearning/weightmatrix.py:134-187) on the device: kernel
 * weights, COO->CSR with duplicates summed, symmetrisation, zero diagonal, zeros dropped.
 * ind (n,kk) int64 and dist (n,kk) fp64 host arrays of which the first k columns are used
 * (k counts the self point).  kernel: 0 = `weights` (n,k) given (user eta), 1 uniform,
 * 2 gaussian, 3 symgaussian, 4 distance, 5 singular.  sym: 0 none, 1 (W+W^T)/2, 2 element-wise
 * max (utils.sparse_max), 3 the symgaussian rule.  Outputs are allocated by the library
 * (release with glx_free): canonical CSR, int32 indices like scipy's. */
int glx_knn_to_csr(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k,
                   int kernel, int sym, int32_t** rowptr_out, int32_t** col_out, double** val_out,
                   int64_t* nnz_out, int device);
/* the same with the CSR written into the caller's arrays: rowptr (n + 1), col and val with room for `cap` entries
 * (2 n k always suffices, n k without symmetrisation); GLX_EINVAL if the result does not fit. */
int glx_knn_to_csr_into(const int64_t* ind, const double* dist, const double* weights, int64_t n, int kk, int k,
                        int kernel, int sym, int64_t cap, int32_t* rowptr, int32_t* col, double* val,
                        int64_t* nnz_out, int device);
/* A BLOCK of rows [row_base, row_base + m) of the same matrix, for the sharded build (one rank's rows, SURVEY.md 8e
 * "symmetrisation by owner rank"): the block's own lists (ind_own, w_own: (m, k), weights given, column ids global < n_cols)
 * plus the reverse entries the owners of the other rows sent -- rev_row[e] = j (a row of the block), rev_src[e] = i,
 * rev_pos[e] = position of j in row i's list, rev_w[e] = w_ij.  Same merge as glx_knn_to_csr, so the rows are bit-identical
 * to the rows of the whole matrix.  sym: 0 none, 1 (W+W^T)/2, 2 element-wise max.  rowptr (m + 1), col / val (cap entries)
 * are the caller's; GLX_EINVAL if the result does not fit. */
int glx_knn_rows_to_csr(const int64_t* ind_own, const double* w_own, int64_t m, int k, int64_t n_cols, int64_t row_base,
                        const int64_t* rev_row, const int64_t* rev_src, const int64_t* rev_pos, const double* rev_w, int64_t n_rev,
                        int sym, int64_t cap, int32_t* rowptr, int32_t* col, double* val, int64_t* nnz_out, int device);

#ifdef __cplusplus
}
#endif
/* Laboratory and auxiliary entry points -- host helpers of the Python boundary, plan overrides and statistics for A/B runs, the
 * device-pointer calls of the torch fallback engine, the stepwise form of the distributed sweep for one-GPU multi-rank tests -- are
 * exported by the same library and declared in glx_experimental.h: no stability promise, not needed to use the path. */
#endif
