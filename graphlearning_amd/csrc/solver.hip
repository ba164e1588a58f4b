// Iteration drivers on top of the sliced-ELL SpMM: the Poisson sweep with its fused stop
// column (reference graphlearning/ssl.py:631-670), the PoissonMBO heat loop (ssl.py:826-827)
// and the one-shot glx_spmm_bias.  All iterations run on the device; the host only reads
// back the per-iteration stop-test maxima in small chunks.
#include "glx_internal.h"
#include <string.h>
#include <map>
#include <math.h>
#include <vector>
#include <algorithm>

static const int ERR_SHARDS = 64;
static const int TAIL_CHUNK = 16;

struct glx_sweep {
  glx_graph* P = nullptr;
  int device = 0;   // copied from P: destruction must not touch the operator (it may already be gone)
  int C = 0, min_iter = 0, max_iter = 0;
  bool has_w = false, use_graph = false;
  RecLayout L;
  SellPlan* plan = nullptr;
  int64_t n_rows = 0, n_cols = 0;
  void* buf[2] = {nullptr, nullptr};
  void* bias = nullptr;
  uint8_t* slot_has_bias = nullptr;
  bool bias_set = false;
  double *deg = nullptr, *vinf = nullptr, *w0 = nullptr;
  unsigned long long* err = nullptr;        // [(max_iter+1) * ERR_SHARDS]
  unsigned long long* h_err = nullptr;      // pinned mirror
  void* dense = nullptr;                    // staging (n_cols, C)
  glx_work* work = nullptr;                // stream + events of this sweep, from the per-device list of idle sets
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  glx_projector* proj = nullptr;            // label decision on the device-resident state (glx_sweep_project)
  hipGraphExec_t head_exec = nullptr;       // captured: reset + min_iter unconditional sweeps
  int64_t runs = 0;                         // glx_sweep_run calls so far (the first one launches eagerly)
  std::map<long, hipGraphExec_t> iter_exec; // heat loop graphs keyed by (iters, parity)
  int cur = 0;
  int64_t launches = 0;
  double err0 = 0.0, thresh = 0.0;
  int tested_first = 0, tested_count = 0;   // stop values err[t], t = tested_first .. +tested_count-1, of the last glx_sweep_run (in h_err)
  // sparse right-hand side (glx_sweep_set_problem_rows): record indices of the rows set by the previous call
  int32_t* row_slot = nullptr;              // [n_rows] record index -> its (first) slot in the plan
  int32_t* prev_rec = nullptr;              // [prev_cap] records whose bias / flag / w0 the previous problem set
  int64_t* prev_row = nullptr;              // [prev_cap] the same rows in the caller's numbering (for w0)
  int64_t prev_m = 0, prev_cap = 0;
  void* rows_stage = nullptr;               // device staging for (rows, Db_rows, w0_rows)
  size_t rows_stage_cap = 0;
  bool vectors_set = false;
};

static size_t rec_bytes(const glx_sweep* s, int64_t rows) { return (size_t)rows * s->L.ld * s->L.esize; }

extern "C" int glx_sweep_destroy(glx_sweep* s) {
  if (!s) return GLX_OK;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->head_exec) hipGraphExecDestroy(s->head_exec);
  for (auto& kv : s->iter_exec) hipGraphExecDestroy(kv.second);
  glx_pool_free(s->buf[0]);
  glx_pool_free(s->buf[1]);
  glx_pool_free(s->bias);
  glx_pool_free(s->slot_has_bias);
  glx_pool_free(s->deg);
  glx_pool_free(s->vinf);
  glx_pool_free(s->w0);
  glx_pool_free(s->err);
  glx_pinned_free(s->h_err);        // (the stream was synchronised at the top)
  glx_pool_free(s->dense);
  hipFree(s->row_slot);
  hipFree(s->prev_rec);
  hipFree(s->prev_row);
  hipFree(s->rows_stage);
  glx_projector_destroy(s->proj);
  glx_work_release(s->work);      // (the stream was synchronised at the top)
  delete s;
  return GLX_OK;
}

extern "C" int glx_sweep_create(glx_graph* P, int C, int min_iter, int max_iter, int use_hipgraph, glx_sweep** out) {
  GLX_CHECK(P && out, GLX_EINVAL, "glx_sweep_create: null argument");
  *out = nullptr;
  GLX_CHECK(min_iter >= 0 && max_iter >= 0, GLX_EINVAL, "glx_sweep_create: negative iteration bound");
  GLX_HIP(hipSetDevice(P->device));
  glx_sweep* s = new glx_sweep();
  s->P = P;
  s->device = P->device;
  s->C = C;
  s->min_iter = min_iter;
  s->max_iter = max_iter;
  s->has_w = max_iter > 0;
  s->use_graph = use_hipgraph != 0;
  s->n_rows = P->n_rows;
  s->n_cols = P->n_cols;
  int rc = glx_make_layout(C, P->dtype, s->has_w, &s->L);
  if (rc) { delete s; return rc; }
  rc = glx_graph_plan(P, s->L.G, &s->plan);
  if (rc) { delete s; return rc; }
// (work buffers from the size-class pool of graph.hip: a dozen hipMalloc / hipFree pairs per sweep object cost milliseconds)
#define SW_POOL(call) do { int rc_ = (call); if (rc_) { glx_sweep_destroy(s); return rc_; } } while (0)
#define SW_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { glx_set_error("%s -> %s", #call, hipGetErrorString(e_)); glx_sweep_destroy(s); return GLX_EHIP; } } while (0)
  rc = glx_work_acquire(P->device, &s->work);
  if (rc) { glx_sweep_destroy(s); return rc; }
  s->stream = s->work->stream;
  s->ev0 = s->work->ev[0];
  s->ev1 = s->work->ev[1];
  const size_t rb = std::max<size_t>(rec_bytes(s, s->n_cols), 64);
  SW_POOL(glx_pool_alloc((void**)&s->buf[0], rb));
  SW_POOL(glx_pool_alloc((void**)&s->buf[1], rb));
  // (on the sweep's own stream: it is non-blocking, so a null-stream memset -- asynchronous to the host --
  // would not be ordered against the first pack / upload and could land after it)
  SW_HIP(hipMemsetAsync(s->buf[0], 0, rb, s->stream));
  SW_HIP(hipMemsetAsync(s->buf[1], 0, rb, s->stream));
  SW_POOL(glx_pool_alloc((void**)&s->bias, std::max<size_t>(rec_bytes(s, s->n_rows), 64)));
  SW_HIP(hipMemsetAsync(s->bias, 0, std::max<size_t>(rec_bytes(s, s->n_rows), 64), s->stream));
  SW_POOL(glx_pool_alloc((void**)&s->slot_has_bias, std::max<size_t>((size_t)s->plan->nslices * s->plan->R, 64)));
  SW_POOL(glx_pool_alloc((void**)&s->dense, std::max<size_t>((size_t)s->n_cols * C * s->L.esize, 64)));
  if (s->has_w) {
    SW_POOL(glx_pool_alloc((void**)&s->deg, std::max<size_t>(s->n_rows * 8, 64)));
    SW_POOL(glx_pool_alloc((void**)&s->vinf, std::max<size_t>(s->n_rows * 8, 64)));
    SW_POOL(glx_pool_alloc((void**)&s->w0, std::max<size_t>(s->n_cols * 8, 64)));
    const size_t eb = (size_t)(max_iter + 1) * ERR_SHARDS * 8;
    SW_POOL(glx_pool_alloc((void**)&s->err, eb));
    SW_POOL(glx_pinned_alloc((void**)&s->h_err, eb));
  }
#undef SW_HIP
  *out = s;
  return GLX_OK;
}

// per-slot flag: does the row's bias record hold any nonzero?  (Poisson's Db = D^-1 b is
// nonzero on the m labelled rows only, ssl.py:620-622,636: the sweep then skips reading it.)
__global__ void bias_flags_kernel(const int32_t* __restrict__ slot_row, uint8_t* __restrict__ flags, int64_t nslots,
                                  const char* __restrict__ bias, int rec_bytes) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const int row = slot_row[slot];
  uint8_t f = 0;
  if (row >= 0) {
    const unsigned long long* r = (const unsigned long long*)(bias + (size_t)row * rec_bytes);
    for (int i = 0; i < rec_bytes / 8; ++i) f |= (r[i] << 1) != 0;   // -0.0 counts as zero
  }
  flags[slot] = f;
}

// captured launch sequences bake in whether a bias is read: drop them when that changes
static void drop_graphs_if_bias_changes(glx_sweep* s, bool will_have_bias) {
  if (s->bias_set == will_have_bias) return;
  if (s->head_exec) { hipGraphExecDestroy(s->head_exec); s->head_exec = nullptr; }
  for (auto& kv : s->iter_exec) hipGraphExecDestroy(kv.second);
  s->iter_exec.clear();
}

static int upload_bias(glx_sweep* s, const void* Db) {
  const int64_t nslots = s->plan->nslices * s->plan->R;
  drop_graphs_if_bias_changes(s, Db != nullptr);
  if (!Db) {
    GLX_HIP(hipMemsetAsync(s->bias, 0, rec_bytes(s, s->n_rows), s->stream));
    GLX_HIP(hipMemsetAsync(s->slot_has_bias, 0, std::max<int64_t>(nslots, 1), s->stream));
    s->bias_set = false;
    return GLX_OK;
  }
  GLX_UP(glx_upload(s->dense, Db, (size_t)s->n_rows * s->C * s->L.esize, s->stream, __func__));
  int rc = glx_pack_records(s->dense, s->bias, s->n_rows, s->L, s->P->dtype, nullptr, s->stream, s->P->d_perm);
  if (rc) return rc;
  if (nslots > 0) {
    hipLaunchKernelGGL(bias_flags_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s->stream,
                       s->plan->d_slot_row, s->slot_has_bias, nslots, (const char*)s->bias, s->L.ld * s->L.esize);
    GLX_HIP(hipGetLastError());
  }
  s->bias_set = true;
  return GLX_OK;
}

extern "C" int glx_sweep_set_problem(glx_sweep* s, const void* Db, const double* w0, const double* deg, const double* vinf) {
  GLX_CHECK(s, GLX_EINVAL, "glx_sweep_set_problem: null sweep");
  GLX_CHECK(s->has_w, GLX_EINVAL, "glx_sweep_set_problem: sweep was created without a stop column (max_iter = 0)");
  GLX_CHECK(w0 && deg && vinf, GLX_EINVAL, "glx_sweep_set_problem: null vector");
  GLX_CHECK(s->n_rows == s->n_cols, GLX_EINVAL, "glx_sweep_set_problem: operator must be square");
  GLX_HIP(hipSetDevice(s->P->device));
  int rc = upload_bias(s, Db);
  if (rc) return rc;
  if (s->row_slot) { hipFree(s->row_slot); s->row_slot = nullptr; s->prev_m = 0; }   // a later sparse problem starts from scratch
  s->vectors_set = true;
  GLX_UP(glx_upload(s->w0, w0, s->n_cols * 8, s->stream, __func__));   // caller order; packed through perm
  std::vector<double> degp, vinfp;
  if (!s->P->h_perm.empty()) {   // the kernel indexes deg / vinf by renumbered row
    degp.resize(s->n_rows);
    vinfp.resize(s->n_rows);
    for (int64_t i = 0; i < s->n_rows; ++i) { degp[i] = deg[s->P->h_perm[i]]; vinfp[i] = vinf[s->P->h_perm[i]]; }
  }
  GLX_UP(glx_upload(s->deg, degp.empty() ? deg : degp.data(), s->n_rows * 8, s->stream, __func__));
  GLX_UP(glx_upload(s->vinf, vinfp.empty() ? vinf : vinfp.data(), s->n_rows * 8, s->stream, __func__));
  double e0 = 0.0;
  for (int64_t i = 0; i < s->n_rows; ++i) {
    const double e = fabs(deg[i] * w0[i] - vinf[i]);
    if (e > e0 || e != e) e0 = e;
    if (e != e) break;   // np.max keeps the NaN
  }
  s->err0 = e0;
  s->thresh = 1.0 / (double)s->n_rows;   // `> 1/n`, ssl.py:667
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// ---- sparse problem upload -------------------------------------------------------------------------
// Poisson's right-hand side Db = D^-1 b and the initial stop vector v0 are nonzero on the m labelled rows only
// (ssl.py:620-622, 639-641): a new training set on a resident graph then costs a few KB of upload instead of a
// dense (n, C) array, and the graph's own vectors (deg, vinf) are uploaded once.
__global__ void permute_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ perm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm ? perm[i] : i];
}

extern "C" int glx_sweep_set_vectors(glx_sweep* s, const double* deg, const double* vinf) {
  GLX_CHECK(s && deg && vinf, GLX_EINVAL, "glx_sweep_set_vectors: null argument");
  GLX_CHECK(s->has_w, GLX_EINVAL, "glx_sweep_set_vectors: sweep was created without a stop column (max_iter = 0)");
  GLX_CHECK(s->n_rows == s->n_cols, GLX_EINVAL, "glx_sweep_set_vectors: operator must be square");
  GLX_HIP(hipSetDevice(s->device));
  double* tmp = (double*)s->dense;   // staging (n, C) of the state dtype: C*esize >= 4 bytes per row is not enough for 2 x fp64
  void* own = nullptr;
  if ((size_t)s->C * s->L.esize < 16) { GLX_HIP(hipMalloc(&own, (size_t)s->n_rows * 16)); tmp = (double*)own; }
  const unsigned grid = (unsigned)((s->n_rows + 255) / 256);
  int rcu = glx_upload(tmp, deg, s->n_rows * 8, s->stream, __func__);
  if (!rcu) rcu = glx_upload(tmp + s->n_rows, vinf, s->n_rows * 8, s->stream, __func__);
  if (rcu) { hipFree(own); return rcu; }
  hipError_t e = hipSuccess;
  if (s->n_rows > 0) {
    hipLaunchKernelGGL(permute_f64_kernel, dim3(grid), dim3(256), 0, s->stream, (const double*)tmp, s->deg, (const int32_t*)s->P->d_perm, s->n_rows);
    hipLaunchKernelGGL(permute_f64_kernel, dim3(grid), dim3(256), 0, s->stream, (const double*)(tmp + s->n_rows), s->vinf,
                       (const int32_t*)s->P->d_perm, s->n_rows);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  hipFree(own);
  GLX_HIP(e);
  s->thresh = 1.0 / (double)s->n_rows;   // `> 1/n`, ssl.py:667
  s->vectors_set = true;
  return GLX_OK;
}

__global__ void row_slot_kernel(const int32_t* __restrict__ slot_row, int32_t* __restrict__ row_slot, int64_t nslots) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const int row = slot_row[slot];
  // the first of a long row's S consecutive slots is the one whose lanes store (and read the bias flag)
  if (row >= 0 && (slot == 0 || slot_row[slot - 1] != row)) row_slot[row] = (int32_t)slot;
}

template <typename T>
__global__ void clear_rows_kernel(char* __restrict__ bias, int rec_bytes, uint8_t* __restrict__ flags, const int32_t* __restrict__ row_slot,
                                  double* __restrict__ w0, const int32_t* __restrict__ prev_rec, const int64_t* __restrict__ prev_row,
                                  int64_t m, int ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * ld) return;
  const int64_t q = i / ld;
  const int c = (int)(i % ld);
  const int32_t rec = prev_rec[q];
  ((T*)(bias + (size_t)rec * rec_bytes))[c] = 0;
  if (c == 0) {
    flags[row_slot[rec]] = 0;
    w0[prev_row[q]] = 0.0;
  }
}

template <typename T>
__global__ void set_rows_kernel(char* __restrict__ bias, int rec_bytes, uint8_t* __restrict__ flags, const int32_t* __restrict__ row_slot,
                                double* __restrict__ w0, const int64_t* __restrict__ rows, const int32_t* __restrict__ inv,
                                const T* __restrict__ Db_rows, const double* __restrict__ w0_rows, int64_t m, int C,
                                int32_t* __restrict__ prev_rec, int64_t* __restrict__ prev_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * C) return;
  const int64_t q = i / C;
  const int c = (int)(i % C);
  const int64_t row = rows[q];
  const int32_t rec = inv ? inv[row] : (int32_t)row;
  const T v = Db_rows[i];
  ((T*)(bias + (size_t)rec * rec_bytes))[c] = v;
  // flag: any nonzero in the record (-0.0 counts as zero, like bias_flags_kernel); benign race: every writer stores 1
  bool nz;
  if constexpr (sizeof(T) == 8) nz = ((unsigned long long)__double_as_longlong((double)v) << 1) != 0;
  else nz = ((unsigned)__float_as_int((float)v) << 1) != 0;
  if (nz) flags[row_slot[rec]] = 1;
  if (c == 0) {
    w0[row] = w0_rows[q];
    prev_rec[q] = rec;
    prev_row[q] = row;
  }
}

extern "C" int glx_sweep_set_problem_rows(glx_sweep* s, int64_t m, const int64_t* rows, const void* Db_rows, const double* w0_rows,
                                          double err0) {
  GLX_CHECK(s, GLX_EINVAL, "glx_sweep_set_problem_rows: null sweep");
  GLX_CHECK(s->has_w, GLX_EINVAL, "glx_sweep_set_problem_rows: sweep was created without a stop column (max_iter = 0)");
  GLX_CHECK(s->vectors_set, GLX_EINVAL, "glx_sweep_set_problem_rows: call glx_sweep_set_vectors first");
  GLX_CHECK(m >= 0 && (m == 0 || (rows && Db_rows && w0_rows)), GLX_EINVAL, "glx_sweep_set_problem_rows: null array");
  for (int64_t q = 0; q < m; ++q)
    GLX_CHECK(rows[q] >= 0 && rows[q] < s->n_rows, GLX_EINVAL, "glx_sweep_set_problem_rows: row %lld out of range", (long long)rows[q]);
  GLX_HIP(hipSetDevice(s->device));
  const int64_t nslots = s->plan->nslices * s->plan->R;
  const int rb = s->L.ld * s->L.esize;
  if (!s->row_slot) {
    GLX_HIP(hipMalloc(&s->row_slot, std::max<size_t>((size_t)s->n_rows * 4, 64)));
    if (nslots > 0) {
      hipLaunchKernelGGL(row_slot_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s->stream, (const int32_t*)s->plan->d_slot_row,
                         s->row_slot, nslots);
      GLX_HIP(hipGetLastError());
    }
    // from a dense problem to sparse ones: start from an all-zero bias
    GLX_HIP(hipMemsetAsync(s->bias, 0, rec_bytes(s, s->n_rows), s->stream));
    GLX_HIP(hipMemsetAsync(s->slot_has_bias, 0, std::max<int64_t>(nslots, 1), s->stream));
    GLX_HIP(hipMemsetAsync(s->w0, 0, std::max<size_t>(s->n_cols * 8, 8), s->stream));
    s->prev_m = 0;
  }
  drop_graphs_if_bias_changes(s, true);
  if (s->prev_m > 0) {
    const int64_t tot = s->prev_m * s->L.ld;
    if (s->P->dtype == GLX_F32)
      hipLaunchKernelGGL(clear_rows_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb, s->slot_has_bias,
                         (const int32_t*)s->row_slot, s->w0, (const int32_t*)s->prev_rec, (const int64_t*)s->prev_row, s->prev_m, s->L.ld);
    else
      hipLaunchKernelGGL(clear_rows_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb, s->slot_has_bias,
                         (const int32_t*)s->row_slot, s->w0, (const int32_t*)s->prev_rec, (const int64_t*)s->prev_row, s->prev_m, s->L.ld);
    GLX_HIP(hipGetLastError());
    s->prev_m = 0;
  }
  if (m > 0) {
    if (s->prev_cap < m) {
      hipFree(s->prev_rec);
      hipFree(s->prev_row);
      s->prev_rec = nullptr;
      s->prev_row = nullptr;
      s->prev_cap = 0;
      const int64_t cap = std::max<int64_t>(m * 2, 256);
      GLX_HIP(hipMalloc(&s->prev_rec, cap * 4));
      GLX_HIP(hipMalloc(&s->prev_row, cap * 8));
      s->prev_cap = cap;
    }
    const size_t es = s->L.esize;
    const size_t b_rows = (size_t)m * 8, b_db = ((size_t)m * s->C * es + 7) / 8 * 8, b_w = (size_t)m * 8;
    if (s->rows_stage_cap < b_rows + b_db + b_w) {
      hipFree(s->rows_stage);
      s->rows_stage = nullptr;
      s->rows_stage_cap = 0;
      GLX_HIP(hipMalloc(&s->rows_stage, 2 * (b_rows + b_db + b_w)));
      s->rows_stage_cap = 2 * (b_rows + b_db + b_w);
    }
    char* st = (char*)s->rows_stage;
    GLX_UP(glx_upload(st, rows, b_rows, s->stream, __func__));
    GLX_UP(glx_upload(st + b_rows, Db_rows, (size_t)m * s->C * es, s->stream, __func__));
    GLX_UP(glx_upload(st + b_rows + b_db, w0_rows, b_w, s->stream, __func__));
    const int64_t tot = m * s->C;
    if (s->P->dtype == GLX_F32)
      hipLaunchKernelGGL(set_rows_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb, s->slot_has_bias,
                         (const int32_t*)s->row_slot, s->w0, (const int64_t*)st, (const int32_t*)s->P->d_inv, (const float*)(st + b_rows),
                         (const double*)(st + b_rows + b_db), m, s->C, s->prev_rec, s->prev_row);
    else
      hipLaunchKernelGGL(set_rows_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb, s->slot_has_bias,
                         (const int32_t*)s->row_slot, s->w0, (const int64_t*)st, (const int32_t*)s->P->d_inv, (const double*)(st + b_rows),
                         (const double*)(st + b_rows + b_db), m, s->C, s->prev_rec, s->prev_row);
    GLX_HIP(hipGetLastError());
    s->prev_m = m;
  }
  s->bias_set = true;
  s->err0 = err0;
  GLX_HIP(hipStreamSynchronize(s->stream));   // the host arrays may go; the staging is reused by the next call
  return GLX_OK;
}

static int enqueue_iterate(glx_sweep* s, int iters);

static int launch_sweep(glx_sweep* s, int t, bool with_stop) {
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.plan = s->plan;
  a.L = s->L;
  a.dtype = s->P->dtype;
  a.xin = s->buf[s->cur];
  a.xout = s->buf[s->cur ^ 1];
  a.bias = s->bias_set ? s->bias : nullptr;
  a.slot_has_bias = s->bias_set ? s->slot_has_bias : nullptr;
  a.has_w = s->has_w;
  a.n_rows = s->n_rows;
  if (with_stop) {
    a.deg = s->deg;
    a.vinf = s->vinf;
    a.thresh = s->thresh;
    a.err_prev = t >= s->min_iter ? s->err + (size_t)t * ERR_SHARDS : nullptr;
    a.err_next = t + 1 >= s->min_iter ? s->err + (size_t)(t + 1) * ERR_SHARDS : nullptr;
  }
  int rc = glx_launch_spmm(a, s->stream);
  if (rc) return rc;
  s->cur ^= 1;
  s->launches++;
  return GLX_OK;
}

// reset state + the min_iter unconditional sweeps; identical every run -> capturable
static int enqueue_head(glx_sweep* s) {
  // the stop values the head's sweeps write: rows min_iter (written by sweep min_iter - 1) and, for min_iter = 0, row 0; the rows of a
  // tail chunk are cleared in front of that chunk (glx_sweep_run) -- not all max_iter + 1 rows on every run (512 KB, 6.6 us of a
  // 620 us step at config 2)
  {
    const int head_rows = std::min(s->min_iter, s->max_iter);
    // (a kernel, not a memset node: this sequence is captured and replayed -- glx_zero_async, glx_internal.h)
    { int rz = glx_zero_async(s->err + (size_t)head_rows * ERR_SHARDS, (size_t)ERR_SHARDS * 8, s->stream); if (rz) return rz; }
  }
  if (s->min_iter == 0) {
    union { double d; unsigned long long u; } cv;
    cv.d = s->err0;
    s->h_err[0] = cv.u;
    GLX_HIP(hipMemcpyAsync(s->err, s->h_err, 8, hipMemcpyHostToDevice, s->stream));
  }
  s->cur = 0;
  int rc = glx_pack_records(nullptr, s->buf[0], s->n_cols, s->L, s->P->dtype, s->w0, s->stream, s->P->d_perm);   // u = 0 (ssl.py:645), w = w0
  if (rc) return rc;
  const int head = std::min(s->min_iter, s->max_iter);
  for (int t = 0; t < head; ++t) {
    rc = launch_sweep(s, t, true);
    if (rc) return rc;
  }
  return GLX_OK;
}

extern "C" int glx_sweep_run(glx_sweep* s, int* T_out, float* device_ms_out) {
  GLX_CHECK(s && s->has_w, GLX_EINVAL, "glx_sweep_run: sweep has no stop column");
  GLX_HIP(hipSetDevice(s->P->device));
  const int head = std::min(s->min_iter, s->max_iter);
  int rc;
  const int64_t launches0 = s->launches;
  GLX_HIP(hipEventRecord(s->ev0, s->stream));
  // A model that is fitted once never earns its launch graph back: capture + instantiate + the first launch of a fresh executable graph +
  // its destruction cost 1.1-1.4 ms against 0.35 ms of launching the same 50 sweeps one by one (they run behind each other either way:
  // 0.62 ms of device time; profiles/r06_fresh_path.txt).  The first run of a sweep object therefore launches eagerly; the graph is captured
  // on the second run and replayed from then on.  Same kernels, same order: the iterates do not depend on the form.
  const bool replay = s->use_graph && (s->runs > 0 || s->head_exec);
  ++s->runs;
  if (replay) {
    if (!s->head_exec) {
      hipGraph_t graph;
      GLX_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
      rc = enqueue_head(s);
      hipError_t e = hipStreamEndCapture(s->stream, &graph);
      if (rc) return rc;
      GLX_HIP(e);
      GLX_HIP(hipGraphInstantiate(&s->head_exec, graph, nullptr, nullptr, 0));
      GLX_HIP(hipGraphDestroy(graph));
    } else {
      s->launches += head;
    }
    if (s->min_iter == 0) {
      union { double d; unsigned long long u; } cv;
      cv.d = s->err0;
      s->h_err[0] = cv.u;
    }
    GLX_HIP(hipGraphLaunch(s->head_exec, s->stream));
    s->cur = head & 1;
  } else {
    rc = enqueue_head(s);
    if (rc) return rc;
  }
  // tail: conditional sweeps in chunks; each kernel exits at once when the stop test
  // (read from the previous kernel's maxima) already holds, so over-launching is harmless.
  int T = head;
  int t = head;
  bool stopped = false;
  while (!stopped && t < s->max_iter) {
    // check err[t] first
    GLX_HIP(hipMemcpyAsync(s->h_err + (size_t)t * ERR_SHARDS, s->err + (size_t)t * ERR_SHARDS, ERR_SHARDS * 8, hipMemcpyDeviceToHost, s->stream));
    GLX_HIP(hipStreamSynchronize(s->stream));
    {
      unsigned long long m = 0;
      for (int k = 0; k < ERR_SHARDS; ++k) m = std::max(m, s->h_err[(size_t)t * ERR_SHARDS + k]);
      union { double d; unsigned long long u; } cv;
      cv.u = m;
      if (!(cv.d > s->thresh)) { stopped = true; T = t; break; }   // NaN stops the loop, like `nan > 1/n` (ssl.py:667)
    }
    const int end = std::min(s->max_iter, t + TAIL_CHUNK);
    const int t0 = t;
    const int cur0 = s->cur;
    // rows t0 + 1 .. end: what this chunk's sweeps write (atomic maxima: they must start from 0)
    GLX_HIP(hipMemsetAsync(s->err + (size_t)(t0 + 1) * ERR_SHARDS, 0, (size_t)(end - t0) * ERR_SHARDS * 8, s->stream));
    for (; t < end; ++t) {
      rc = launch_sweep(s, t, true);
      if (rc) return rc;
    }
    // which of the chunk's sweeps really ran?
    GLX_HIP(hipMemcpyAsync(s->h_err + (size_t)(t0 + 1) * ERR_SHARDS, s->err + (size_t)(t0 + 1) * ERR_SHARDS,
                           (size_t)(end - t0) * ERR_SHARDS * 8, hipMemcpyDeviceToHost, s->stream));
    GLX_HIP(hipStreamSynchronize(s->stream));
    T = end;
    for (int q = t0 + 1; q < end; ++q) {   // sweep q ran iff err[q] > thresh
      unsigned long long m = 0;
      for (int k = 0; k < ERR_SHARDS; ++k) m = std::max(m, s->h_err[(size_t)q * ERR_SHARDS + k]);
      union { double d; unsigned long long u; } cv;
      cv.u = m;
      if (!(cv.d > s->thresh)) { T = q; stopped = true; break; }
    }
    if (stopped) s->cur = cur0 ^ ((T - t0) & 1);
    t = stopped ? T : end;
  }
  if (!stopped) T = t;
  GLX_HIP(hipEventRecord(s->ev1, s->stream));
  GLX_HIP(hipStreamSynchronize(s->stream));
  if (device_ms_out) GLX_HIP(hipEventElapsedTime(device_ms_out, s->ev0, s->ev1));
  if (T_out) *T_out = T;
  // count the sweeps that really ran: kernels of a tail chunk launched past the stop test exit at once
  s->launches = launches0 + T;
  s->tested_first = head;
  s->tested_count = T - head + (stopped ? 1 : 0);   // err[head..T]; the value at T = max_iter is never compared (ssl.py:667)
  return GLX_OK;
}

// The stop values the last glx_sweep_run compared with 1/n: vals[i] = max|v_t - v_inf| at t = *first + i, the last one
// being the value at t = T.  They are computed as deg*(P w) (the fused column), not as the reference's separate
// product RW*v (ssl.py:669): equal up to rounding.  The caller (ssl.poisson) uses them to recognise the one case in
// which the two roundings could decide differently -- a value within a few ulps of 1/n -- and settles that case with
// the reference's own recurrence.
extern "C" int glx_sweep_stop_values(const glx_sweep* s, int64_t cap, double* vals, int* first, int* count) {
  GLX_CHECK(s && first && count, GLX_EINVAL, "glx_sweep_stop_values: null argument");
  *first = s->tested_first;
  *count = s->tested_count;
  if (!vals) return GLX_OK;
  GLX_CHECK(cap >= s->tested_count, GLX_EINVAL, "glx_sweep_stop_values: %lld values, room for %lld", (long long)s->tested_count, (long long)cap);
  for (int i = 0; i < s->tested_count; ++i) {
    const size_t t = (size_t)(s->tested_first + i);
    unsigned long long m = 0;
    for (int k = 0; k < ERR_SHARDS; ++k) m = std::max(m, s->h_err[t * ERR_SHARDS + k]);
    union { double d; unsigned long long u; } cv;
    cv.u = m;
    vals[i] = cv.d;
  }
  return GLX_OK;
}

extern "C" int glx_sweep_fetch(glx_sweep* s, void* u_out) {
  GLX_CHECK(s && u_out, GLX_EINVAL, "glx_sweep_fetch: null argument");
  GLX_HIP(hipSetDevice(s->P->device));
  int rc = glx_unpack_records(s->buf[s->cur], s->dense, s->n_rows, s->L, s->P->dtype, s->stream, s->P->d_perm);
  if (rc) return rc;
  GLX_UP(glx_download(u_out, s->dense, (size_t)s->n_rows * s->C * s->L.esize, s->stream, __func__));
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// ssl.predict / ssl.volume_label_projection (ssl.py:230-266, 172-209) on the sweep's current state
// without a host round trip; with to_onehot the state is then replaced by onehot(labels), which is
// the hand-over between the heat sweeps and the thresholding of PoissonMBO (ssl.py:826-832).
// then_iterate > 0 (needs to_onehot): that many sweeps are enqueued behind the one-hot state before the host has even seen the
// decision -- PoissonMBO's next heat chunk (ssl.py:826-832) starts where the thresholding ends instead of two host round trips later
// (~75 us of idle device per outer step at config 5, profiles/r04_mbo_step.txt).
extern "C" int glx_sweep_project_iterate(glx_sweep* s, const double* priors, double* weights_inout, int64_t* labels_out, double* err_out,
                                         int* steps_out, int max_steps, int similarity, int to_onehot, int then_iterate) {
  GLX_CHECK(s && weights_inout, GLX_EINVAL, "glx_sweep_project: null argument");
  GLX_CHECK(!s->has_w || !to_onehot, GLX_EINVAL, "glx_sweep_project: to_onehot needs a sweep created with max_iter = 0");
  GLX_CHECK(then_iterate >= 0 && (then_iterate == 0 || (to_onehot && s->n_rows == s->n_cols)), GLX_EINVAL,
            "glx_sweep_project: then_iterate needs to_onehot and a square operator");
  GLX_HIP(hipSetDevice(s->device));
  const int dtype = s->P->dtype;
  // an fp64 state is unpacked straight into the projector's own (n, C) array; a float32 one goes through `dense` and is widened
  void* prob = s->dense;
  int rc;
  if (dtype == GLX_F64) {
    double* scores = nullptr;
    rc = glx_project_scores(&s->proj, s->n_rows, s->C, &scores);
    if (rc) return rc;
    prob = scores;
  }
  rc = glx_unpack_records(s->buf[s->cur], prob, s->n_rows, s->L, dtype, s->stream, s->P->d_perm);
  if (rc) return rc;
  const long long* d_labels = nullptr;
  // what goes with the decision (the labels, waited for) and what follows it (the one-hot state, the next sweeps: enqueued, not
  // awaited -- the next call on this sweep is ordered behind them in its stream)
  const std::function<int(bool)> hook = [&](bool after) -> int {
    if (!after) {
      if (labels_out) GLX_UP(glx_download(labels_out, d_labels, (size_t)s->n_rows * 8, s->stream, __func__));
      return GLX_OK;
    }
    if (to_onehot) {
      int rh = glx_onehot_records(d_labels, s->buf[s->cur], dtype, s->n_cols, s->L, s->P->d_perm, s->stream);
      if (rh) return rh;
    }
    return then_iterate > 0 ? enqueue_iterate(s, then_iterate) : GLX_OK;
  };
  return glx_project_device(&s->proj, prob, dtype, s->n_rows, s->C, priors, weights_inout, err_out, steps_out, max_steps, similarity,
                            s->stream, &d_labels, &hook);
}

extern "C" int glx_sweep_launches(const glx_sweep* s, int64_t* n) {
  GLX_CHECK(s && n, GLX_EINVAL, "glx_sweep_launches: null argument");
  *n = s->launches;
  return GLX_OK;
}

extern "C" int glx_sweep_set_state(glx_sweep* s, const void* u0, const void* Db) {
  GLX_CHECK(s, GLX_EINVAL, "glx_sweep_set_state: null sweep");
  GLX_CHECK(!s->has_w, GLX_EINVAL, "glx_sweep_set_state: only for sweeps created with max_iter = 0");
  GLX_HIP(hipSetDevice(s->P->device));
  int rc = upload_bias(s, Db);
  if (rc) return rc;
  GLX_HIP(hipStreamSynchronize(s->stream));   // `dense` staging is reused below
  if (u0) {
    GLX_UP(glx_upload(s->dense, u0, (size_t)s->n_cols * s->C * s->L.esize, s->stream, __func__));
    rc = glx_pack_records(s->dense, s->buf[0], s->n_cols, s->L, s->P->dtype, nullptr, s->stream, s->P->d_perm);
  } else {
    rc = glx_pack_records(nullptr, s->buf[0], s->n_cols, s->L, s->P->dtype, nullptr, s->stream);
  }
  if (rc) return rc;
  s->cur = 0;
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// rows of a bias given one by one (heat sweeps: no stop column, no flags per row needed beyond bias_flags_kernel's pass)
template <typename T>
__global__ void scatter_bias_rows_kernel(char* __restrict__ bias, int rec_bytes, const int64_t* __restrict__ rows, const int32_t* __restrict__ inv,
                                         const T* __restrict__ Db_rows, int64_t m, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * C) return;
  const int64_t row = rows[i / C];
  const int32_t rec = inv ? inv[row] : (int32_t)row;
  ((T*)(bias + (size_t)rec * rec_bytes))[i % C] = Db_rows[i];
}

extern "C" int glx_sweep_set_state_labels(glx_sweep* s, const int64_t* labels, int64_t m, const int64_t* rows, const void* Db_rows) {
  GLX_CHECK(s && labels, GLX_EINVAL, "glx_sweep_set_state_labels: null argument");
  GLX_CHECK(!s->has_w, GLX_EINVAL, "glx_sweep_set_state_labels: only for sweeps created with max_iter = 0");
  GLX_CHECK(s->n_rows == s->n_cols, GLX_EINVAL, "glx_sweep_set_state_labels: operator must be square");
  GLX_CHECK(m >= 0 && (m == 0 || (rows && Db_rows)), GLX_EINVAL, "glx_sweep_set_state_labels: null array");
  for (int64_t q = 0; q < m; ++q)
    GLX_CHECK(rows[q] >= 0 && rows[q] < s->n_rows, GLX_EINVAL, "glx_sweep_set_state_labels: row %lld out of range", (long long)rows[q]);
  GLX_HIP(hipSetDevice(s->P->device));
  const int64_t nslots = s->plan->nslices * s->plan->R;
  const int rb = s->L.ld * s->L.esize;
  const size_t es = s->L.esize;
  const size_t b_lab = (size_t)s->n_rows * 8, b_rows = (size_t)m * 8, b_db = ((size_t)m * s->C * es + 7) / 8 * 8;
  if (s->rows_stage_cap < b_lab + b_rows + b_db) {
    GLX_HIP(hipStreamSynchronize(s->stream));
    hipFree(s->rows_stage);
    s->rows_stage = nullptr;
    s->rows_stage_cap = 0;
    GLX_HIP(hipMalloc(&s->rows_stage, b_lab + 2 * (b_rows + b_db) + 64));
    s->rows_stage_cap = b_lab + 2 * (b_rows + b_db) + 64;
  }
  char* st = (char*)s->rows_stage;
  drop_graphs_if_bias_changes(s, m > 0);
  GLX_UP(glx_upload(st, labels, b_lab, s->stream, __func__));
  int rc = glx_onehot_records((const long long*)st, s->buf[0], s->P->dtype, s->n_rows, s->L, s->P->d_perm, s->stream);   // u = onehot(labels)
  if (rc) return rc;
  GLX_HIP(hipMemsetAsync(s->bias, 0, rec_bytes(s, s->n_rows), s->stream));
  GLX_HIP(hipMemsetAsync(s->slot_has_bias, 0, std::max<int64_t>(nslots, 1), s->stream));
  if (m > 0) {
    GLX_UP(glx_upload(st + b_lab, rows, b_rows, s->stream, __func__));
    GLX_UP(glx_upload(st + b_lab + b_rows, Db_rows, (size_t)m * s->C * es, s->stream, __func__));
    const int64_t tot = m * s->C;
    if (s->P->dtype == GLX_F32)
      hipLaunchKernelGGL(scatter_bias_rows_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int64_t*)(st + b_lab), (const int32_t*)s->P->d_inv, (const float*)(st + b_lab + b_rows), m, s->C);
    else
      hipLaunchKernelGGL(scatter_bias_rows_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int64_t*)(st + b_lab), (const int32_t*)s->P->d_inv, (const double*)(st + b_lab + b_rows), m, s->C);
    GLX_HIP(hipGetLastError());
    if (nslots > 0) {
      hipLaunchKernelGGL(bias_flags_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s->stream,
                         s->plan->d_slot_row, s->slot_has_bias, nslots, (const char*)s->bias, rb);
      GLX_HIP(hipGetLastError());
    }
  }
  s->bias_set = m > 0;
  s->cur = 0;
  GLX_HIP(hipStreamSynchronize(s->stream));   // the host arrays may go
  return GLX_OK;
}

static int enqueue_iterate(glx_sweep* s, int iters) {
  int rc;
  if (s->use_graph && iters > 1) {
    const long key = (long)iters * 4 + s->cur * 2 + (s->bias_set ? 1 : 0);
    auto it = s->iter_exec.find(key);
    if (it == s->iter_exec.end()) {
      hipGraph_t graph;
      hipGraphExec_t exec;
      const int cur0 = s->cur;
      GLX_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
      rc = GLX_OK;
      for (int t = 0; t < iters && rc == GLX_OK; ++t) rc = launch_sweep(s, t, false);
      hipError_t e = hipStreamEndCapture(s->stream, &graph);
      if (rc) return rc;
      GLX_HIP(e);
      GLX_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      GLX_HIP(hipGraphDestroy(graph));
      s->iter_exec[key] = exec;
      s->cur = cur0;
      s->launches -= iters;
      it = s->iter_exec.find(key);
    }
    GLX_HIP(hipGraphLaunch(it->second, s->stream));
    s->cur ^= (iters & 1);
    s->launches += iters;
  } else {
    for (int t = 0; t < iters; ++t) {
      rc = launch_sweep(s, t, false);
      if (rc) return rc;
    }
  }
  return GLX_OK;
}

extern "C" int glx_sweep_iterate(glx_sweep* s, int iters) {
  GLX_CHECK(s && iters >= 0, GLX_EINVAL, "glx_sweep_iterate: bad argument");
  GLX_CHECK(!s->has_w, GLX_EINVAL, "glx_sweep_iterate: only for sweeps created with max_iter = 0");
  GLX_CHECK(s->n_rows == s->n_cols, GLX_EINVAL, "glx_sweep_iterate: operator must be square");
  GLX_HIP(hipSetDevice(s->P->device));
  // (enqueued, not awaited: whatever reads the state next -- glx_sweep_project, glx_sweep_fetch, another glx_sweep_iterate -- runs
  // in this sweep's stream behind it; PoissonMBO's 20 outer steps each saved a host round trip)
  return enqueue_iterate(s, iters);
}

extern "C" int glx_spmm_bias(glx_graph* A, const void* Db, const void* u_in, void* u_out, int C, int iters) {
  GLX_CHECK(A && u_in && u_out, GLX_EINVAL, "glx_spmm_bias: null argument");
  GLX_CHECK(iters >= 0, GLX_EINVAL, "glx_spmm_bias: negative iters");
  GLX_CHECK(A->n_rows == A->n_cols || iters <= 1, GLX_EINVAL, "glx_spmm_bias: iterating needs a square operator");
  glx_sweep* s = nullptr;
  int rc = glx_sweep_create(A, C, 0, 0, 0, &s);
  if (rc) return rc;
  rc = glx_sweep_set_state(s, u_in, Db);
  if (!rc) {
    for (int t = 0; t < iters && !rc; ++t) rc = launch_sweep(s, t, false);
    if (!rc && hipStreamSynchronize(s->stream) != hipSuccess) { glx_set_error("glx_spmm_bias: sync failed"); rc = GLX_EHIP; }
  }
  if (!rc) rc = glx_sweep_fetch(s, u_out);
  glx_sweep_destroy(s);
  return rc;
}

// ---- affine fixed-point iteration with a sup-norm stop ---------------------------------------
// u <- A u + b until max |u_new - u_old| <= tol: the power iteration of graph.page_rank
// (graphlearning/graph.py:1405-1410: `w = alpha*P@u + (1-alpha)*v ; err = np.max(np.absolute(w-u))`,
// `while err > tol`).  The records of both iterates are compared element by element (padding is 0
// in both); NaN bit patterns order above +inf, so a NaN difference ends the loop like `nan > tol`.
template <typename T>
__global__ __launch_bounds__(256) void absdiff_max_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t total,
                                                          unsigned long long* __restrict__ out) {
  unsigned long long m = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const T d = a[i] - b[i];                  // the subtraction in the array dtype, like numpy
    const double ad = fabs((double)d);
    const unsigned long long u = (unsigned long long)__double_as_longlong(ad);
    m = u > m ? u : m;
  }
  __shared__ unsigned long long s_m[256];
  s_m[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off && s_m[threadIdx.x + off] > s_m[threadIdx.x]) s_m[threadIdx.x] = s_m[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && s_m[0] != 0) atomicMax(out, s_m[0]);
}

extern "C" int glx_affine_iterate(glx_graph* A, const void* b, const void* u0, void* u_out, int C, double tol, int64_t max_iter,
                                  int64_t* iters_out, double* err_out) {
  GLX_CHECK(A && u0 && u_out, GLX_EINVAL, "glx_affine_iterate: null argument");
  GLX_CHECK(A->n_rows == A->n_cols, GLX_EINVAL, "glx_affine_iterate: operator must be square");
  GLX_CHECK(max_iter >= 0, GLX_EINVAL, "glx_affine_iterate: negative max_iter");
  glx_sweep* s = nullptr;
  int rc = glx_sweep_create(A, C, 0, 0, 0, &s);
  if (rc) return rc;
  rc = glx_sweep_set_state(s, u0, b);
  unsigned long long* d_err = nullptr;
  unsigned long long* h_err = nullptr;
  int64_t it = 0;
  double err = tol + 1.0;                      // graph.py:1404
  if (!rc && hipMalloc(&d_err, 8) != hipSuccess) { glx_set_error("glx_affine_iterate: hipMalloc failed"); rc = GLX_EHIP; }
  if (!rc && hipHostMalloc((void**)&h_err, 8, hipHostMallocDefault) != hipSuccess) { glx_set_error("glx_affine_iterate: hipHostMalloc failed"); rc = GLX_EHIP; }
  const int64_t total = s ? (int64_t)s->n_rows * s->L.ld : 0;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (total + 255) / 256));
  while (!rc && err > tol && it < max_iter) {
    hipError_t e = hipMemsetAsync(d_err, 0, 8, s->stream);
    if (e == hipSuccess) {
      rc = launch_sweep(s, (int)(it & 1), false);
      if (rc) break;
      if (A->dtype == GLX_F32)
        hipLaunchKernelGGL(absdiff_max_kernel<float>, dim3(grid), dim3(256), 0, s->stream, (const float*)s->buf[0], (const float*)s->buf[1], total, d_err);
      else
        hipLaunchKernelGGL(absdiff_max_kernel<double>, dim3(grid), dim3(256), 0, s->stream, (const double*)s->buf[0], (const double*)s->buf[1], total, d_err);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_err, d_err, 8, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { glx_set_error("glx_affine_iterate: %s", hipGetErrorString(e)); rc = GLX_EHIP; break; }
    err = __builtin_bit_cast(double, *h_err);
    ++it;
  }
  if (!rc) rc = glx_sweep_fetch(s, u_out);
  if (iters_out) *iters_out = it;
  if (err_out) *err_out = err;
  hipFree(d_err);
  if (h_err) hipHostFree(h_err);
  glx_sweep_destroy(s);
  return rc;
}

extern "C" int glx_poisson_sweep(glx_graph* P, const void* Db, const double* w0, const double* deg, const double* vinf,
                                 int C, int min_iter, int max_iter, void* u_out, int* T_out) {
  GLX_CHECK(P && u_out, GLX_EINVAL, "glx_poisson_sweep: null argument");
  if (max_iter == 0) {   // zero sweeps: u = 0
    memset(u_out, 0, (size_t)P->n_rows * C * (P->dtype == GLX_F32 ? 4 : 8));
    if (T_out) *T_out = 0;
    return GLX_OK;
  }
  glx_sweep* s = nullptr;
  int rc = glx_sweep_create(P, C, min_iter, max_iter, 0, &s);
  if (rc) return rc;
  rc = glx_sweep_set_problem(s, Db, w0, deg, vinf);
  if (!rc) rc = glx_sweep_run(s, T_out, nullptr);
  if (!rc) rc = glx_sweep_fetch(s, u_out);
  glx_sweep_destroy(s);
  return rc;
}

// ---- device-pointer entry points (rank-local sweeps of the vertex-partitioned solver) --------
// The caller (graphlearning_amd/dist.py) owns the buffers -- torch tensors in the record layout
// of glx_record_layout -- and the stream; nothing here synchronises.

extern "C" int glx_record_layout(int C, int dtype, int has_w, int32_t out[6]) {
  GLX_CHECK(out, GLX_EINVAL, "glx_record_layout: null output");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  out[0] = L.ld;
  out[1] = L.woff;
  out[2] = L.ld * L.esize;
  out[3] = L.G;
  out[4] = L.nvec;
  out[5] = L.esize;
  return GLX_OK;
}

static int require_caller_order(glx_graph* P, const char* who) {
  int rc = glx_graph_ensure_order(P);
  if (rc) return rc;
  GLX_CHECK(P->h_perm.empty(), GLX_EINVAL, "%s: the operator was renumbered internally; create it with glx_graph_set_order(g, NULL) "
            "to use caller-ordered device records", who);
  return GLX_OK;
}

extern "C" int glx_graph_slots(glx_graph* P, int C, int has_w, int64_t* nslots) {
  GLX_CHECK(P && nslots, GLX_EINVAL, "glx_graph_slots: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, P->dtype, has_w != 0, &L);
  if (rc) return rc;
  SellPlan* plan = nullptr;
  rc = glx_graph_plan(P, L.G, &plan);
  if (rc) return rc;
  *nslots = plan->nslices * plan->R;
  return GLX_OK;
}

extern "C" int glx_bias_flags_dev(glx_graph* P, int C, int has_w, const void* bias_rec, uint8_t* flags, void* stream) {
  GLX_CHECK(P && bias_rec && flags, GLX_EINVAL, "glx_bias_flags_dev: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, P->dtype, has_w != 0, &L);
  if (rc) return rc;
  SellPlan* plan = nullptr;
  rc = glx_graph_plan(P, L.G, &plan);
  if (rc) return rc;
  const int64_t nslots = plan->nslices * plan->R;
  if (nslots == 0) return GLX_OK;
  hipLaunchKernelGGL(bias_flags_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     plan->d_slot_row, flags, nslots, (const char*)bias_rec, L.ld * L.esize);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

extern "C" int glx_sweep_step_dev(glx_graph* P, int C, int has_w, const void* xin, void* xout, const void* bias_rec,
                                  const uint8_t* slot_flags, const double* deg, const double* vinf, void* err_next,
                                  void* stream) {
  GLX_CHECK(P && xin && xout, GLX_EINVAL, "glx_sweep_step_dev: null argument");
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  int rc = require_caller_order(P, "glx_sweep_step_dev");
  if (rc) return rc;
  rc = glx_make_layout(C, P->dtype, has_w != 0, &a.L);
  if (rc) return rc;
  SellPlan* plan = nullptr;
  rc = glx_graph_plan(P, a.L.G, &plan);
  if (rc) return rc;
  a.plan = plan;
  a.dtype = P->dtype;
  a.xin = xin;
  a.xout = xout;
  a.bias = bias_rec;
  a.slot_has_bias = bias_rec ? slot_flags : nullptr;
  a.has_w = has_w != 0;
  a.n_rows = P->n_rows;
  a.deg = deg;
  a.vinf = vinf;
  a.err_next = (unsigned long long*)err_next;   // 64 fp64-bit-pattern maxima, caller zeroes them
  GLX_CHECK(!err_next || (has_w && deg && vinf), GLX_EINVAL, "glx_sweep_step_dev: the stop test needs the stop column, deg and vinf");
  return glx_launch_spmm(a, (hipStream_t)stream);
}

extern "C" int glx_pack_records_dev(const void* dense, void* rec, int64_t n, int C, int dtype, int has_w, const double* w,
                                    void* stream) {
  GLX_CHECK(rec, GLX_EINVAL, "glx_pack_records_dev: null output");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  return glx_pack_records(dense, rec, n, L, dtype, w, (hipStream_t)stream);
}

extern "C" int glx_unpack_records_dev(const void* rec, void* dense, int64_t n, int C, int dtype, int has_w, void* stream) {
  GLX_CHECK(rec && dense, GLX_EINVAL, "glx_unpack_records_dev: null argument");
  RecLayout L;
  int rc = glx_make_layout(C, dtype, has_w != 0, &L);
  if (rc) return rc;
  return glx_unpack_records(rec, dense, n, L, dtype, (hipStream_t)stream);
}
