// Several training sets of ssl.poisson(solver='gradient_descent') as ONE sweep (SURVEY 8 f-1 for the headline kernel; reference
// graphlearning/ssl.py:292-396 runs one `_fit` per training set, each the loop of ssl.py:631-670).  The B trials' label matrices
// are B column groups of the same vertex record -- columns b C .. b C + C - 1 belong to trial b --, each group with its own fp64 stop
// value w_b = D^-1 v_b riding along (its training set decides v_0, ssl.py:639-641) and its own stop test: group b runs sweep t iff
// t < min_iter or max|deg w_b - vinf| > 1/n held after sweep t - 1, exactly the `while` of ssl.py:667.  A group that has stopped
// neither gathers nor changes (its elements are copied forward), so every trial's iterate u_T and its T are those of its own fit,
// bit for bit: the arithmetic of a column never depends on which other columns share its record.
// What is gained: the index / value stream of the operator, the launch and the prologue are paid once per B trials, and a gather
// fetches whole 128-byte lines of useful columns (10 fp64 columns alone fill 80 of 128 bytes).
#include "glx_internal.h"
#include <string.h>
#include <math.h>
#include <map>
#include <vector>
#include <algorithm>

static const int GRP_TAIL_CHUNK = 16;

struct glx_sweep_groups {
  glx_graph* P = nullptr;
  int device = 0;
  int C = 0, B = 0, min_iter = 0, max_iter = 0;
  RecLayout L;
  SellPlan* plan = nullptr;
  int64_t n = 0;
  void* buf[2] = {nullptr, nullptr};
  void* bias = nullptr;
  uint8_t* slot_has_bias = nullptr;
  double *deg = nullptr, *vinf = nullptr;
  double* w0 = nullptr;                        // [B][n] stop values of the start, caller order
  unsigned long long* err = nullptr;           // [(max_iter + 1)][B][GLX_GRP_SHARDS]
  unsigned long long* h_err = nullptr;         // page-locked mirror
  void* dense = nullptr;                       // (n, C) staging of one group
  void* stage = nullptr;                       // device staging of a training set's rows
  size_t stage_cap = 0;
  glx_work* work = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::map<unsigned, hipGraphExec_t> head_exec;   // keyed by the mask of groups in use
  glx_projector* proj = nullptr;
  int cur = 0;
  double thresh = 0.0;
  bool vectors_set = false, flags_stale = true;
  std::vector<std::vector<int64_t>> rows;      // per group: the labelled rows of its current training set (cleared by the next one)
  std::vector<double> err0;                    // per group: max|v_0 - vinf| (the test before the first sweep, min_iter = 0)
  std::vector<int> T;                          // per group: sweeps of the last run
  std::vector<char> stopped;                   // per group: did the stop test end the last run (not max_iter)?
  int64_t launches = 0;
};

static size_t grp_rec_bytes(const glx_sweep_groups* s) { return (size_t)s->L.ld * s->L.esize; }
static size_t grp_err_row(const glx_sweep_groups* s) { return (size_t)s->B * GLX_GRP_SHARDS; }

extern "C" int glx_sweep_groups_destroy(glx_sweep_groups* s) {
  if (!s) return GLX_OK;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  for (auto& kv : s->head_exec) hipGraphExecDestroy(kv.second);
  glx_pool_free(s->buf[0]);
  glx_pool_free(s->buf[1]);
  glx_pool_free(s->bias);
  glx_pool_free(s->slot_has_bias);
  glx_pool_free(s->deg);
  glx_pool_free(s->vinf);
  glx_pool_free(s->w0);
  glx_pool_free(s->err);
  if (s->h_err) hipHostFree(s->h_err);
  glx_pool_free(s->dense);
  glx_pool_free(s->stage);
  glx_projector_destroy(s->proj);
  glx_work_release(s->work);
  delete s;
  return GLX_OK;
}

extern "C" int glx_sweep_groups_create(glx_graph* P, int C, int B, int min_iter, int max_iter, glx_sweep_groups** out) {
  GLX_CHECK(P && out, GLX_EINVAL, "glx_sweep_groups_create: null argument");
  *out = nullptr;
  GLX_CHECK(P->n_rows == P->n_cols, GLX_EINVAL, "glx_sweep_groups_create: operator must be square");
  GLX_CHECK(B >= 2 && B <= 32, GLX_EINVAL, "glx_sweep_groups_create: %d groups (2 .. 32; one training set is glx_sweep_create)", B);
  GLX_CHECK(min_iter >= 0 && max_iter >= 1, GLX_EINVAL, "glx_sweep_groups_create: bad iteration bounds (%d, %d)", min_iter, max_iter);
  GLX_HIP(hipSetDevice(P->device));
  glx_sweep_groups* s = new glx_sweep_groups();
  s->P = P;
  s->device = P->device;
  s->C = C;
  s->B = B;
  s->min_iter = min_iter;
  s->max_iter = max_iter;
  s->n = P->n_rows;
  int rc = glx_make_layout_groups(C, B, P->dtype, &s->L);
  if (rc) { delete s; return rc; }
  rc = glx_graph_plan(P, s->L.G, &s->plan);
  if (rc) { delete s; return rc; }
#define SG_POOL(call) do { int rc_ = (call); if (rc_) { glx_sweep_groups_destroy(s); return rc_; } } while (0)
#define SG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { glx_set_error("%s -> %s", #call, hipGetErrorString(e_)); glx_sweep_groups_destroy(s); return GLX_EHIP; } } while (0)
  rc = glx_work_acquire(P->device, &s->work);
  if (rc) { glx_sweep_groups_destroy(s); return rc; }
  s->stream = s->work->stream;
  s->ev0 = s->work->ev[0];
  s->ev1 = s->work->ev[1];
  const size_t rb = std::max<size_t>((size_t)s->n * grp_rec_bytes(s), 128);
  SG_POOL(glx_pool_alloc(&s->buf[0], rb));
  SG_POOL(glx_pool_alloc(&s->buf[1], rb));
  SG_POOL(glx_pool_alloc(&s->bias, rb));
  SG_HIP(hipMemsetAsync(s->buf[0], 0, rb, s->stream));
  SG_HIP(hipMemsetAsync(s->buf[1], 0, rb, s->stream));
  SG_HIP(hipMemsetAsync(s->bias, 0, rb, s->stream));
  const size_t nslots = std::max<size_t>((size_t)s->plan->nslices * s->plan->R, 64);
  SG_POOL(glx_pool_alloc((void**)&s->slot_has_bias, nslots));
  SG_HIP(hipMemsetAsync(s->slot_has_bias, 0, nslots, s->stream));
  SG_POOL(glx_pool_alloc((void**)&s->deg, std::max<size_t>((size_t)s->n * 8, 64)));
  SG_POOL(glx_pool_alloc((void**)&s->vinf, std::max<size_t>((size_t)s->n * 8, 64)));
  SG_POOL(glx_pool_alloc((void**)&s->w0, std::max<size_t>((size_t)s->n * 8 * B, 64)));
  SG_HIP(hipMemsetAsync(s->w0, 0, std::max<size_t>((size_t)s->n * 8 * B, 64), s->stream));
  SG_POOL(glx_pool_alloc(&s->dense, std::max<size_t>((size_t)s->n * std::max(C * s->L.esize, 16), 64)));
  const size_t eb = (size_t)(max_iter + 1) * grp_err_row(s) * 8;
  SG_POOL(glx_pool_alloc((void**)&s->err, eb));
  SG_HIP(hipHostMalloc((void**)&s->h_err, eb, hipHostMallocDefault));
  SG_HIP(hipStreamSynchronize(s->stream));
#undef SG_HIP
#undef SG_POOL
  s->rows.resize(B);
  s->err0.assign(B, 0.0);
  s->T.assign(B, 0);
  s->stopped.assign(B, 0);
  s->thresh = 1.0 / (double)s->n;     // `> 1/n`, ssl.py:667
  *out = s;
  return GLX_OK;
}

__global__ void grp_permute_f64_kernel(const double* __restrict__ src, double* __restrict__ dst, const int32_t* __restrict__ perm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm ? perm[i] : i];
}

// the graph's own vectors of the stop test (ssl.py:642-643), indexed by record inside the kernel
extern "C" int glx_sweep_groups_set_vectors(glx_sweep_groups* s, const double* deg, const double* vinf) {
  GLX_CHECK(s && deg && vinf, GLX_EINVAL, "glx_sweep_groups_set_vectors: null argument");
  GLX_HIP(hipSetDevice(s->device));
  if (s->n == 0) { s->vectors_set = true; return GLX_OK; }
  double* tmp = (double*)s->dense;      // (n * max(C esize, 16) bytes: room for two fp64 vectors)
  const unsigned grid = (unsigned)((s->n + 255) / 256);
  GLX_UP(glx_upload(tmp, deg, (size_t)s->n * 8, s->stream, __func__));
  GLX_UP(glx_upload(tmp + s->n, vinf, (size_t)s->n * 8, s->stream, __func__));
  hipLaunchKernelGGL(grp_permute_f64_kernel, dim3(grid), dim3(256), 0, s->stream, (const double*)tmp, s->deg, (const int32_t*)s->P->d_perm, s->n);
  hipLaunchKernelGGL(grp_permute_f64_kernel, dim3(grid), dim3(256), 0, s->stream, (const double*)(tmp + s->n), s->vinf, (const int32_t*)s->P->d_perm, s->n);
  GLX_HIP(hipGetLastError());
  GLX_HIP(hipStreamSynchronize(s->stream));
  s->vectors_set = true;
  return GLX_OK;
}

// rows of group b's bias columns and start stop values: vals == nullptr clears them
template <typename T>
__global__ void grp_set_rows_kernel(char* __restrict__ bias, int rec_bytes, const int32_t* __restrict__ inv, const int64_t* __restrict__ rows,
                                    const T* __restrict__ vals, const double* __restrict__ w0_rows, double* __restrict__ w0_b, int64_t m, int C,
                                    int col0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * C) return;
  const int64_t q = i / C;
  const int c = (int)(i % C);
  const int64_t row = rows[q];
  const int32_t rec = inv ? inv[row] : (int32_t)row;
  ((T*)(bias + (size_t)rec * rec_bytes))[col0 + c] = vals ? vals[i] : (T)0;
  if (c == 0) w0_b[row] = w0_rows ? w0_rows[q] : 0.0;
}

// Training set of group b: its m labelled rows, their rows of Db = D^-1 b (m x C, state dtype) and of w0 = v_0 / deg, and
// err0 = max|v_0 - vinf| (ssl.py:620-622, 636, 639-641, 667).  m = 0 leaves the group without a problem (u stays 0).
extern "C" int glx_sweep_groups_set_problem_rows(glx_sweep_groups* s, int b, int64_t m, const int64_t* rows, const void* Db_rows,
                                                 const double* w0_rows, double err0) {
  GLX_CHECK(s, GLX_EINVAL, "glx_sweep_groups_set_problem_rows: null sweep");
  GLX_CHECK(b >= 0 && b < s->B, GLX_EINVAL, "glx_sweep_groups_set_problem_rows: group %d of %d", b, s->B);
  GLX_CHECK(s->vectors_set, GLX_EINVAL, "glx_sweep_groups_set_problem_rows: call glx_sweep_groups_set_vectors first");
  GLX_CHECK(m >= 0 && (m == 0 || (rows && Db_rows && w0_rows)), GLX_EINVAL, "glx_sweep_groups_set_problem_rows: null array");
  for (int64_t q = 0; q < m; ++q)
    GLX_CHECK(rows[q] >= 0 && rows[q] < s->n, GLX_EINVAL, "glx_sweep_groups_set_problem_rows: row %lld out of range", (long long)rows[q]);
  GLX_HIP(hipSetDevice(s->device));
  const size_t es = s->L.esize;
  const int64_t mp = (int64_t)s->rows[b].size();
  const size_t b_prev = (size_t)mp * 8, b_rows = (size_t)m * 8, b_db = ((size_t)m * s->C * es + 7) / 8 * 8, b_w = (size_t)m * 8;
  const size_t need = b_prev + b_rows + b_db + b_w + 64;
  if (s->stage_cap < need) {
    GLX_HIP(hipStreamSynchronize(s->stream));
    glx_pool_free(s->stage);
    s->stage = nullptr;
    s->stage_cap = 0;
    int rc = glx_pool_alloc(&s->stage, 2 * need);
    if (rc) return rc;
    s->stage_cap = 2 * need;
  }
  char* st = (char*)s->stage;
  const int rb = (int)grp_rec_bytes(s);
  double* w0_b = s->w0 + (size_t)b * s->n;
  const bool f32 = s->P->dtype == GLX_F32;
  if (mp > 0) {      // the previous training set of this group: its bias rows and start values back to zero
    GLX_UP(glx_upload(st, s->rows[b].data(), b_prev, s->stream, __func__));
    const int64_t tot = mp * s->C;
    if (f32)
      hipLaunchKernelGGL(grp_set_rows_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int32_t*)s->P->d_inv, (const int64_t*)st, (const float*)nullptr, (const double*)nullptr, w0_b, mp, s->C, b * s->C);
    else
      hipLaunchKernelGGL(grp_set_rows_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int32_t*)s->P->d_inv, (const int64_t*)st, (const double*)nullptr, (const double*)nullptr, w0_b, mp, s->C, b * s->C);
    GLX_HIP(hipGetLastError());
  }
  if (m > 0) {
    char* q0 = st + b_prev;
    GLX_UP(glx_upload(q0, rows, b_rows, s->stream, __func__));
    GLX_UP(glx_upload(q0 + b_rows, Db_rows, (size_t)m * s->C * es, s->stream, __func__));
    GLX_UP(glx_upload(q0 + b_rows + b_db, w0_rows, b_w, s->stream, __func__));
    const int64_t tot = m * s->C;
    if (f32)
      hipLaunchKernelGGL(grp_set_rows_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int32_t*)s->P->d_inv, (const int64_t*)q0, (const float*)(q0 + b_rows), (const double*)(q0 + b_rows + b_db), w0_b, m,
                         s->C, b * s->C);
    else
      hipLaunchKernelGGL(grp_set_rows_kernel<double>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->bias, rb,
                         (const int32_t*)s->P->d_inv, (const int64_t*)q0, (const double*)(q0 + b_rows), (const double*)(q0 + b_rows + b_db), w0_b, m,
                         s->C, b * s->C);
    GLX_HIP(hipGetLastError());
  }
  GLX_HIP(hipStreamSynchronize(s->stream));      // the host arrays may go; the staging is reused by the next call
  s->rows[b].assign(rows, rows + m);
  s->err0[b] = err0;
  s->flags_stale = true;
  return GLX_OK;
}

// per-slot flag: does the row's bias record hold a nonzero in ANY group?  (-0.0 counts as zero)
__global__ void grp_bias_flags_kernel(const int32_t* __restrict__ slot_row, uint8_t* __restrict__ flags, int64_t nslots,
                                      const char* __restrict__ bias, int rec_bytes) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots) return;
  const int row = slot_row[slot];
  uint8_t f = 0;
  if (row >= 0) {
    const unsigned long long* r = (const unsigned long long*)(bias + (size_t)row * rec_bytes);
    for (int i = 0; i < rec_bytes / 8; ++i) f |= (r[i] << 1) != 0;
  }
  flags[slot] = f;
}

// the start of every run: u = 0 (ssl.py:645), stop value of group b = w0_b (record order through perm)
__global__ void grp_init_stop_kernel(char* __restrict__ rec, int rec_bytes, int woff, const double* __restrict__ w0, const int32_t* __restrict__ perm,
                                     int64_t n, int B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * B) return;
  const int64_t row = i / B;
  const int b = (int)(i % B);
  *(double*)(rec + (size_t)row * rec_bytes + woff + 8 * b) = w0[(size_t)b * n + (perm ? perm[row] : row)];
}

static int grp_launch_sweep(glx_sweep_groups* s, int t, unsigned used_mask) {
  SweepArgs a;
  memset(&a, 0, sizeof(a));
  a.plan = s->plan;
  a.L = s->L;
  a.dtype = s->P->dtype;
  a.xin = s->buf[s->cur];
  a.xout = s->buf[s->cur ^ 1];
  a.bias = s->bias;
  a.slot_has_bias = s->slot_has_bias;
  a.has_w = true;
  a.n_rows = s->n;
  a.deg = s->deg;
  a.vinf = s->vinf;
  a.thresh = s->thresh;
  a.ngroups = s->B;
  a.used_mask = used_mask;
  a.err_prev = t >= s->min_iter ? s->err + (size_t)t * grp_err_row(s) : nullptr;
  a.err_next = t + 1 >= s->min_iter ? s->err + (size_t)(t + 1) * grp_err_row(s) : nullptr;
  int rc = glx_launch_spmm(a, s->stream);
  if (rc) return rc;
  s->cur ^= 1;
  s->launches++;
  return GLX_OK;
}

// reset + the min_iter unconditional sweeps: identical for every run with the same groups in use -> captured once
static int grp_enqueue_head(glx_sweep_groups* s, unsigned used_mask) {
  const int head = std::min(s->min_iter, s->max_iter);
  const size_t er = grp_err_row(s);
  // (kernels, not memset nodes: this sequence is captured and replayed -- glx_zero_async, glx_internal.h)
  { int rz = glx_zero_async(s->err + (size_t)head * er, er * 8, s->stream); if (rz) return rz; }
  if (s->min_iter == 0)      // the test in front of the first sweep: err0 of every group, written to the mirror by the caller
    GLX_HIP(hipMemcpyAsync(s->err, s->h_err, er * 8, hipMemcpyHostToDevice, s->stream));
  s->cur = 0;
  { int rz = glx_zero_async(s->buf[0], (size_t)s->n * grp_rec_bytes(s), s->stream); if (rz) return rz; }
  if (s->n > 0) {
    const int64_t tot = s->n * s->B;
    hipLaunchKernelGGL(grp_init_stop_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s->stream, (char*)s->buf[0], (int)grp_rec_bytes(s),
                       s->L.woff, (const double*)s->w0, (const int32_t*)s->P->d_perm, s->n, s->B);
    GLX_HIP(hipGetLastError());
  }
  for (int t = 0; t < head; ++t) {
    int rc = grp_launch_sweep(s, t, used_mask);
    if (rc) return rc;
  }
  return GLX_OK;
}

static double grp_err_value(const glx_sweep_groups* s, int t, int b) {
  const unsigned long long* r = s->h_err + (size_t)t * grp_err_row(s) + (size_t)b * GLX_GRP_SHARDS;
  unsigned long long m = 0;
  for (int k = 0; k < GLX_GRP_SHARDS; ++k) m = std::max(m, r[k]);
  union { double d; unsigned long long u; } cv;
  cv.u = m;
  return cv.d;
}

// Run the groups 0 .. used - 1 (the others stay idle).  T_out[b] = sweeps of group b, exactly the T of its own fit.
extern "C" int glx_sweep_groups_run(glx_sweep_groups* s, int used, int* T_out, float* device_ms_out) {
  GLX_CHECK(s && T_out, GLX_EINVAL, "glx_sweep_groups_run: null argument");
  GLX_CHECK(used >= 1 && used <= s->B, GLX_EINVAL, "glx_sweep_groups_run: %d of %d groups", used, s->B);
  GLX_HIP(hipSetDevice(s->device));
  const unsigned used_mask = used >= 32 ? 0xffffffffu : ((1u << used) - 1u);
  const int head = std::min(s->min_iter, s->max_iter);
  const size_t er = grp_err_row(s);
  if (s->flags_stale) {
    const int64_t nslots = s->plan->nslices * s->plan->R;
    if (nslots > 0) {
      hipLaunchKernelGGL(grp_bias_flags_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s->stream, (const int32_t*)s->plan->d_slot_row,
                         s->slot_has_bias, nslots, (const char*)s->bias, (int)grp_rec_bytes(s));
      GLX_HIP(hipGetLastError());
    }
    s->flags_stale = false;
  }
  GLX_HIP(hipEventRecord(s->ev0, s->stream));
  if (s->min_iter == 0) {
    memset(s->h_err, 0, er * 8);
    for (int b = 0; b < s->B; ++b) {
      union { double d; unsigned long long u; } cv;
      cv.d = s->err0[b];
      if (cv.d != cv.d) cv.u = 0x7ff8000000000000ull;
      s->h_err[(size_t)b * GLX_GRP_SHARDS] = cv.u;
    }
  }
  auto it = s->head_exec.find(used_mask);
  if (it == s->head_exec.end()) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    const int64_t l0 = s->launches;
    GLX_HIP(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int rc = grp_enqueue_head(s, used_mask);
    hipError_t e = hipStreamEndCapture(s->stream, &graph);
    s->launches = l0;
    if (rc) return rc;
    GLX_HIP(e);
    GLX_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    GLX_HIP(hipGraphDestroy(graph));
    it = s->head_exec.emplace(used_mask, exec).first;
  }
  GLX_HIP(hipGraphLaunch(it->second, s->stream));
  s->cur = head & 1;
  s->launches += head;
  // tail: sweeps in chunks; every kernel decides per group from the previous sweep's maxima (a stopped group is copied forward, a
  // launch behind the last group's stop returns at once), the host only reads the maxima back
  std::vector<int> T(s->B, -1);        // -1: still running
  std::vector<char> stopped(s->B, 0);
  for (int b = used; b < s->B; ++b) T[b] = 0;
  auto running = [&]() { for (int b = 0; b < used; ++b) if (T[b] < 0) return true; return false; };
  auto test = [&](int t) {             // the `while` of ssl.py:667 at sweep count t, for the groups still running
    for (int b = 0; b < used; ++b)
      if (T[b] < 0 && !(grp_err_value(s, t, b) > s->thresh)) { T[b] = t; stopped[b] = 1; }
  };
  int t = head;
  while (t < s->max_iter && running()) {
    GLX_HIP(hipMemcpyAsync(s->h_err + (size_t)t * er, s->err + (size_t)t * er, er * 8, hipMemcpyDeviceToHost, s->stream));
    GLX_HIP(hipStreamSynchronize(s->stream));
    test(t);
    if (!running()) break;
    const int end = std::min(s->max_iter, t + GRP_TAIL_CHUNK);
    GLX_HIP(hipMemsetAsync(s->err + (size_t)(t + 1) * er, 0, (size_t)(end - t) * er * 8, s->stream));
    const int t0 = t;
    for (; t < end; ++t) {
      int rc = grp_launch_sweep(s, t, used_mask);
      if (rc) return rc;
    }
    GLX_HIP(hipMemcpyAsync(s->h_err + (size_t)(t0 + 1) * er, s->err + (size_t)(t0 + 1) * er, (size_t)(end - t0) * er * 8, hipMemcpyDeviceToHost,
                           s->stream));
    GLX_HIP(hipStreamSynchronize(s->stream));
    for (int q = t0 + 1; q < end; ++q) test(q);
  }
  int T_max = head;
  for (int b = 0; b < used; ++b) {
    if (T[b] < 0) T[b] = s->max_iter;      // the value at t = max_iter is never compared (ssl.py:667)
    T_max = std::max(T_max, T[b]);
  }
  // sweep T_max - 1 was the last one that wrote anything (later launches return at once): every group's u_T sits in its target
  s->cur = T_max & 1;
  GLX_HIP(hipEventRecord(s->ev1, s->stream));
  GLX_HIP(hipStreamSynchronize(s->stream));
  if (device_ms_out) GLX_HIP(hipEventElapsedTime(device_ms_out, s->ev0, s->ev1));
  for (int b = 0; b < s->B; ++b) T_out[b] = T[b];
  s->T = T;
  s->stopped = stopped;
  return GLX_OK;
}

// The stop values group b's last run compared with 1/n (see glx_sweep_stop_values): t = *first .. *first + *count - 1.
extern "C" int glx_sweep_groups_stop_values(const glx_sweep_groups* s, int b, int64_t cap, double* vals, int* first, int* count) {
  GLX_CHECK(s && first && count, GLX_EINVAL, "glx_sweep_groups_stop_values: null argument");
  GLX_CHECK(b >= 0 && b < s->B, GLX_EINVAL, "glx_sweep_groups_stop_values: group %d of %d", b, s->B);
  const int head = std::min(s->min_iter, s->max_iter);
  *first = head;
  *count = std::max(0, s->T[b] - head + (s->stopped[b] ? 1 : 0));
  if (!vals) return GLX_OK;
  GLX_CHECK(cap >= *count, GLX_EINVAL, "glx_sweep_groups_stop_values: %d values, room for %lld", *count, (long long)cap);
  for (int i = 0; i < *count; ++i) vals[i] = grp_err_value(s, head + i, b);
  return GLX_OK;
}

template <typename T>
__global__ void grp_unpack_kernel(const T* __restrict__ rec, T* __restrict__ dense, int64_t n, int C, int ld, int col0,
                                  const int32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  const int64_t row = i / C;
  const int c = (int)(i % C);
  dense[(perm ? (int64_t)perm[row] : row) * C + c] = rec[row * ld + col0 + c];
}

static int grp_unpack(glx_sweep_groups* s, int b, void* dense_dev) {
  const int64_t tot = s->n * s->C;
  if (tot == 0) return GLX_OK;
  const unsigned grid = (unsigned)((tot + 255) / 256);
  if (s->P->dtype == GLX_F32)
    hipLaunchKernelGGL(grp_unpack_kernel<float>, dim3(grid), dim3(256), 0, s->stream, (const float*)s->buf[s->cur], (float*)dense_dev, s->n, s->C,
                       s->L.ld, b * s->C, (const int32_t*)s->P->d_perm);
  else
    hipLaunchKernelGGL(grp_unpack_kernel<double>, dim3(grid), dim3(256), 0, s->stream, (const double*)s->buf[s->cur], (double*)dense_dev, s->n, s->C,
                       s->L.ld, b * s->C, (const int32_t*)s->P->d_perm);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// group b's (n, C) iterate of the last run, caller order, state dtype
extern "C" int glx_sweep_groups_fetch(glx_sweep_groups* s, int b, void* u_out) {
  GLX_CHECK(s && u_out, GLX_EINVAL, "glx_sweep_groups_fetch: null argument");
  GLX_CHECK(b >= 0 && b < s->B, GLX_EINVAL, "glx_sweep_groups_fetch: group %d of %d", b, s->B);
  GLX_HIP(hipSetDevice(s->device));
  int rc = grp_unpack(s, b, s->dense);
  if (rc) return rc;
  GLX_UP(glx_download(u_out, s->dense, (size_t)s->n * s->C * s->L.esize, s->stream, __func__));
  GLX_HIP(hipStreamSynchronize(s->stream));
  return GLX_OK;
}

// ssl.predict / ssl.volume_label_projection (ssl.py:230-266, 172-209) on group b's iterate without a host round trip
extern "C" int glx_sweep_groups_project(glx_sweep_groups* s, int b, const double* priors, double* weights_inout, int64_t* labels_out,
                                        double* err_out, int* steps_out, int max_steps, int similarity) {
  GLX_CHECK(s && weights_inout, GLX_EINVAL, "glx_sweep_groups_project: null argument");
  GLX_CHECK(b >= 0 && b < s->B, GLX_EINVAL, "glx_sweep_groups_project: group %d of %d", b, s->B);
  GLX_HIP(hipSetDevice(s->device));
  const int dtype = s->P->dtype;
  void* prob = s->dense;
  int rc;
  if (dtype == GLX_F64) {
    double* scores = nullptr;
    rc = glx_project_scores(&s->proj, s->n, s->C, &scores);
    if (rc) return rc;
    prob = scores;
  }
  rc = grp_unpack(s, b, prob);
  if (rc) return rc;
  const long long* d_labels = nullptr;
  const std::function<int(bool)> hook = [&](bool after) -> int {
    if (!after && labels_out) GLX_UP(glx_download(labels_out, d_labels, (size_t)s->n * 8, s->stream, __func__));
    return GLX_OK;
  };
  return glx_project_device(&s->proj, prob, dtype, s->n, s->C, priors, weights_inout, err_out, steps_out, max_steps, similarity, s->stream,
                            &d_labels, &hook);
}

extern "C" int glx_sweep_groups_launches(const glx_sweep_groups* s, int64_t* n) {
  GLX_CHECK(s && n, GLX_EINVAL, "glx_sweep_groups_launches: null argument");
  *n = s->launches;
  return GLX_OK;
}
