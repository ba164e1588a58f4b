// ssl.predict (reference graphlearning/ssl.py:230-266) and ssl.volume_label_projection
// (ssl.py:172-209) on device.  All arithmetic is elementwise IEEE fp64 with contraction
// off and integer class counts, so labels and weights are bit-identical to numpy's.
#include "glx_internal.h"
#include <functional>
#include <algorithm>
#include <string.h>
#include <map>
#include <mutex>

static const int PROJ_CHUNK = 32;

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int lo = __shfl_xor(__double2loint(v), off), hi = __shfl_xor(__double2hiint(v), off);
    const double o = __hiloint2double(hi, lo);
    v = (o < v || o != o) ? o : v;   // NaN propagates like np.min
  }
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int lo = __shfl_xor(__double2loint(v), off), hi = __shfl_xor(__double2hiint(v), off);
    const double o = __hiloint2double(hi, lo);
    v = (o > v || o != o) ? o : v;
  }
  return v;
}

// min / max of prob: per-block partial results, and the block that finishes last reduces them into mm[0..1] (one launch; minimum and
// maximum with NaN propagation do not depend on the order).  ticket: a zeroed counter, back at zero afterwards.
__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ a, int64_t total, double* __restrict__ bmin,
                                                     double* __restrict__ bmax, unsigned* __restrict__ ticket, double* __restrict__ mm) {
  double mn = __longlong_as_double(0x7ff0000000000000ll), mx = -mn;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const double v = a[i];
    mn = (v < mn || v != v) ? v : mn;
    mx = (v > mx || v != v) ? v : mx;
  }
  mn = wave_min_d(mn);
  mx = wave_max_d(mx);
  __shared__ double s_mn[4], s_mx[4];
  if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = (s_mn[w] < mn || s_mn[w] != s_mn[w]) ? s_mn[w] : mn;
      mx = (s_mx[w] > mx || s_mx[w] != s_mx[w]) ? s_mx[w] : mx;
    }
    glx_agent_store(&bmin[blockIdx.x], mn);
    glx_agent_store(&bmax[blockIdx.x], mx);
  }
  if (!glx_arrive_last(ticket, gridDim.x, s_mn)) return;
  mn = __longlong_as_double(0x7ff0000000000000ll);
  mx = -mn;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) {
    const double lo = glx_load_part<true>(&bmin[b]), hi = glx_load_part<true>(&bmax[b]);
    mn = (lo < mn || lo != lo) ? lo : mn;
    mx = (hi > mx || hi != hi) ? hi : mx;
  }
  mn = wave_min_d(mn);
  mx = wave_max_d(mx);
  if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = (s_mn[w] < mn || s_mn[w] != s_mn[w]) ? s_mn[w] : mn;
      mx = (s_mx[w] > mx || s_mx[w] != s_mx[w]) ? s_mx[w] : mx;
    }
    mm[0] = mn;
    mm[1] = mx;
  }
}

// scores = (prob - min) / max(prob - min)                                   (ssl.py:256-257)
// F32: prob came from a float32 state (the reference's use_cuda branch keeps self.prob float32, so the
// subtraction and the division round in float32, ssl.py:256-257; only the product with the fp64 class
// weights is wider) -- the values in `a` are exact float32 numbers and stay so.
template <bool F32>
__global__ __launch_bounds__(256) void scores_kernel(double* __restrict__ a, int64_t total, const double* __restrict__ mm) {
#pragma clang fp contract(off)
  if constexpr (F32) {
    const float mn = (float)mm[0];
    const float den = (float)mm[1] - mn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const float s = (float)a[i] - mn;
      a[i] = (double)(s / den);
    }
  } else {
    const double mn = mm[0];
    const double den = mm[1] - mn;   // max(prob - min) == fl(max - min): rounding is monotone
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const double s = a[i] - mn;
      a[i] = s / den;
    }
  }
}

struct ProjState {
  double* w;        // [C]
  double* priors;   // [C]
  long long* counts;  // [C]
  long long* cnt_part;   // [blocks][C] per-workgroup class counts of the current step
  double* err;      // [1]
  int* steps;       // [1]
  int* done;        // [1]
};

// labels = argmax_c scores[i,c]*w[c] (first index wins, NaN wins like np.argmax); class histogram
// do_hist = 2: the block that finishes last also takes the gradient step on the class weights (proj_update below, ssl.py:198-205) --
// one launch per step of the projection instead of two
__device__ void proj_update_step(ProjState st, int64_t n, int C, double dt, int max_steps);

__global__ __launch_bounds__(256) void argmax_hist_kernel(const double* __restrict__ scores, int64_t n, int C, ProjState st,
                                                          long long* __restrict__ labels, int similarity, int check_done,
                                                          int do_hist, double dt, int max_steps) {
#pragma clang fp contract(off)
  if (check_done && *st.done) return;
  extern __shared__ long long s_cnt[];
  for (int c = threadIdx.x; c < C; c += 256) s_cnt[c] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double* row = scores + i * C;
    double best = row[0] * st.w[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const double v = row[c] * st.w[c];
      const bool better = similarity ? (v > best) : (v < best);
      if ((better || v != v) && !(best != best)) { best = v; bi = c; }
    }
    labels[i] = bi;
    if (do_hist) atomicAdd((unsigned long long*)&s_cnt[bi], 1ull);
  }
  if (do_hist) {
    // The workgroup's class counts leave as one row of plain (agent-scope) stores and the workgroup that arrives last adds the rows:
    // C atomics per workgroup on the same C addresses cost 22 of this kernel's 26 us at 274 workgroups (profiles/r04_mbo_step.txt) --
    // same-address atomics from eight XCDs serialise at ~7 ns apiece.
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      __hip_atomic_store((unsigned long long*)&st.cnt_part[(size_t)blockIdx.x * C + c], (unsigned long long)s_cnt[c], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (!glx_arrive_last((unsigned*)(st.done + 1), gridDim.x, (double*)s_cnt)) return;
    for (int c = threadIdx.x; c < C; c += 256) s_cnt[c] = 0;
    __syncthreads();
    for (int64_t q = threadIdx.x; q < (int64_t)gridDim.x * C; q += 256) {
      const unsigned long long v = __hip_atomic_load((const unsigned long long*)&st.cnt_part[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v) atomicAdd((unsigned long long*)&s_cnt[q % C], v);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256)
      __hip_atomic_store((unsigned long long*)&st.counts[c], (unsigned long long)s_cnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (do_hist == 2 && threadIdx.x == 0) proj_update_step(st, n, C, dt, max_steps);
  }
}

// one gradient step on the class weights                                     (ssl.py:198-205)
__device__ void proj_update_step(ProjState st, int64_t n, int C, double dt, int max_steps) {
#pragma clang fp contract(off)
  if (*st.done) return;
  double err = 0.0;
  for (int c = 0; c < C; ++c) {
    const double size = (double)__hip_atomic_load((unsigned long long*)&st.counts[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (double)n;   // np.mean of a 0/1 column
    const double grad = size - st.priors[c];
    const double ag = fabs(grad);
    if (ag > err || ag != ag) err = ag;                      // np.max propagates NaN
    const double t = dt * grad;
    st.w[c] = st.w[c] + t;
    st.counts[c] = 0;
  }
  const double w0 = st.w[0];
  for (int c = 0; c < C; ++c) st.w[c] = st.w[c] / w0;
  *st.err = err;
  *st.steps += 1;
  if (!(err > 1e-3) || *st.steps >= max_steps) *st.done = 1;
}

// Device state of one decision in ONE block of 8-byte words, so that one upload sets it up and one download reads it back:
//   [0, C) class weights w | [C] err | [C+1] steps | [C+2] done, hist ticket (2 ints) | [C+3] min/max ticket | [C+4, 2C+4) priors |
//   [2C+4, 3C+4) class counts | [3C+4, 3C+6) min, max of prob
struct ProjBufs {
  double *scores = nullptr, *bmin = nullptr, *bmax = nullptr, *state = nullptr;
  long long* labels = nullptr;
  long long* cnt_part = nullptr;  // [PROJ_ROW_BLOCKS][C]
  hipEvent_t ev = nullptr;        // behind the copy a host look waits for (work enqueued after it does not hold the look up)
  double* h_image = nullptr;      // page-locked: the state's initial image going up / [w | err | steps | done] coming back
  float* stage32 = nullptr;       // float32 input of the one-shot entry point, before widening
  size_t stage_cap = 0;
  hipStream_t stream = nullptr;   // owned only by the one-shot entry point
  int64_t cap_n = 0;
  int cap_C = 0;
  double* w() const { return state; }
  double* err() const { return state + cap_C; }
  int* steps() const { return (int*)(state + cap_C + 1); }
  int* done() const { return (int*)(state + cap_C + 2); }
  unsigned* mm_ticket() const { return (unsigned*)(state + cap_C + 3); }
  double* priors() const { return state + cap_C + 4; }
  long long* counts() const { return (long long*)(state + 2 * cap_C + 4); }
  double* mm() const { return state + 3 * cap_C + 4; }
  size_t image_words() const { return (size_t)3 * cap_C + 4; }
  void release() {
    // (size-class pools of graph.hip: hipFree of the score block cost 0.23 ms per model, hipHostFree 0.25 -- the pools' contract is that
    // nothing in flight uses a block handed back, which hipFree enforced by waiting for the device: wait for this object's stream)
    if (stream) hipStreamSynchronize(stream);
    glx_pool_free(scores); glx_pool_free(bmin); glx_pool_free(bmax); glx_pool_free(state); glx_pool_free(labels); glx_pool_free(cnt_part);
    cnt_part = nullptr;
    if (ev) hipEventDestroy(ev);
    ev = nullptr;
    glx_pinned_free(h_image);
    scores = bmin = bmax = state = nullptr;
    labels = nullptr;
    h_image = nullptr;
    cap_n = 0;
    cap_C = 0;
  }
  ~ProjBufs() {
    release();
    hipFree(stage32);
    if (stream) hipStreamDestroy(stream);
  }
};

// (256 workgroups: every one draws a ticket on ONE address at the end -- 1024 of them were most of the min / max kernel's 17 us)
static int proj_blocks(int64_t total) { return (int)std::min<int64_t>((total + 255) / 256, 256); }
static const int PROJ_ROW_BLOCKS = 2048;   // most workgroups of an argmax pass (sizes cnt_part)

#define PJ_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
static int proj_alloc(ProjBufs& b, int64_t n, int C) {
  if (b.cap_n >= n && b.cap_C == C && b.scores) return GLX_OK;      // (the state block's layout depends on C)
  const int64_t keep_n = std::max<int64_t>(n, b.cap_n);
  b.release();
  const int64_t total = keep_n * C;
  const int nb = proj_blocks(total);
  PJ_POOL(glx_pool_alloc((void**)&b.scores, total * 8));
  PJ_POOL(glx_pool_alloc((void**)&b.bmin, nb * 8));
  PJ_POOL(glx_pool_alloc((void**)&b.bmax, nb * 8));
  PJ_POOL(glx_pool_alloc((void**)&b.state, ((size_t)3 * C + 6) * 8));
  PJ_POOL(glx_pool_alloc((void**)&b.labels, keep_n * 8));
  PJ_POOL(glx_pool_alloc((void**)&b.cnt_part, (size_t)PROJ_ROW_BLOCKS * C * 8));
  GLX_HIP(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  PJ_POOL(glx_pinned_alloc((void**)&b.h_image, ((size_t)3 * C + 6) * 8));
  b.cap_n = keep_n;
  b.cap_C = C;
  return GLX_OK;
}

// the decision itself: b.scores holds prob (n, C) fp64 on the device (overwritten by the scores);
// on return b.labels holds the labels and weights_inout the updated class weights
// `hook` (optional) is called behind every launch of the FINAL decision: hook(false) enqueues what the caller wants back with the
// decision (it is waited for), hook(true) what merely follows it (enqueued behind the event the host look waits on, so it runs on
// while the host goes on: PoissonMBO's one-hot state and its next chunk of heat sweeps).  A decision taken speculatively in the
// first look and found unfinished is taken again later -- the hook runs again and must be idempotent in that sense.
static int proj_core(ProjBufs& b, hipStream_t st, int64_t n, int C, const double* priors, double* weights_inout, double* err_out,
                     int* steps_out, int max_steps, int similarity, bool f32 = false, const std::function<int(bool)>* hook = nullptr) {
  const int64_t total = n * C;
  const int nb = proj_blocks(total);
  const int nbr = (int)std::min<int64_t>((n + 255) / 256, PROJ_ROW_BLOCKS);
  // one upload sets the whole state: weights, priors, zeroed err / steps / done / tickets / class counts
  memset(b.h_image, 0, b.image_words() * 8);
  memcpy(b.h_image, weights_inout, (size_t)C * 8);
  if (priors) memcpy(b.h_image + C + 4, priors, (size_t)C * 8);
  GLX_HIP(hipMemcpyAsync(b.state, b.h_image, b.image_words() * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, st, (const double*)b.scores, total, b.bmin, b.bmax, b.mm_ticket(), b.mm());
  GLX_HIP(hipGetLastError());
  if (f32)
    hipLaunchKernelGGL(scores_kernel<true>, dim3(nb), dim3(256), 0, st, b.scores, total, (const double*)b.mm());
  else
    hipLaunchKernelGGL(scores_kernel<false>, dim3(nb), dim3(256), 0, st, b.scores, total, (const double*)b.mm());
  GLX_HIP(hipGetLastError());
  ProjState ps;
  ps.w = b.w();
  ps.priors = b.priors();
  ps.counts = b.counts();
  ps.cnt_part = b.cnt_part;
  ps.err = b.err();
  ps.steps = b.steps();
  ps.done = b.done();
  const double dt = similarity ? -0.1 : 0.1;   // ssl.py:195-197
  const size_t shm = (size_t)C * 8;
  int steps = 0;
  double err = 1.0;   // ssl.py:201
  // what comes back lands in the page-locked image (behind the words that went up: the upload has long finished by then)
  double* h_back = b.h_image;
  bool decided = false;          // b.labels and h_back hold the final decision and state
  if (max_steps > 0) {
    int done = 0;
    // steps per host look: 2, 4, 8 ... PROJ_CHUNK.  Most decisions need one or two steps (class sizes that already match the
    // priors: every thresholding of PoissonMBO at config 5), and every launch past the stopping step still costs its ~4 us:
    // 32 per look made a one-step projection 0.22 ms; the 10^4-step projections reach the full chunk after four looks.
    // The FIRST look also carries the final decision and the whole state: a projection that stops within two steps is one host
    // round trip (it was two: ~20 us of idle device each, 21 times per PoissonMBO fit).
    int chunk = 2;
    bool first = true;
    while (!done) {
      for (int q = 0; q < chunk; ++q) {
        hipLaunchKernelGGL(argmax_hist_kernel, dim3(nbr), dim3(256), shm, st, (const double*)b.scores, n, C, ps, b.labels, similarity, 1, 2, dt, max_steps);
        GLX_HIP(hipGetLastError());
      }
      if (first) {
        hipLaunchKernelGGL(argmax_hist_kernel, dim3(nbr), dim3(256), shm, st, (const double*)b.scores, n, C, ps, b.labels, similarity, 0, 0, dt, max_steps);
        GLX_HIP(hipGetLastError());
        GLX_HIP(hipMemcpyAsync(h_back, b.state, (size_t)(C + 3) * 8, hipMemcpyDeviceToHost, st));      // [w | err | steps | done]
        if (hook) { int rh = (*hook)(false); if (rh) return rh; }
        GLX_HIP(hipEventRecord(b.ev, st));
        if (hook) { int rh = (*hook)(true); if (rh) return rh; }
        GLX_HIP(hipEventSynchronize(b.ev));
      } else {
        GLX_HIP(hipMemcpyAsync(h_back + C + 2, b.done(), 4, hipMemcpyDeviceToHost, st));
        GLX_HIP(hipStreamSynchronize(st));
      }
      done = *(const int*)(h_back + C + 2);
      decided = first && done;
      first = false;
      chunk = std::min(PROJ_CHUNK, chunk * 2);
    }
  }
  if (!decided) {
    // final predict with the (updated) weights                                 (ssl.py:209)
    hipLaunchKernelGGL(argmax_hist_kernel, dim3(nbr), dim3(256), shm, st, (const double*)b.scores, n, C, ps, b.labels, similarity, 0, 0, dt, max_steps);
    GLX_HIP(hipGetLastError());
    GLX_HIP(hipMemcpyAsync(h_back, b.state, (size_t)(C + 2) * 8, hipMemcpyDeviceToHost, st));      // [w | err | steps]
    if (hook) { int rh = (*hook)(false); if (rh) return rh; }
    GLX_HIP(hipEventRecord(b.ev, st));
    if (hook) { int rh = (*hook)(true); if (rh) return rh; }
    GLX_HIP(hipEventSynchronize(b.ev));
  }
  memcpy(weights_inout, h_back, (size_t)C * 8);
  if (max_steps > 0) {
    err = h_back[C];
    steps = *(const int*)(h_back + C + 1);
  }
  if (err_out) *err_out = err;
  if (steps_out) *steps_out = steps;
  return GLX_OK;
}

static int argmax_project_any(const void* prob, int prob_dtype, int64_t n, int C, const double* priors, double* weights_inout,
                              int64_t* labels_out, double* err_out, int* steps_out, int max_steps, int similarity, int device);

extern "C" int glx_argmax_project(const double* prob, int64_t n, int C, const double* priors, double* weights_inout,
                                  int64_t* labels_out, double* err_out, int* steps_out, int max_steps, int similarity,
                                  int device) {
  return argmax_project_any(prob, GLX_F64, n, C, priors, weights_inout, labels_out, err_out, steps_out, max_steps, similarity, device);
}

extern "C" int glx_argmax_project_t(const void* prob, int prob_dtype, int64_t n, int C, const double* priors, double* weights_inout,
                                    int64_t* labels_out, double* err_out, int* steps_out, int max_steps, int similarity,
                                    int device) {
  GLX_CHECK(prob_dtype == GLX_F32 || prob_dtype == GLX_F64, GLX_EINVAL, "glx_argmax_project_t: bad dtype %d", prob_dtype);
  return argmax_project_any(prob, prob_dtype, n, C, priors, weights_inout, labels_out, err_out, steps_out, max_steps, similarity, device);
}

template <typename T>
__global__ __launch_bounds__(256) void to_f64_kernel(const T* __restrict__ src, double* __restrict__ dst, int64_t total);

static int argmax_project_any(const void* prob, int prob_dtype, int64_t n, int C, const double* priors, double* weights_inout,
                              int64_t* labels_out, double* err_out, int* steps_out, int max_steps, int similarity, int device) {
  GLX_CHECK(prob && weights_inout && labels_out, GLX_EINVAL, "glx_argmax_project: null argument");
  GLX_CHECK(n >= 1 && C >= 1, GLX_EINVAL, "glx_argmax_project: empty input (n=%lld, C=%d)", (long long)n, C);
  GLX_CHECK(max_steps == 0 || priors, GLX_EINVAL, "glx_argmax_project: projection needs priors");
  GLX_CHECK(C <= 4096, GLX_EUNSUPPORTED, "glx_argmax_project: C=%d too large", C);
  GLX_HIP(hipSetDevice(device));
  // work buffers and stream are kept per device between calls (ssl_trials calls predict once per
  // trial: a dozen hipMalloc/hipFree pairs cost more than the decision itself); intentionally never
  // destroyed -- a static destructor would run after the HIP runtime has shut down
  static std::mutex mu;
  static std::map<int, ProjBufs*>* cache = new std::map<int, ProjBufs*>();
  std::lock_guard<std::mutex> lock(mu);
  ProjBufs*& slot = (*cache)[device];
  if (!slot) {
    slot = new ProjBufs();
    GLX_HIP(hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking));
  }
  ProjBufs& b = *slot;
  hipStream_t st = b.stream;
  int rc = proj_alloc(b, n, C);
  if (rc) return rc;
  if (prob_dtype == GLX_F32) {
    if (b.stage_cap < (size_t)n * C) {
      hipFree(b.stage32);
      b.stage32 = nullptr;
      b.stage_cap = 0;
      GLX_HIP(hipMalloc(&b.stage32, (size_t)n * C * 4));
      b.stage_cap = (size_t)n * C;
    }
    GLX_UP(glx_upload(b.stage32, prob, (size_t)n * C * 4, st, __func__));
    hipLaunchKernelGGL(to_f64_kernel<float>, dim3((unsigned)(((size_t)n * C + 255) / 256)), dim3(256), 0, st, (const float*)b.stage32, b.scores,
                       (int64_t)n * C);
    GLX_HIP(hipGetLastError());
  } else {
    GLX_UP(glx_upload(b.scores, prob, (size_t)n * C * 8, st, __func__));
  }
  rc = proj_core(b, st, n, C, priors, weights_inout, err_out, steps_out, max_steps, similarity, prob_dtype == GLX_F32);
  if (rc) return rc;
  GLX_UP(glx_download(labels_out, b.labels, n * 8, st, __func__));
  GLX_HIP(hipStreamSynchronize(st));
  return GLX_OK;
}

// ---- the same decision on a device-resident (n, C) array (glx_sweep_project) --------------------
struct glx_projector { ProjBufs b; };

template <typename T>
__global__ __launch_bounds__(256) void to_f64_kernel(const T* __restrict__ src, double* __restrict__ dst, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) dst[i] = (double)src[i];
}

// dense[i, c] = (labels[i] == c), utils.labels_to_onehot (utils.py:536-572) for labels in 0..C-1
template <typename T>
__global__ __launch_bounds__(256) void onehot_kernel(const long long* __restrict__ labels, T* __restrict__ dense, int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * C) return;
  dense[i] = (labels[i / C] == (long long)(i % C)) ? (T)1 : (T)0;
}

// the projector's own fp64 (n, C) input array: a caller whose state is fp64 writes its prob straight into it and passes it as
// `dense_dev` (no copy); anything else is widened / copied into it
int glx_project_scores(glx_projector** pp, int64_t n, int C, double** scores_out) {
  GLX_CHECK(pp && scores_out, GLX_EINVAL, "glx_project_scores: null argument");
  GLX_CHECK(C <= 4096, GLX_EUNSUPPORTED, "glx_sweep_project: C=%d too large", C);
  if (!*pp) *pp = new glx_projector();
  int rc = proj_alloc((*pp)->b, n, C);
  if (rc) return rc;
  *scores_out = (*pp)->b.scores;
  return GLX_OK;
}

int glx_project_device(glx_projector** pp, const void* dense_dev, int dtype, int64_t n, int C, const double* priors,
                       double* weights_inout, double* err_out, int* steps_out, int max_steps, int similarity, hipStream_t st,
                       const long long** d_labels_out, const std::function<int(bool)>* hook) {
  GLX_CHECK(pp && dense_dev && weights_inout, GLX_EINVAL, "glx_project_device: null argument");
  GLX_CHECK(max_steps == 0 || priors, GLX_EINVAL, "glx_sweep_project: projection needs priors");
  GLX_CHECK(C <= 4096, GLX_EUNSUPPORTED, "glx_sweep_project: C=%d too large", C);
  if (!*pp) *pp = new glx_projector();
  ProjBufs& b = (*pp)->b;
  int rc = proj_alloc(b, n, C);
  if (rc) return rc;
  const int64_t total = n * C;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(to_f64_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dense_dev, b.scores, total);
  else if ((const void*)b.scores != dense_dev)
    hipLaunchKernelGGL(to_f64_kernel<double>, dim3(grid), dim3(256), 0, st, (const double*)dense_dev, b.scores, total);
  GLX_HIP(hipGetLastError());
  if (d_labels_out) *d_labels_out = b.labels;      // (known before the decision: the hook uses it)
  rc = proj_core(b, st, n, C, priors, weights_inout, err_out, steps_out, max_steps, similarity, dtype == GLX_F32, hook);
  if (rc) return rc;
  return GLX_OK;
}

int glx_onehot_device(const long long* d_labels, void* dense_dev, int dtype, int64_t n, int C, hipStream_t st) {
  const unsigned grid = (unsigned)((n * C + 255) / 256);
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(onehot_kernel<float>, dim3(grid), dim3(256), 0, st, d_labels, (float*)dense_dev, n, C);
  else
    hipLaunchKernelGGL(onehot_kernel<double>, dim3(grid), dim3(256), 0, st, d_labels, (double*)dense_dev, n, C);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

// vertex records of onehot(labels): record i holds vertex perm[i] (null: i), columns 0..C-1 = (label == column), padding zero --
// the state PoissonMBO's heat sweeps continue from (ssl.py:832), written without the detour through a dense (n, C) array
template <typename T>
__global__ __launch_bounds__(256) void onehot_records_kernel(const long long* __restrict__ labels, T* __restrict__ rec, int64_t n, int C, int ld,
                                                             const int32_t* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * ld) return;
  const int64_t row = i / ld;
  const int c = (int)(i % ld);
  rec[i] = (c < C && labels[perm ? (int64_t)perm[row] : row] == (long long)c) ? (T)1 : (T)0;
}

int glx_onehot_records(const long long* d_labels, void* rec, int dtype, int64_t n, const RecLayout& L, const int32_t* perm, hipStream_t st) {
  const unsigned grid = (unsigned)((n * L.ld + 255) / 256);
  if (n == 0) return GLX_OK;
  if (dtype == GLX_F32)
    hipLaunchKernelGGL(onehot_records_kernel<float>, dim3(grid), dim3(256), 0, st, d_labels, (float*)rec, n, L.C, L.ld, perm);
  else
    hipLaunchKernelGGL(onehot_records_kernel<double>, dim3(grid), dim3(256), 0, st, d_labels, (double*)rec, n, L.C, L.ld, perm);
  GLX_HIP(hipGetLastError());
  return GLX_OK;
}

void glx_projector_destroy(glx_projector* p) { delete p; }
