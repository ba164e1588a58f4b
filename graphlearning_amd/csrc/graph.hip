// Sparse operator residency: CSR (host copy, entry order preserved) -> sliced-ELL plans in HBM.
// Replaces utils.torch_sparse (reference graphlearning/utils.py:288-317).
#include "glx_internal.h"
#include <cstring>
#include <stdarg.h>
#include <algorithm>
#include <numeric>
#include <chrono>
#include <thread>
#include <string.h>
#include <stdlib.h>

static thread_local std::string g_err;

void glx_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

extern "C" const char* glx_last_error(void) { return g_err.c_str(); }
extern "C" int glx_version(void) { return 100; }

extern "C" int glx_device_count(int* n) {
  GLX_CHECK(n != nullptr, GLX_EINVAL, "glx_device_count: null output");
  GLX_HIP(hipGetDeviceCount(n));
  return GLX_OK;
}

extern "C" int glx_device_synchronize(void) {
  GLX_HIP(hipDeviceSynchronize());
  return GLX_OK;
}

extern "C" void glx_free(void* p) { free(p); }

// ---- device work-buffer pool -----------------------------------------------------------------------------
// hipMalloc / hipFree cost 0.1-1 ms each (hipFree synchronises the device): a kNN build makes thirty of them for 4 ms of
// kernels.  Work buffers of the one-shot entry points (knn.hip, assemble.hip) come from size-class free lists instead;
// at most POOL_CAP bytes stay cached per process, larger blocks go straight back to the runtime.  Callers release a
// buffer only after the stream that used it has been synchronised.
#include <atomic>
#include <map>
#include <mutex>
#define GLX_POOL(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
static const size_t POOL_CAP = 1ull << 30, POOL_BLOCK_MAX = 256ull << 20;
struct PoolState {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> idle;        // (device, class bytes) -> block
  std::map<void*, std::pair<int, size_t>> live;              // block -> (device, class bytes)
  size_t cached = 0;
};
static PoolState& pool() { static PoolState* p = new PoolState(); return *p; }   // never destroyed: outlives the HIP runtime's teardown
// glx_pool_set_enabled(0): every block straight from / back to the runtime and no idle work sets -- the ablation switch of the
// randomised soak (a result that changes with it names a buffer handed on while still in use) and of tests/test_gpu_switches.py
static bool g_pool_enabled = true;
// glx_pool_set_poison(b): every block handed out is first filled with the byte b (-1: off, the default) -- a debugging aid: a kernel
// that reads a work buffer before anything wrote it then computes from the pattern instead of from whatever an earlier call left there
static int g_pool_poison = -1;

static size_t pool_class(size_t bytes) {
  size_t c = 4096;
  while (c < bytes) c <<= 1;
  if (c > (64u << 20)) c = (bytes + (16u << 20) - 1) / (16u << 20) * (16u << 20);   // large blocks: 16 MiB granularity
  return c;
}

namespace {
struct WorkCache {
  std::mutex mu;
  std::map<int, std::vector<glx_work*>> idle;     // device -> sets nobody holds
};
WorkCache& work_cache() {
  static WorkCache* w = new WorkCache;   // never destroyed: HIP objects must not be torn down after the runtime at exit
  return *w;
}
void work_destroy(glx_work* w) {
  for (int i = 0; i < 4; ++i)
    if (w->ev[i]) hipEventDestroy(w->ev[i]);
  if (w->ev_side) hipEventDestroy(w->ev_side);
  if (w->side) hipStreamDestroy(w->side);
  if (w->stream) hipStreamDestroy(w->stream);
  if (w->stage) hipHostFree(w->stage);
  delete w;
}
}  // namespace

int glx_work_stage(glx_work* w, size_t bytes, void** out) {
  if (bytes > w->stage_bytes) {
    size_t want = (size_t)1 << 16;
    while (want < bytes) want <<= 1;
    if (w->stage) hipHostFree(w->stage);
    w->stage = nullptr;
    w->stage_bytes = 0;
    GLX_HIP(hipHostMalloc(&w->stage, want, hipHostMallocDefault));
    w->stage_bytes = want;
  }
  *out = w->stage;
  return GLX_OK;
}

int glx_work_acquire(int device, glx_work** out) {
  WorkCache& wc = work_cache();
  {
    std::lock_guard<std::mutex> lk(wc.mu);
    auto& v = wc.idle[device];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return GLX_OK;
    }
  }
  glx_work* w = new glx_work;
  w->device = device;
  hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&w->ev_side, hipEventDisableTiming);
  for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&w->ev[i]);
  if (e != hipSuccess) {
    glx_set_error("glx_work_acquire: %s", hipGetErrorString(e));
    work_destroy(w);
    return GLX_EHIP;
  }
  // The FIRST work set of a device starts the copy engines: the runtime creates a copy queue the first time an engine is picked
  // (7.7 ms each, measured with rocprofv3 --hip-trace: the first host-to-device copy, the first device-to-host copy, and one more
  // device-to-host copy when a second engine is drawn), which otherwise lands in the middle of the first graph builds.  Two
  // transfers per direction in flight at once, through page-locked memory, draw them now -- next to the 150 ms of runtime start-up.
  {
    static std::mutex mu;
    static std::vector<int> warmed;
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(warmed.begin(), warmed.end(), device) == warmed.end()) {
      warmed.push_back(device);
      // (two more streams than the set owns: a solver object's own stream is the third or fourth the process creates, and its first
      // device-to-host copy drew one more engine -- 10 ms inside the first fit of a process, scripts/first_fit_probe.py)
      const size_t half = (size_t)1 << 20;
      void *d = nullptr, *h = nullptr;
      hipStream_t extra[2] = {nullptr, nullptr};
      for (int q = 0; q < 2; ++q)
        if (hipStreamCreateWithFlags(&extra[q], hipStreamNonBlocking) != hipSuccess) extra[q] = nullptr;
      if (hipMalloc(&d, 8 * half) == hipSuccess && hipHostMalloc(&h, 8 * half, hipHostMallocDefault) == hipSuccess) {
        hipStream_t sts[4] = {w->stream, w->side, extra[0], extra[1]};
        for (int rep = 0; rep < 2; ++rep)
          for (int q = 0; q < 4; ++q) {
            if (!sts[q]) continue;
            hipMemcpyAsync((char*)h + (2 * q) * half, (char*)d + (2 * q) * half, half, hipMemcpyDeviceToHost, sts[q]);
            hipMemcpyAsync((char*)d + (2 * q + 1) * half, (char*)h + (2 * q + 1) * half, half, hipMemcpyHostToDevice, sts[q]);
          }
        for (int q = 0; q < 4; ++q)
          if (sts[q]) hipStreamSynchronize(sts[q]);
      }
      for (int q = 0; q < 2; ++q)
        if (extra[q]) hipStreamDestroy(extra[q]);
      if (h) hipHostFree(h);
      if (d) hipFree(d);
      (void)hipGetLastError();
    }
  }
  *out = w;
  return GLX_OK;
}

// the holder has synchronised the stream (nothing of its work is left on it)
void glx_work_release(glx_work* w) {
  if (!w) return;
  WorkCache& wc = work_cache();
  {
    std::lock_guard<std::mutex> lk(wc.mu);
    auto& v = wc.idle[w->device];
    if (g_pool_enabled && v.size() < 8) {
      v.push_back(w);
      return;
    }
  }
  work_destroy(w);
}

static void pinned_release_idle();
extern "C" int glx_pool_set_enabled(int enabled) {
  g_pool_enabled = enabled != 0;
  if (!g_pool_enabled) {             // what is idle now goes back to the runtime (blocks in use follow when they are released)
    std::vector<void*> idle;
    std::vector<glx_work*> sets;
    {
      PoolState& ps = pool();
      std::lock_guard<std::mutex> lk(ps.mu);
      for (auto& kv : ps.idle) idle.push_back(kv.second);
      ps.idle.clear();
      ps.cached = 0;
    }
    {
      WorkCache& wc = work_cache();
      std::lock_guard<std::mutex> lk(wc.mu);
      for (auto& kv : wc.idle) { sets.insert(sets.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
    }
    for (void* p : idle) hipFree(p);
    for (glx_work* w : sets) work_destroy(w);
    pinned_release_idle();
  }
  return GLX_OK;
}

extern "C" int glx_pool_set_poison(int byte) {
  g_pool_poison = byte < 0 ? -1 : (byte & 0xff);
  return GLX_OK;
}

static int pool_alloc_raw(void** out, size_t bytes);
int glx_pool_alloc(void** out, size_t bytes) {
  const int rc = pool_alloc_raw(out, bytes);
  if (!rc && g_pool_poison >= 0 && *out) {
    GLX_HIP(hipDeviceSynchronize());
    GLX_HIP(hipMemset(*out, g_pool_poison, pool_class(std::max<size_t>(bytes, 1))));
    GLX_HIP(hipDeviceSynchronize());
  }
  return rc;
}

static int pool_alloc_raw(void** out, size_t bytes) {
  int dev = 0;
  GLX_HIP(hipGetDevice(&dev));
  const size_t c = pool_class(std::max<size_t>(bytes, 1));
  PoolState& ps = pool();
  {
    std::lock_guard<std::mutex> lk(ps.mu);
    auto it = ps.idle.find({dev, c});
    if (it != ps.idle.end()) {
      *out = it->second;
      ps.idle.erase(it);
      ps.cached -= c;
      ps.live[*out] = {dev, c};
      return GLX_OK;
    }
  }
  *out = nullptr;
  hipError_t e = hipMalloc(out, c);
  if (e == hipErrorOutOfMemory) {
    // the idle blocks of the pool (up to POOL_CAP) are memory the runtime could hand out: give them back and try once more
    (void)hipGetLastError();
    std::vector<void*> idle;
    {
      std::lock_guard<std::mutex> lk(ps.mu);
      for (auto& kv : ps.idle) idle.push_back(kv.second);
      ps.idle.clear();
      ps.cached = 0;
    }
    for (void* p : idle) hipFree(p);
    e = hipMalloc(out, c);
  }
  GLX_HIP(e);
  std::lock_guard<std::mutex> lk(ps.mu);
  ps.live[*out] = {dev, c};
  return GLX_OK;
}

void glx_pool_free(void* p) {
  if (!p) return;
  PoolState& ps = pool();
  std::pair<int, size_t> key;
  {
    std::lock_guard<std::mutex> lk(ps.mu);
    auto it = ps.live.find(p);
    if (it == ps.live.end()) { hipFree(p); return; }
    key = it->second;
    ps.live.erase(it);
    if (g_pool_enabled && key.second <= POOL_BLOCK_MAX && ps.cached + key.second <= POOL_CAP) {
      ps.idle.insert({key, p});
      ps.cached += key.second;
      return;
    }
  }
  hipFree(p);
}

// Small page-locked blocks (the stop-value mirrors of a sweep object, the projector's image): hipHostMalloc 0.03-0.13 ms, hipHostFree
// 0.25 ms each -- two of each per model on a fresh graph.  Power-of-two classes from 4 KiB to 4 MiB, at most 32 MiB idle; follows the
// pool's switch (glx_pool_set_enabled).  The contract is the device pool's: nothing in flight still writes a block that is handed back.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::multimap<size_t, void*> idle;
  std::map<void*, size_t> live;
  size_t cached = 0;
};
PinnedPool& pinned_pool() { static PinnedPool* p = new PinnedPool(); return *p; }
}  // namespace
int glx_pinned_alloc(void** out, size_t bytes) {
  size_t c = 4096;
  while (c < bytes) c <<= 1;
  PinnedPool& pp = pinned_pool();
  if (c <= ((size_t)4 << 20)) {
    std::lock_guard<std::mutex> lk(pp.mu);
    auto it = pp.idle.find(c);
    if (it != pp.idle.end()) {
      *out = it->second;
      pp.idle.erase(it);
      pp.cached -= c;
      pp.live[*out] = c;
      return GLX_OK;
    }
  }
  *out = nullptr;
  GLX_HIP(hipHostMalloc(out, c, hipHostMallocDefault));
  std::lock_guard<std::mutex> lk(pp.mu);
  pp.live[*out] = c;
  return GLX_OK;
}
static void pinned_release_idle() {
  std::vector<void*> idle;
  {
    PinnedPool& pp = pinned_pool();
    std::lock_guard<std::mutex> lk(pp.mu);
    for (auto& kv : pp.idle) idle.push_back(kv.second);
    pp.idle.clear();
    pp.cached = 0;
  }
  for (void* p : idle) hipHostFree(p);
}
void glx_pinned_free(void* p) {
  if (!p) return;
  PinnedPool& pp = pinned_pool();
  {
    std::lock_guard<std::mutex> lk(pp.mu);
    auto it = pp.live.find(p);
    if (it != pp.live.end()) {
      const size_t c = it->second;
      pp.live.erase(it);
      if (g_pool_enabled && c <= ((size_t)4 << 20) && pp.cached + c <= ((size_t)32 << 20)) {
        pp.idle.insert({c, p});
        pp.cached += c;
        return;
      }
    }
  }
  hipHostFree(p);
}

// A few host worker threads that stay around (spawning eight std::threads costs ~0.3 ms: as much as hashing 25 MB).  parallel_for
// runs fn(t) for t in [0, nt) on the workers and the caller, and returns when all are done.  One job at a time; a process forked
// from one that had workers starts its own (threads do not survive fork).
#include <condition_variable>
#include <functional>
#include <unistd.h>
namespace {
struct HostPool {
  std::mutex mu, job_mu;
  std::condition_variable cv, done;
  std::vector<std::thread>* workers = nullptr;
  const std::function<void(int)>* fn = nullptr;
  int next = 0, total = 0, pending = 0;
  unsigned long gen = 0;
  pid_t pid = 0;
  void loop() {
    unsigned long seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return gen != seen; });
      seen = gen;
      while (next < total) {
        const int t = next++;
        lk.unlock();
        (*fn)(t);
        lk.lock();
        if (--pending == 0) done.notify_all();
      }
    }
  }
  void run(int nt, const std::function<void(int)>& f) {
    std::lock_guard<std::mutex> one(job_mu);
    if (nt <= 1) { for (int t = 0; t < nt; ++t) f(t); return; }
    {
      std::unique_lock<std::mutex> lk(mu);
      if (pid != getpid()) {            // first use, or a forked child: the parent's workers are not here (their handles are left alone)
        workers = new std::vector<std::thread>();
        pid = getpid();
      }
      const int want = std::min(7, nt - 1);
      while ((int)workers->size() < want) workers->emplace_back([this] { loop(); });
      fn = &f;
      next = 0;
      total = nt;
      pending = nt;
      ++gen;
    }
    cv.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    while (next < total) {               // the caller works too
      const int t = next++;
      lk.unlock();
      f(t);
      lk.lock();
      --pending;
    }
    done.wait(lk, [&] { return pending == 0; });
  }
};
HostPool& host_pool() {
  static HostPool* p = new HostPool();   // never destroyed: its workers may outlive static destruction
  return *p;
}
}  // namespace

// 128-bit content fingerprint (host): 4 MiB chunks hashed independently by a few threads -- two 64-bit multiply-mix lanes per
// chunk over 16-byte blocks -- and the chunk digests folded in order.  Not cryptographic: it tells an edited matrix from an
// unedited one (utils.matrix_fingerprint), 25 MB in ~0.2 ms instead of ~1 ms on one core.
static inline uint64_t fp_mix(uint64_t a, uint64_t b) {
  const __uint128_t m = (__uint128_t)(a ^ 0x9e3779b97f4a7c15ull) * (b ^ 0xc2b2ae3d27d4eb4full);
  return (uint64_t)m ^ (uint64_t)(m >> 64);
}
static void fp_chunk(const unsigned char* p, size_t len, uint64_t seed, uint64_t out[2]) {
  uint64_t h0 = seed ^ 0x243f6a8885a308d3ull ^ len, h1 = seed ^ 0x13198a2e03707344ull;
  size_t i = 0;
  for (; i + 32 <= len; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    h0 = fp_mix(h0 ^ w[0], w[1]) + (h0 << 1);
    h1 = fp_mix(h1 ^ w[2], w[3]) + (h1 << 1);
  }
  uint64_t tail[4] = {0, 0, 0, 0};
  memcpy(tail, p + i, len - i);
  h0 = fp_mix(h0 ^ tail[0], tail[1] + 0x5851f42d4c957f2dull);
  h1 = fp_mix(h1 ^ tail[2], tail[3] + 0x14057b7ef767814full);
  out[0] = fp_mix(h0, h1 + len);
  out[1] = fp_mix(h1, h0 ^ (len * 0x9fb21c651e98df25ull));
}
extern "C" int glx_host_fingerprint(const void* data, size_t bytes, uint64_t seed, uint64_t out[2]) {
  GLX_CHECK(out && (data || bytes == 0), GLX_EINVAL, "glx_host_fingerprint: null argument");
  const size_t CH = (size_t)1 << 20;
  const size_t nch = std::max<size_t>(1, (bytes + CH - 1) / CH);
  std::vector<uint64_t> dig(2 * nch);
  const unsigned char* p = (const unsigned char*)data;
  auto work = [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; ++c) fp_chunk(p + c * CH, std::min(CH, bytes - std::min(bytes, c * CH)), seed + c, &dig[2 * c]);
  };
  const int nt = (int)std::min<size_t>(8, nch);
  host_pool().run(nt, [&](int t) { work(nch * t / nt, nch * (t + 1) / nt); });
  uint64_t a = seed ^ bytes, b = ~seed;
  for (size_t c = 0; c < nch; ++c) {
    a = fp_mix(a ^ dig[2 * c], b + dig[2 * c + 1]);
    b = fp_mix(b ^ dig[2 * c + 1], a);
  }
  out[0] = a;
  out[1] = b;
  return GLX_OK;
}

// Page-locked, device-visible host memory for result arrays.  From 1 MiB on: anonymous memory aligned to 2 MiB with transparent huge
// pages asked for, faulted in by a few threads, then registered with the runtime -- 1.1 ms for 19 MB (one huge-page fault zeroes
// 2 MiB at memory speed) where hipHostMalloc takes 2.7-3.2 ms (1.7 ms with any explicit flag; scripts/probes/pin_probe.hip): fresh
// result arrays were most of what the first graph build of a new size paid.  Smaller blocks: hipHostMalloc.
#include <sys/mman.h>
namespace {
struct HostBlocks {
  std::mutex mu;
  std::map<void*, std::pair<void*, size_t>> mapped;      // user pointer -> (mmap base, mmap length)
};
HostBlocks& host_blocks() {
  static HostBlocks* h = new HostBlocks();
  return *h;
}
}  // namespace

extern "C" int glx_host_alloc(size_t bytes, void** out) {
  GLX_CHECK(out, GLX_EINVAL, "glx_host_alloc: null output");
  *out = nullptr;
  const size_t HUGE = (size_t)2 << 20;
  if (bytes >= ((size_t)1 << 20)) {
    const size_t len = (bytes + HUGE - 1) / HUGE * HUGE;
    void* base = mmap(nullptr, len + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base != MAP_FAILED) {
      char* a = (char*)(((uintptr_t)base + HUGE - 1) & ~(uintptr_t)(HUGE - 1));
      madvise(a, len, MADV_HUGEPAGE);
      const int64_t npages = (int64_t)(len / HUGE);
      const int nt = (int)std::min<int64_t>(4, npages);
      host_pool().run(nt, [&](int t) {               // one store per 4 KiB: faults the range in whether or not huge pages are granted
        for (size_t off = (size_t)(npages * t / nt) * HUGE, end = (size_t)(npages * (t + 1) / nt) * HUGE; off < end; off += 4096)
          *(volatile char*)(a + off) = 0;
      });
      if (hipHostRegister(a, len, hipHostRegisterDefault) == hipSuccess) {
        std::lock_guard<std::mutex> lk(host_blocks().mu);
        host_blocks().mapped[a] = {base, len + HUGE};
        *out = a;
        return GLX_OK;
      }
      (void)hipGetLastError();
      munmap(base, len + HUGE);
    }
  }
  GLX_HIP(hipHostMalloc(out, bytes > 0 ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped));
  return GLX_OK;
}

extern "C" int glx_host_free(void* p) {
  if (!p) return GLX_OK;
  std::pair<void*, size_t> m{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(host_blocks().mu);
    auto it = host_blocks().mapped.find(p);
    if (it != host_blocks().mapped.end()) {
      m = it->second;
      host_blocks().mapped.erase(it);
    }
  }
  if (m.first) {
    hipError_t e = hipHostUnregister(p);
    munmap(m.first, m.second);
    GLX_HIP(e);
    return GLX_OK;
  }
  GLX_HIP(hipHostFree(p));
  return GLX_OK;
}

// copy `len` bytes (a multiple of 8 when a sum is asked for) and add the 64-bit words up (wrapping)
static unsigned long long copy_and_sum(char* dst, const char* src, size_t len, bool want_sum) {
  if (!want_sum) { memcpy(dst, src, len); return 0ull; }
  const unsigned long long* s8 = (const unsigned long long*)src;
  unsigned long long* d8 = (unsigned long long*)dst;
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  const size_t nw = len / 8;
  size_t i = 0;
  for (; i + 4 <= nw; i += 4) {
    const unsigned long long v0 = s8[i], v1 = s8[i + 1], v2 = s8[i + 2], v3 = s8[i + 3];
    d8[i] = v0; d8[i + 1] = v1; d8[i + 2] = v2; d8[i + 3] = v3;
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < nw; ++i) { d8[i] = s8[i]; a0 += s8[i]; }
  return a0 + a1 + a2 + a3;
}

namespace {
struct Uploader {                 // per calling thread and device; never destroyed (HIP objects must not outlive the runtime's teardown)
  void* stage = nullptr;          // page-locked: two halves that take turns
  size_t bytes = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  unsigned long long* sum = nullptr;        // device word of the check
  unsigned long long* sum_host = nullptr;   // its page-locked mirror
};
Uploader* my_uploader() {
  static thread_local std::map<int, Uploader*> per_device;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  Uploader*& u = per_device[dev];
  if (!u) u = new Uploader();
  return u;
}
}  // namespace

// the staged copy: returns the wrapping sum of the 64-bit words (a zero-padded last word for a length that is no multiple of 8) when asked
static int upload_staged(Uploader* w, void* dst, const void* src, size_t bytes, hipStream_t st, unsigned long long* sum_out, size_t stage_shift) {
  if (sum_out) *sum_out = 0ull;
  if (bytes == 0) return GLX_OK;
  const size_t HALF_MAX = (size_t)16 << 20;
  size_t half = (size_t)1 << 18;
  while (half < bytes + stage_shift && half < HALF_MAX) half <<= 1;
  if (w->bytes < 2 * half) {
    if (w->stage) hipHostFree(w->stage);          // (nothing reads it any more: every call ends behind its last copy, see below)
    w->stage = nullptr;
    w->bytes = 0;
    GLX_HIP(hipHostMalloc(&w->stage, 2 * half, hipHostMallocDefault));
    w->bytes = 2 * half;
  }
  const size_t h = w->bytes / 2;
  GLX_CHECK(stage_shift % 64 == 0 && stage_shift < h / 2, GLX_EINVAL, "glx_upload: bad staging shift");
  // bytes of a half in use per piece (a multiple of 64: whole words); big uploads go in pieces of 4 MB so that the host threads fill one half
  // while the copy engine empties the other
  const size_t room = std::min<size_t>((h - stage_shift) / 64 * 64, bytes > ((size_t)6 << 20) ? ((size_t)4 << 20) : (size_t)-1);
  for (int i = 0; i < 2; ++i)
    if (!w->ev[i]) GLX_HIP(hipEventCreateWithFlags(&w->ev[i], hipEventDisableTiming));
  int turn = 0;
  unsigned long long total = 0;
  for (size_t off = 0; off < bytes; off += room, turn ^= 1) {
    const size_t len = std::min(room, bytes - off);
    const size_t whole = len / 8 * 8;
    char* stage = (char*)w->stage + (size_t)turn * h + stage_shift;
    // the copy of THIS call that last read this half.  Never an event of an earlier call: the runtime's hipEventSynchronize looks at the
    // stream the event was last recorded on, and that stream may be gone by now (a work set's stream destroyed with its set: the pool
    // switched off) -- "operation not permitted on an event last recorded in a capturing stream" out of freed memory, in the first
    // upload after such a stream's address was reused (round 6, tests/test_gpu_switches.py).  Every call therefore ends with its
    // copies complete (the checked path waits for its sum, the unchecked one for the stream) and starts with both halves free.
    if (off >= 2 * room) GLX_HIP(hipEventSynchronize(w->ev[turn]));
    const int nt = (int)std::min<size_t>(8, std::max<size_t>(1, whole >> 19));
    if (nt > 1) {
      unsigned long long part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      host_pool().run(nt, [&](int t) {
        const size_t a = whole * (size_t)t / nt / 64 * 64, b2 = t + 1 == nt ? whole : whole * (size_t)(t + 1) / nt / 64 * 64;
        part[t] = copy_and_sum(stage + a, (const char*)src + off + a, b2 - a, sum_out != nullptr);
      });
      for (int t = 0; t < 8; ++t) total += part[t];
    } else {
      total += copy_and_sum(stage, (const char*)src + off, whole, sum_out != nullptr);
    }
    if (len > whole) {                                   // the last bytes of the upload: a word padded with zeros for the sum
      unsigned long long tail = 0;
      memcpy(&tail, (const char*)src + off + whole, len - whole);
      memcpy(stage + whole, (const char*)src + off + whole, len - whole);
      total += tail;
    }
    GLX_HIP(hipMemcpyAsync((char*)dst + off, stage, len, hipMemcpyHostToDevice, st));
    GLX_HIP(hipEventRecord(w->ev[turn], st));
  }
  if (sum_out) *sum_out = total;
  else GLX_HIP(hipStreamSynchronize(st));             // (the checked caller synchronises behind its sum kernel)
  return GLX_OK;
}

// (one atomic per workgroup and at most UPLOAD_SUM_BLOCKS of them: 8192 wavefronts adding to ONE address took 100 us for 11 MB, the
// additions themselves 5)
#define UPLOAD_SUM_BLOCKS 512
__global__ __launch_bounds__(256) void upload_sum_kernel(const unsigned long long* __restrict__ p, int64_t nwords, int tail_bytes,
                                                         unsigned long long* __restrict__ out) {
  unsigned long long a = 0;
  const int64_t step = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < nwords; i += 4 * step) {          // four loads in flight per thread
    const unsigned long long v0 = p[i], v1 = p[i + step], v2 = p[i + 2 * step], v3 = p[i + 3 * step];
    a += v0 + v1 + v2 + v3;
  }
  for (; i < nwords; i += step) a += p[i];
  if (tail_bytes && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned char* t = (const unsigned char*)(p + nwords);
    unsigned long long v = 0;
    for (int q = 0; q < tail_bytes; ++q) v |= (unsigned long long)t[q] << (8 * q);
    a += v;
  }
  for (int off = 32; off >= 1; off >>= 1) a += (unsigned long long)__shfl_xor((long long)a, off);
  __shared__ unsigned long long sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = sh[0] + sh[1] + sh[2] + sh[3];
    if (t) atomicAdd(out, t);
  }
}
static inline unsigned upload_sum_grid(int64_t nwords) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(UPLOAD_SUM_BLOCKS, (nwords + 1023) / 1024));
}

static std::atomic<unsigned long long> g_upload_stats[4];       // uploads checked, sums that differed, uploads repeated successfully, given up
extern "C" int glx_upload_stats(unsigned long long out[4]) {
  GLX_CHECK(out, GLX_EINVAL, "glx_upload_stats: null output");
  for (int q = 0; q < 4; ++q) out[q] = g_upload_stats[q].load();
  return GLX_OK;
}
static int g_upload_mode = 0;      // glx_upload_set_mode: 0 staged + checked (default), 1 staged, 2 hipMemcpyAsync from the caller's memory (rounds 1-5)
extern "C" int glx_upload_set_mode(int mode) {
  GLX_CHECK(mode >= 0 && mode <= 2, GLX_EINVAL, "glx_upload_set_mode: 0 (staged + checked), 1 (staged) or 2 (direct)");
  g_upload_mode = mode;
  return GLX_OK;
}

int glx_upload(void* dst, const void* src, size_t bytes, hipStream_t st, const char* what) {
  if (bytes == 0) return GLX_OK;
  GLX_CHECK(dst && src, GLX_EINVAL, "%s: null pointer in an upload of %zu bytes", what, bytes);
  if (bytes < ((size_t)128 << 10) || g_upload_mode == 2) {
    GLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    return GLX_OK;
  }
  Uploader* w = my_uploader();
  const bool aligned = ((uintptr_t)dst % 8 == 0) && ((uintptr_t)src % 8 == 0);
  // (word sums need 8-byte aligned ends; the staging alone is what keeps the transfer off the runtime's pageable path)
  if (!aligned || g_upload_mode == 1) return upload_staged(w, dst, src, bytes, st, nullptr, 0);
  if (!w->sum) GLX_HIP(hipMalloc((void**)&w->sum, 64));
  if (!w->sum_host) GLX_HIP(hipHostMalloc((void**)&w->sum_host, 64, hipHostMallocDefault));
  ++g_upload_stats[0];
  for (int attempt = 0; attempt < 4; ++attempt) {
    unsigned long long want = 0;
    // (a repeat goes through another part of the staging area)
    int rc = upload_staged(w, dst, src, bytes, st, &want, (size_t)attempt * 12288);
    if (rc) return rc;
    GLX_HIP(hipMemsetAsync(w->sum, 0, 8, st));
    const int64_t nw = (int64_t)(bytes / 8);
    hipLaunchKernelGGL(upload_sum_kernel, dim3(upload_sum_grid(nw)), dim3(256), 0, st,
                       (const unsigned long long*)dst, nw, (int)(bytes % 8), w->sum);
    GLX_HIP(hipGetLastError());
    GLX_HIP(hipMemcpyAsync(w->sum_host, w->sum, 8, hipMemcpyDeviceToHost, st));
    GLX_HIP(hipStreamSynchronize(st));
    if (*w->sum_host == want) {
      if (attempt) ++g_upload_stats[2];
      return GLX_OK;
    }
    ++g_upload_stats[1];
    // what went wrong where: the staging area against the caller's array (single-piece uploads: the area still holds the whole array), and
    // the device copy, read back THROUGH PAGE-LOCKED MEMORY (a read-back into pageable memory can show the same holes), against it
    size_t stage_bad = 0, dev_bad = 0, first = 0, last = 0, zeros = 0;
    const unsigned long long* s8 = (const unsigned long long*)src;
    const size_t nwords = bytes / 8;
    if (bytes <= ((size_t)6 << 20) && bytes + (size_t)attempt * 12288 <= w->bytes / 2) {      // (one piece: the area still holds the whole array)
      const unsigned long long* g8 = (const unsigned long long*)((const char*)w->stage + (size_t)attempt * 12288);
      for (size_t i = 0; i < nwords; ++i) stage_bad += g8[i] != s8[i];
    }
    unsigned long long* back = nullptr;
    if (hipHostMalloc((void**)&back, std::max<size_t>(bytes, 64), hipHostMallocDefault) == hipSuccess && back) {
      if (hipMemcpy(back, dst, bytes, hipMemcpyDeviceToHost) == hipSuccess) {
        for (size_t i = 0; i < nwords; ++i)
          if (back[i] != s8[i]) { if (!dev_bad) first = i; last = i; ++dev_bad; zeros += back[i] == 0; }
      }
      hipHostFree(back);
    }
    (void)hipGetLastError();
    fprintf(stderr, "[glx] upload check (%s, pid %d, attempt %d): %zu bytes arrived with sum %016llx instead of %016llx -- the staging area differs from "
                    "the caller's array in %zu words, the device copy in %zu (bytes %zu .. %zu of the upload, %zu of them zero); repeating the upload\n",
            what, (int)getpid(), attempt, bytes, *w->sum_host, want, stage_bad, dev_bad, first * 8, last * 8 + 7, zeros);
  }
  ++g_upload_stats[3];
  glx_set_error("%s: the upload of %zu bytes did not arrive intact in four attempts", what, bytes);
  return GLX_EHIP;
}

int glx_download(void* dst, const void* src, size_t bytes, hipStream_t st, const char* what) {
  if (bytes == 0) return GLX_OK;
  GLX_CHECK(dst && src, GLX_EINVAL, "%s: null pointer in a download of %zu bytes", what, bytes);
  bool direct = bytes < ((size_t)128 << 10) || g_upload_mode == 2;
  if (!direct) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged)) direct = true;   // page-locked
    (void)hipGetLastError();
  }
  if (direct) {
    GLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    return GLX_OK;
  }
  Uploader* w = my_uploader();
  const bool check = g_upload_mode == 0 && ((uintptr_t)dst % 8 == 0) && ((uintptr_t)src % 8 == 0) && bytes % 8 == 0;
  if (check) {
    if (!w->sum) GLX_HIP(hipMalloc((void**)&w->sum, 64));
    if (!w->sum_host) GLX_HIP(hipHostMalloc((void**)&w->sum_host, 64, hipHostMallocDefault));
    ++g_upload_stats[0];
  }
  for (int attempt = 0; attempt < 4; ++attempt) {
    const size_t shift = (size_t)attempt * 12288;
    // the staging area is the uploads' own: one half of it is used here, piece by piece
    const size_t HALF_MAX = (size_t)16 << 20;
    size_t half = (size_t)1 << 18;
    while (half < bytes + shift && half < HALF_MAX) half <<= 1;
    if (w->bytes < 2 * half) {
      if (w->stage) hipHostFree(w->stage);      // (every upload of this thread ended with its copies complete: upload_staged)
      w->stage = nullptr;
      w->bytes = 0;
      GLX_HIP(hipHostMalloc(&w->stage, 2 * half, hipHostMallocDefault));
      w->bytes = 2 * half;
    }
    // (no upload of this thread still reads the area -- and no event of an earlier call is waited for here either: the stream it was
    // recorded on may be gone, and the runtime's hipEventSynchronize looks at that stream; see upload_staged)
    const size_t room = (w->bytes / 2 - shift) / 64 * 64;
    unsigned long long want = 0, got = 0;
    if (check) {
      GLX_HIP(hipMemsetAsync(w->sum, 0, 8, st));
      const int64_t nw = (int64_t)(bytes / 8);
      hipLaunchKernelGGL(upload_sum_kernel, dim3(upload_sum_grid(nw)), dim3(256), 0, st,
                         (const unsigned long long*)src, nw, 0, w->sum);
      GLX_HIP(hipGetLastError());
      GLX_HIP(hipMemcpyAsync(w->sum_host, w->sum, 8, hipMemcpyDeviceToHost, st));
    }
    char* stage = (char*)w->stage + shift;
    for (size_t off = 0; off < bytes; off += room) {
      const size_t len = std::min(room, bytes - off);
      GLX_HIP(hipMemcpyAsync(stage, (const char*)src + off, len, hipMemcpyDeviceToHost, st));
      GLX_HIP(hipStreamSynchronize(st));
      const size_t whole = check ? len : 0;
      const int nt = (int)std::min<size_t>(4, std::max<size_t>(1, len >> 20));
      if (nt > 1) {
        unsigned long long part[4] = {0, 0, 0, 0};
        host_pool().run(nt, [&](int t) {
          const size_t a = len * (size_t)t / nt / 64 * 64, b2 = t + 1 == nt ? len : len * (size_t)(t + 1) / nt / 64 * 64;
          part[t] = copy_and_sum((char*)dst + off + a, stage + a, b2 - a, whole != 0);
        });
        got += part[0] + part[1] + part[2] + part[3];
      } else {
        got += copy_and_sum((char*)dst + off, stage, len, whole != 0);
      }
    }
    if (!check) return GLX_OK;
    want = *w->sum_host;
    if (got == want) {
      if (attempt) ++g_upload_stats[2];
      return GLX_OK;
    }
    ++g_upload_stats[1];
    fprintf(stderr, "[glx] download check (%s, pid %d, attempt %d): %zu bytes came down with sum %016llx instead of %016llx; repeating the download\n", what,
            (int)getpid(), attempt, bytes, got, want);
  }
  ++g_upload_stats[3];
  glx_set_error("%s: the download of %zu bytes did not arrive intact in four attempts", what, bytes);
  return GLX_EHIP;
}

int glx_download_sync(void* dst, const void* src, size_t bytes, const char* what) {
  GLX_UP(glx_download(dst, src, bytes, nullptr, what));
  GLX_HIP(hipStreamSynchronize(nullptr));
  return GLX_OK;
}

int glx_upload_sync(void* dst, const void* src, size_t bytes, const char* what) {
  GLX_UP(glx_upload(dst, src, bytes, nullptr, what));
  GLX_HIP(hipStreamSynchronize(nullptr));
  return GLX_OK;
}

int glx_make_layout(int C, int dtype, bool has_w, RecLayout* L) {
  GLX_CHECK(C >= 1, GLX_EINVAL, "layout: C must be >= 1 (got %d)", C);
  GLX_CHECK(dtype == GLX_F32 || dtype == GLX_F64, GLX_EINVAL, "layout: bad dtype %d", dtype);
  const int es = dtype == GLX_F32 ? 4 : 8;
  const int nvec = (C + 3) / 4;
  const int lanes = nvec + (has_w ? 1 : 0);
  GLX_CHECK(lanes <= 64, GLX_EUNSUPPORTED, "layout: C=%d needs %d lanes per row (max 64 -> C <= %d)", C, lanes,
            has_w ? 252 : 256);
  int G = 4;
  while (G < lanes) G *= 2;
  int bytes = lanes * 4 * es;   // the stop value gets a full 4-wide slot so every lane issues the same gather
  int rb = 32;
  while (rb < bytes && rb < 128) rb *= 2;      // 32 / 64 / 128-byte records stay line-aligned
  if (rb < bytes) rb = (bytes + 63) / 64 * 64; // larger records: multiple of 64 bytes
  L->C = C;
  L->nvec = nvec;
  L->ld = rb / es;
  L->woff = has_w ? nvec * 4 * es : -1;
  L->G = G;
  L->esize = es;
  L->ngroups = 1;
  L->nstop = has_w ? 1 : 0;
  return GLX_OK;
}

int glx_make_layout_groups(int C, int B, int dtype, RecLayout* L) {
  GLX_CHECK(C >= 1 && B >= 1 && B <= 32, GLX_EINVAL, "layout: %d groups of %d columns (1 .. 32 groups)", B, C);
  GLX_CHECK(dtype == GLX_F32 || dtype == GLX_F64, GLX_EINVAL, "layout: bad dtype %d", dtype);
  if (B == 1) return glx_make_layout(C, dtype, true, L);
  const int es = dtype == GLX_F32 ? 4 : 8;
  const int nvec = (C * B + 3) / 4;
  const int nstop = (B * 8 + 4 * es - 1) / (4 * es);     // fp64 stop values packed behind the columns: 4 (fp64 state) / 2 (fp32) per lane
  const int lanes = nvec + nstop;
  GLX_CHECK(lanes <= 64, GLX_EUNSUPPORTED, "layout: %d groups of %d columns need %d lanes per row (max 64)", B, C, lanes);
  int G = 8;
  while (G < lanes) G *= 2;
  const int bytes = lanes * 4 * es;
  const int rb = (bytes + 127) / 128 * 128;              // whole 128-byte lines: a gather never shares a line with another record
  L->C = C * B;
  L->nvec = nvec;
  L->ld = rb / es;
  L->woff = nvec * 4 * es;
  L->G = G;
  L->esize = es;
  L->ngroups = B;
  L->nstop = nstop;
  return GLX_OK;
}

static int graph_keep_order(glx_graph* g) {
  GLX_CHECK(g->plans.empty() && !g->order_ready, GLX_EINVAL, "glx_graph_set_order: call before the operator is first used");
  g->keep_order = true;
  return GLX_OK;
}

extern "C" int glx_graph_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                                const int32_t* col, const double* val, int state_dtype, int device,
                                glx_graph** out) {
  GLX_CHECK(out != nullptr, GLX_EINVAL, "glx_graph_create: null output");
  *out = nullptr;
  GLX_CHECK(n_rows >= 0 && n_cols >= 0 && nnz >= 0, GLX_EINVAL, "glx_graph_create: negative size");
  GLX_CHECK(n_rows < (1ll << 31) && n_cols < (1ll << 31), GLX_EINVAL, "glx_graph_create: n must fit int32");
  GLX_CHECK(rowptr && (nnz == 0 || (col && val)), GLX_EINVAL, "glx_graph_create: null CSR array");
  GLX_CHECK(state_dtype == GLX_F32 || state_dtype == GLX_F64, GLX_EINVAL, "glx_graph_create: bad dtype %d", state_dtype);
  GLX_CHECK(rowptr[0] == 0 && rowptr[n_rows] == nnz, GLX_EINVAL, "glx_graph_create: rowptr[0]=%d rowptr[n]=%d nnz=%lld",
            rowptr[0], rowptr[n_rows], (long long)nnz);
  int ndev = 0;
  GLX_HIP(hipGetDeviceCount(&ndev));
  GLX_CHECK(device >= 0 && device < ndev, GLX_EINVAL, "glx_graph_create: device %d of %d", device, ndev);
  int max_row = 0;
  for (int64_t i = 0; i < n_rows; ++i) {
    const int len = rowptr[i + 1] - rowptr[i];
    GLX_CHECK(len >= 0, GLX_EINVAL, "glx_graph_create: rowptr not monotone at row %lld", (long long)i);
    max_row = std::max(max_row, len);
  }
  for (int64_t e = 0; e < nnz; ++e)
    GLX_CHECK(col[e] >= 0 && col[e] < n_cols, GLX_EINVAL, "glx_graph_create: column %d out of range at entry %lld", col[e], (long long)e);
  glx_graph* g = new glx_graph();
  g->n_rows = n_rows;
  g->n_cols = n_cols;
  g->nnz = nnz;
  g->dtype = state_dtype;
  g->device = device;
  g->max_row = max_row;
  g->h_rowptr.assign(rowptr, rowptr + n_rows + 1);
  g->h_col.assign(col, col + nnz);
  g->h_val.assign(val, val + nnz);
  g->plans.reserve(8);   // plan pointers handed out stay valid
  *out = g;
  return GLX_OK;
}

// The same operator object from CSR arrays that stay on the DEVICE: nothing but the row pointers is kept on the host, plans are
// filled straight from the resident arrays.  rowsum_out (may be NULL): the row sums `A * ones` as scipy's csr_matvec forms them
// (sequentially, in stored order) -- the degree vector of reference graph.py:108-122 -- computed on the device.  What a fresh
// ssl.poisson fit does with weightmatrix.knn's matrix (1.3 ms of host passes and two copies of the 14 MB operator otherwise).
__global__ __launch_bounds__(256) void csr_check_rowsum_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               const double* __restrict__ val, int64_t n_rows, int64_t n_cols,
                                                               double* __restrict__ rowsum, int* __restrict__ bad) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  double s = 0.0;
  int b = 0;
  for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int32_t c = col[e];
    if (c < 0 || c >= n_cols) b = 1;
    s = s + val[e] * 1.0;
  }
  if (rowsum) rowsum[i] = s;
  if (b) *bad = 1;
}

extern "C" int glx_graph_create_resident(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                                         const double* val, int state_dtype, int device, double* rowsum_out, glx_graph** out) {
  GLX_CHECK(out != nullptr, GLX_EINVAL, "glx_graph_create_resident: null output");
  *out = nullptr;
  GLX_CHECK(n_rows >= 0 && n_cols >= 0 && nnz >= 0, GLX_EINVAL, "glx_graph_create_resident: negative size");
  GLX_CHECK(n_rows < (1ll << 31) && n_cols < (1ll << 31), GLX_EINVAL, "glx_graph_create_resident: n must fit int32");
  GLX_CHECK(rowptr && (nnz == 0 || (col && val)), GLX_EINVAL, "glx_graph_create_resident: null CSR array");
  GLX_CHECK(state_dtype == GLX_F32 || state_dtype == GLX_F64, GLX_EINVAL, "glx_graph_create_resident: bad dtype %d", state_dtype);
  GLX_CHECK(rowptr[0] == 0 && rowptr[n_rows] == nnz, GLX_EINVAL, "glx_graph_create_resident: rowptr[0]=%d rowptr[n]=%d nnz=%lld",
            rowptr[0], rowptr[n_rows], (long long)nnz);
  int ndev = 0;
  GLX_HIP(hipGetDeviceCount(&ndev));
  GLX_CHECK(device >= 0 && device < ndev, GLX_EINVAL, "glx_graph_create_resident: device %d of %d", device, ndev);
  int max_row = 0;
  for (int64_t i = 0; i < n_rows; ++i) {
    const int len = rowptr[i + 1] - rowptr[i];
    GLX_CHECK(len >= 0, GLX_EINVAL, "glx_graph_create_resident: rowptr not monotone at row %lld", (long long)i);
    max_row = std::max(max_row, len);
  }
  GLX_HIP(hipSetDevice(device));
  glx_graph* g = new glx_graph();
  g->n_rows = n_rows;
  g->n_cols = n_cols;
  g->nnz = nnz;
  g->dtype = state_dtype;
  g->device = device;
  g->max_row = max_row;
  g->h_rowptr.assign(rowptr, rowptr + n_rows + 1);
  g->plans.reserve(8);
  struct Guard { glx_graph* g; ~Guard() { if (g) glx_graph_destroy(g); } } guard{g};
  glx_work* w = nullptr;
  int rc = glx_work_acquire(device, &w);
  if (rc) return rc;
  struct WorkGuard { glx_work* w; ~WorkGuard() { hipStreamSynchronize(w->stream); glx_work_release(w); } } wguard{w};
  hipStream_t st = w->stream;
  GLX_POOL(glx_pool_alloc((void**)&g->d_src_rowptr, (size_t)(n_rows + 1) * 4));
  GLX_POOL(glx_pool_alloc((void**)&g->d_src_col, std::max<size_t>((size_t)nnz * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&g->d_src_val, std::max<size_t>((size_t)nnz * 8, 8)));
  GLX_UP(glx_upload(g->d_src_rowptr, rowptr, (size_t)(n_rows + 1) * 4, st, __func__));
  if (nnz > 0) {
    GLX_UP(glx_upload(g->d_src_col, col, (size_t)nnz * 4, st, __func__));
    GLX_UP(glx_upload(g->d_src_val, val, (size_t)nnz * 8, st, __func__));
  }
  // column range check (+ the row sums) on the device; results through the work set's page-locked staging area
  char* stage = nullptr;
  rc = glx_work_stage(w, (size_t)n_rows * 8 + 64, (void**)&stage);
  if (rc) return rc;
  double* d_sum = nullptr;
  int* d_bad = nullptr;
  struct Tmp { void *a = nullptr, *b = nullptr; ~Tmp() { glx_pool_free(a); glx_pool_free(b); } } tmp;
  rc = glx_pool_alloc(&tmp.a, std::max<size_t>((size_t)n_rows * 8, 8));
  if (!rc) rc = glx_pool_alloc(&tmp.b, 64);
  if (rc) return rc;
  d_sum = (double*)tmp.a;
  d_bad = (int*)tmp.b;
  GLX_HIP(hipMemsetAsync(d_bad, 0, 4, st));
  if (n_rows > 0) {
    hipLaunchKernelGGL(csr_check_rowsum_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, (const int32_t*)g->d_src_rowptr,
                       (const int32_t*)g->d_src_col, (const double*)g->d_src_val, n_rows, n_cols, rowsum_out ? d_sum : nullptr, d_bad);
    GLX_HIP(hipGetLastError());
  }
  int* h_bad = (int*)(stage + (size_t)n_rows * 8);
  GLX_HIP(hipMemcpyAsync(h_bad, d_bad, 4, hipMemcpyDeviceToHost, st));
  if (rowsum_out && n_rows > 0) GLX_HIP(hipMemcpyAsync(stage, d_sum, (size_t)n_rows * 8, hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  GLX_CHECK(*h_bad == 0, GLX_EINVAL, "glx_graph_create_resident: column index out of range");
  if (rowsum_out && n_rows > 0) memcpy(rowsum_out, stage, (size_t)n_rows * 8);
  guard.g = nullptr;
  *out = g;
  return GLX_OK;
}

// Row i of the operator = row i of the resident source with its entries in reverse order (reverse_rows != 0) and multiplied by
// row_scale[i] (NULL: as they are).  Before the operator is first used.  reverse + D^-1: P = D^-1 W^T of a symmetric W exactly as
// scipy writes it down (`D * W.transpose()`: csr_matmat emits every row in reverse, reference ssl.py:634-635).
extern "C" int glx_graph_set_row_transform(glx_graph* g, const double* row_scale, int reverse_rows) {
  GLX_CHECK(g, GLX_EINVAL, "glx_graph_set_row_transform: null graph");
  GLX_CHECK(g->d_src_col != nullptr, GLX_EINVAL, "glx_graph_set_row_transform: not a resident graph (glx_graph_create_resident)");
  GLX_CHECK(g->plans.empty(), GLX_EINVAL, "glx_graph_set_row_transform: call before the operator is first used");
  GLX_HIP(hipSetDevice(g->device));
  g->reverse_rows = reverse_rows != 0;
  glx_pool_free(g->d_row_scale);
  g->d_row_scale = nullptr;
  if (row_scale && g->n_rows > 0) {
    GLX_POOL(glx_pool_alloc((void**)&g->d_row_scale, (size_t)g->n_rows * 8));
    glx_work* w = nullptr;             // (a work set's stream, not a blocking copy on the NULL stream: see glx_graph_set_order)
    int rc = glx_work_acquire(g->device, &w);
    if (rc) return rc;
    struct WorkGuard { glx_work* w; ~WorkGuard() { hipStreamSynchronize(w->stream); glx_work_release(w); } } wguard{w};
    GLX_UP(glx_upload(g->d_row_scale, row_scale, (size_t)g->n_rows * 8, w->stream, __func__));
    GLX_HIP(hipStreamSynchronize(w->stream));
  }
  return GLX_OK;
}

static void free_plan(SellPlan& p) {      // (callers: glx_graph_destroy behind its device synchronisation)
  glx_pool_free(p.d_slot_row);
  glx_pool_free(p.d_slot_len);
  glx_pool_free(p.d_slice_hdr);
  glx_pool_free(p.d_col);
  glx_pool_free(p.d_val);
  p = SellPlan();
}

extern "C" int glx_graph_destroy(glx_graph* g) {
  if (!g) return GLX_OK;
  hipSetDevice(g->device);
  // the operator's buffers go back to the size-class pool (hipFree of a block of megabytes costs 0.23 ms and there were ten of them: a
  // third of a first fit on a fresh graph, profiles/r06_fresh_path.txt).  hipFree waited for the device; the pool's contract is that nothing
  // in flight still uses a block that is handed back -- so wait here, once.
  hipDeviceSynchronize();
  for (auto& p : g->plans) free_plan(p);
  if (g->cg_ws) glx_cg_ws_destroy(g->cg_ws);
  glx_pool_free(g->d_perm);
  glx_pool_free(g->d_inv);
  glx_pool_free(g->d_src_rowptr);
  glx_pool_free(g->d_src_col);
  glx_pool_free(g->d_src_val);
  glx_pool_free(g->d_row_scale);
  delete g;
  return GLX_OK;
}

extern "C" int glx_graph_info(const glx_graph* g, int64_t info[8]) {
  GLX_CHECK(g && info, GLX_EINVAL, "glx_graph_info: null argument");
  for (int i = 0; i < 8; ++i) info[i] = 0;
  info[0] = g->n_rows;
  info[1] = g->n_cols;
  info[2] = g->nnz;
  if (!g->plans.empty()) {
    info[3] = g->plans[0].stored;
    info[4] = g->plans[0].nslices;
    info[5] = g->plans[0].R;
  }
  info[6] = g->max_row;
  info[7] = g->h_perm.empty() ? 0 : 1;
  return GLX_OK;
}

// Reverse Cuthill-McKee on the symmetrised pattern: neighbours get nearby ids, so the
// contiguous id range an XCD works on mostly gathers records of that same range (its own L2).
static void rcm_order_arrays(int64_t n, const int32_t* h_rowptr, const int32_t* h_col, std::vector<int32_t>& perm, bool sort_children);

static void rcm_order(const glx_graph* g, std::vector<int32_t>& perm, bool sort_children = true) {
  rcm_order_arrays(g->n_rows, g->h_rowptr.data(), g->h_col.data(), perm, sort_children);   // (resident sources: glx_graph_ensure_order fetched the pattern)
}

static void rcm_order_arrays(int64_t n, const int32_t* h_rowptr, const int32_t* h_col, std::vector<int32_t>& perm, bool sort_children) {
  // The order is a locality heuristic: any permutation is correct.  When every vertex has as many
  // stored entries in its row as in its column the pattern is (almost certainly) symmetric --
  // P = D^-1 W^T, Laplacians -- and the rows themselves serve as adjacency lists; otherwise the
  // pattern is symmetrised first.
  std::vector<int64_t> indeg(n, 0);
  for (int64_t e = 0; e < h_rowptr[n]; ++e) indeg[h_col[e]]++;
  bool balanced = true;
  for (int64_t i = 0; i < n && balanced; ++i) balanced = indeg[i] == (int64_t)(h_rowptr[i + 1] - h_rowptr[i]);
  std::vector<int64_t> ptr_own;
  std::vector<int32_t> adj_own;
  const int32_t* adj;
  const int64_t* ptr;
  std::vector<int64_t> ptr64;
  if (balanced) {
    ptr64.assign(h_rowptr, h_rowptr + n + 1);
    ptr = ptr64.data();
    adj = h_col;
  } else {
    ptr_own.assign(n + 1, 0);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t e = h_rowptr[i]; e < h_rowptr[i + 1]; ++e) {
        ptr_own[i + 1]++;
        ptr_own[h_col[e] + 1]++;
      }
    for (int64_t i = 0; i < n; ++i) ptr_own[i + 1] += ptr_own[i];
    adj_own.resize(ptr_own[n]);
    std::vector<int64_t> fill(ptr_own.begin(), ptr_own.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int64_t e = h_rowptr[i]; e < h_rowptr[i + 1]; ++e) {
        const int32_t j = h_col[e];
        adj_own[fill[i]++] = j;
        adj_own[fill[j]++] = (int32_t)i;
      }
    ptr = ptr_own.data();
    adj = adj_own.data();
  }
  auto degree = [&](int32_t v) { return ptr[v + 1] - ptr[v]; };
  // vertices by ascending degree, ties by index (a counting sort: degrees are small integers)
  std::vector<int32_t> by_deg(n);
  {
    int64_t dmax = 0;
    for (int64_t v = 0; v < n; ++v) dmax = std::max<int64_t>(dmax, degree((int32_t)v));
    std::vector<int64_t> first(dmax + 2, 0);
    for (int64_t v = 0; v < n; ++v) first[degree((int32_t)v) + 1]++;
    for (int64_t q = 0; q <= dmax; ++q) first[q + 1] += first[q];
    for (int64_t v = 0; v < n; ++v) by_deg[first[degree((int32_t)v)]++] = (int32_t)v;
  }
  std::vector<char> seen(n, 0);
  perm.clear();
  perm.reserve(n);
  std::vector<int32_t> nb;
  for (int32_t start : by_deg) {
    if (seen[start]) continue;
    seen[start] = 1;
    size_t head = perm.size();
    perm.push_back(start);
    while (head < perm.size()) {
      const int32_t v = perm[head++];
      nb.clear();
      for (int64_t e = ptr[v]; e < ptr[v + 1]; ++e) {
        const int32_t u = adj[e];
        if (!seen[u]) { seen[u] = 1; nb.push_back(u); }
      }
      if (sort_children) std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return degree(a) < degree(b) || (degree(a) == degree(b) && a < b); });
      perm.insert(perm.end(), nb.begin(), nb.end());
    }
  }
  std::reverse(perm.begin(), perm.end());
}

// Vertex renumbering used by every plan of this operator (square operators only):
// d_perm[new] = old, d_inv[old] = new.  Dense operands live on the device in the NEW order;
// pack/unpack translate.  The entry order inside a row never changes.
int glx_graph_ensure_order(glx_graph* g) {
  if (g->order_ready) return GLX_OK;
  g->order_ready = true;
  const int64_t n = g->n_rows;
  if (g->keep_order || g->n_rows != g->n_cols || n < 4096) return GLX_OK;
  if (g->d_src_col && g->h_col.empty() && g->nnz > 0) {      // a resident source: the pass below reads the pattern on the host
    g->h_col.resize(g->nnz);
    GLX_HIP(hipSetDevice(g->device));
    GLX_UP(glx_download_sync(g->h_col.data(), g->d_src_col, (size_t)g->nnz * 4, __func__));
  }
  const auto t_rcm0 = std::chrono::steady_clock::now();
  rcm_order(g, g->h_perm, true);
  if (getenv("GLX_TIMING")) fprintf(stderr, "[glx] locality order of %lld vertices: %.1f ms\n", (long long)n,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rcm0).count());
  g->h_inv.assign(n, 0);
  for (int64_t i = 0; i < n; ++i) g->h_inv[g->h_perm[i]] = (int32_t)i;
  GLX_HIP(hipSetDevice(g->device));
  GLX_POOL(glx_pool_alloc((void**)&g->d_perm, n * 4));
  GLX_POOL(glx_pool_alloc((void**)&g->d_inv, n * 4));
  GLX_UP(glx_upload_sync(g->d_perm, g->h_perm.data(), n * 4, __func__));
  GLX_UP(glx_upload_sync(g->d_inv, g->h_inv.data(), n * 4, __func__));
  return GLX_OK;
}

// Host helper of ssl.poisson's operator set-up for a symmetric W: row i of P = D^-1 W^T is row i of W scaled by
// scale[i] = 1/deg_i, with the entries in REVERSE order (the order scipy's csr_matmat leaves them in, which the sweep's
// bit-exactness depends on); deg_out[i] = the row sum in stored order from 0.0 (scipy's csr_matvec with a vector of ones).
// Plain loops over the CSR arrays: the numpy formulation of the same thing cost 8 ms at 70 000 vertices.
extern "C" int glx_host_row_sums(int64_t n, const int32_t* rowptr, const double* val, double* sum_out) {
#pragma clang fp contract(off)
  GLX_CHECK(rowptr && val && sum_out, GLX_EINVAL, "glx_host_row_sums: null argument");
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) s = s + val[e] * 1.0;
    sum_out[i] = s;
  }
  return GLX_OK;
}

// The nonzero rows of -L[:, cols] * F (ssl.laplace's right-hand side, reference ssl.py:1236) from the CSC image of L: scipy forms the
// CSR matrix -L[:, cols] and csr_matvecs adds a row's products one after another from 0, i.e. for canonical L and distinct cols the
// terms (-l_ij) * F[pos(j), :] in ascending j.  Rows listed in `cols` themselves are left out (they are not part of the Dirichlet
// system), every other row comes out times row_scale[row] (M*b, ssl.py:1249).  *count_out = -1: duplicate columns, the caller falls
// back to the literal expression.
extern "C" int glx_host_neg_columns_rows(int64_t n, const int32_t* cptr, const int32_t* crow, const double* cval, int64_t m, const int64_t* cols,
                                         const double* F, int k, const double* row_scale, int64_t cap, int32_t* rows_out, double* vals_out,
                                         int64_t* count_out) {
#pragma clang fp contract(off)
  GLX_CHECK(cptr && crow && cval && F && rows_out && vals_out && count_out && (cols || m == 0), GLX_EINVAL, "glx_host_neg_columns_rows: null argument");
  GLX_CHECK(n >= 0 && m >= 0 && k > 0, GLX_EINVAL, "glx_host_neg_columns_rows: bad size");
  for (int64_t q = 0; q < m; ++q) GLX_CHECK(cols[q] >= 0 && cols[q] < n, GLX_EINVAL, "glx_host_neg_columns_rows: column %lld out of range", (long long)cols[q]);
  std::vector<int64_t> pos((size_t)m);
  for (int64_t q = 0; q < m; ++q) pos[q] = q;
  std::stable_sort(pos.begin(), pos.end(), [&](int64_t a, int64_t b) { return cols[a] < cols[b]; });
  for (int64_t q = 1; q < m; ++q)
    if (cols[pos[q]] == cols[pos[q - 1]]) { *count_out = -1; return GLX_OK; }
  static thread_local std::vector<int32_t> slot;       // row -> slot of this call, -1 elsewhere (only touched entries are reset)
  if ((int64_t)slot.size() < n) slot.assign((size_t)n, -1);
  std::vector<int32_t> touched;
  std::vector<double> acc;
  for (int64_t q = 0; q < m; ++q) {
    const int64_t j = cols[pos[q]];
    const double* f = F + (size_t)pos[q] * k;
    for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
      const int32_t row = crow[e];
      int32_t sl = slot[row];
      if (sl < 0) {
        sl = (int32_t)touched.size();
        slot[row] = sl;
        touched.push_back(row);
        acc.resize(acc.size() + (size_t)k, 0.0);
      }
      const double nv = -cval[e];
      double* a = acc.data() + (size_t)sl * k;
      for (int c = 0; c < k; ++c) a[c] = a[c] + nv * f[c];
    }
  }
  std::vector<int32_t> ord(touched);
  std::sort(ord.begin(), ord.end());
  for (int64_t q = 0; q < m; ++q) {                    // the labelled rows themselves are not part of the system
    const int32_t sl = slot[cols[q]];
    if (sl >= 0) slot[cols[q]] = -2 - sl;
  }
  int64_t cnt = 0;
  int rc = GLX_OK;
  for (int32_t row : ord) {
    const int32_t sl = slot[row];
    if (sl < 0) continue;
    if (cnt >= cap) { rc = GLX_EINVAL; glx_set_error("glx_host_neg_columns_rows: more than %lld rows", (long long)cap); break; }
    const double sc = row_scale ? row_scale[row] : 1.0;
    const double* a = acc.data() + (size_t)sl * k;
    rows_out[cnt] = row;
    for (int c = 0; c < k; ++c) vals_out[(size_t)cnt * k + c] = row_scale ? sc * a[c] : a[c];
    ++cnt;
  }
  for (int32_t row : touched) slot[row] = -1;
  *count_out = cnt;
  return rc;
}

extern "C" int glx_host_reverse_scale_rows(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, const double* scale,
                                           int32_t* col_out, double* val_out) {
#pragma clang fp contract(off)
  GLX_CHECK(rowptr && col && val && scale && col_out && val_out, GLX_EINVAL, "glx_host_reverse_scale_rows: null argument");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(8, n / 16384));
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t a = rowptr[i], b = rowptr[i + 1];
      const double sc = scale[i];
      for (int64_t e = a; e < b; ++e) {
        const int64_t src = a + (b - 1 - e);
        col_out[e] = col[src];
        val_out[e] = sc * val[src];
      }
    }
  };
  if (nt == 1) {
    work(0, n);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
  }
  return GLX_OK;
}

// The library's locality order (the breadth-first / reverse Cuthill-McKee pass of rcm_order) for a pattern the caller holds on the
// host, restricted to the columns [col_lo, col_lo + n): a rank of the vertex-partitioned sweep orders ITS rows by their links
// among themselves (halo columns ignored) -- the rectangular rank-local operator is never renumbered by the library.
// perm_out[new] = old.
extern "C" int glx_host_locality_order(int64_t n, const int32_t* rowptr, const int32_t* col, int64_t col_lo, int32_t* perm_out) {
  GLX_CHECK(n >= 0 && rowptr && col && perm_out, GLX_EINVAL, "glx_host_locality_order: null argument");
  std::vector<int32_t> rp(n + 1, 0), cc;
  cc.reserve((size_t)rowptr[n]);
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
      const int64_t j = (int64_t)col[e] - col_lo;
      if (j >= 0 && j < n) cc.push_back((int32_t)j);
    }
    rp[i + 1] = (int32_t)cc.size();
  }
  std::vector<int32_t> perm;
  rcm_order_arrays(n, rp.data(), cc.data(), perm, true);
  GLX_CHECK((int64_t)perm.size() == n, GLX_EINVAL, "glx_host_locality_order: internal error");
  memcpy(perm_out, perm.data(), (size_t)n * 4);
  return GLX_OK;
}

// rows of a CSR matrix in another order (row i of the result = row perm[i] of the input, entries in their stored order), on host
// threads: scipy's fancy row indexing takes seconds at 2 x 10^8 entries
extern "C" int glx_host_permute_rows(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, const int64_t* perm,
                                     int32_t* rowptr_out, int32_t* col_out, double* val_out) {
  GLX_CHECK(n >= 0 && rowptr && col && val && perm && rowptr_out && col_out && val_out, GLX_EINVAL, "glx_host_permute_rows: null argument");
  rowptr_out[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    GLX_CHECK(perm[i] >= 0 && perm[i] < n, GLX_EINVAL, "glx_host_permute_rows: row %lld out of range", (long long)perm[i]);
    const int64_t len = rowptr[perm[i] + 1] - rowptr[perm[i]];
    GLX_CHECK((int64_t)rowptr_out[i] + len < (1ll << 31), GLX_EUNSUPPORTED, "glx_host_permute_rows: more than 2^31 entries");
    rowptr_out[i + 1] = rowptr_out[i] + (int32_t)len;
  }
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(16, n / 65536));
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t a = rowptr[perm[i]], len = rowptr[perm[i] + 1] - a, o = rowptr_out[i];
      memcpy(col_out + o, col + a, (size_t)len * 4);
      memcpy(val_out + o, val + a, (size_t)len * 8);
    }
  };
  if (nt == 1) {
    work(0, n);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
  }
  return GLX_OK;
}

// The caller's own vertex order (perm[new] = old) instead of the library's reverse Cuthill-McKee pass: whoever built the graph
// may know a better one -- weightmatrix.knn has the FEATURES in hand, and an order by a tree over feature space keeps
// neighbours close without looking at the graph at all.  Before the operator is first used.
extern "C" int glx_graph_set_order(glx_graph* g, const int32_t* perm) {
  GLX_CHECK(g, GLX_EINVAL, "glx_graph_set_order: null graph");
  if (!perm) return graph_keep_order(g);      // the caller's order as it is: no renumbering at all
  GLX_CHECK(g->plans.empty() && !g->order_ready, GLX_EINVAL, "glx_graph_set_order: call before the operator is first used");
  GLX_CHECK(g->n_rows == g->n_cols, GLX_EINVAL, "glx_graph_set_order: operator must be square");
  const int64_t n = g->n_rows;
  std::vector<int32_t> inv(n, -1);
  for (int64_t i = 0; i < n; ++i) {
    GLX_CHECK(perm[i] >= 0 && perm[i] < n && inv[perm[i]] < 0, GLX_EINVAL, "glx_graph_set_order: not a permutation at position %lld", (long long)i);
    inv[perm[i]] = (int32_t)i;
  }
  g->h_perm.assign(perm, perm + n);
  g->h_inv.swap(inv);
  g->order_ready = true;
  GLX_HIP(hipSetDevice(g->device));
  GLX_POOL(glx_pool_alloc((void**)&g->d_perm, std::max<size_t>(n * 4, 4)));
  GLX_POOL(glx_pool_alloc((void**)&g->d_inv, std::max<size_t>(n * 4, 4)));
  // through a work set's stream and its page-locked staging: a blocking hipMemcpy runs on the NULL stream, whose copy queue the first
  // such call of a process creates (9 ms of a fresh model's first fit_predict)
  glx_work* w = nullptr;
  int rc = glx_work_acquire(g->device, &w);
  if (rc) return rc;
  struct WorkGuard { glx_work* w; ~WorkGuard() { hipStreamSynchronize(w->stream); glx_work_release(w); } } wguard{w};
  GLX_UP(glx_upload(g->d_perm, g->h_perm.data(), (size_t)n * 4, w->stream, __func__));
  GLX_UP(glx_upload(g->d_inv, g->h_inv.data(), (size_t)n * 4, w->stream, __func__));
  GLX_HIP(hipStreamSynchronize(w->stream));
  return GLX_OK;
}

extern "C" int glx_graph_order(glx_graph* g, int32_t* perm_out) {
  GLX_CHECK(g && perm_out, GLX_EINVAL, "glx_graph_order: null argument");
  int rc = glx_graph_ensure_order(g);
  if (rc) return rc;
  for (int64_t i = 0; i < g->n_rows; ++i) perm_out[i] = g->h_perm.empty() ? (int32_t)i : g->h_perm[i];
  return GLX_OK;
}


// The sliced-ELL image is assembled ON THE DEVICE (round 3): the CSR arrays go up as they are and one wavefront per slice
// writes the slice's chunks -- image position (chunk k, lane w) holds entry jj = (k S + seg) 4 + t of the row in slot w / 4
// (G = 4: seg = slot within the row's S slots, t = w % 4; wider rows: jj = k G + w % G), or a zero for positions past the end
// of the row -- so nothing needs a memset and the host never touches the 1.3 M (... 2 x 10^8) entries: the host fill took 6 ms at
// 70 000 vertices with 16 threads (300 ms at 18 M entries), more than everything else of the plan together.
template <typename T>
__global__ __launch_bounds__(256) void sell_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ccol,
                                                        const double* __restrict__ cval, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ inv, const int32_t* __restrict__ slot_row,
                                                        const int32_t* __restrict__ slot_len, const SliceHdr* __restrict__ hdr,
                                                        int64_t nslices, int64_t head, int G, int32_t* __restrict__ col, T* __restrict__ val,
                                                        const double* __restrict__ row_scale, int reverse) {
#pragma clang fp contract(off)
  // row_scale / reverse (resident sources, glx_graph_set_row_transform): entry jj of the operator's row is entry len - 1 - jj of
  // the source row, times row_scale[row] -- one rounding, the host expression `scale[i] * w` of the reference's D^-1 W^T
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const SliceHdr h = hdr[s];
  const int S = h.S & 0xff, R = 64 / G;
  const int slot = lane / G, t = lane % G;
  const int32_t row = slot_row[s * R + slot];
  const int len = slot_len[s * R + slot];
  const int seg = slot & (S - 1);
  const int32_t orow = row >= 0 ? (perm ? perm[row] : row) : 0;
  const int64_t b = row >= 0 ? rowptr[orow] : 0;
  const double sc = (row >= 0 && row_scale) ? row_scale[orow] : 1.0;
  for (int k = 0; k < h.nchunks || k == 0; ++k) {          // chunk 0 always exists (the dense head region)
    const int jj = G == 4 ? (k * S + seg) * 4 + t : k * G + t;
    int32_t c = 0;
    double v = 0.0;
    if (row >= 0 && jj < len) {
      const int64_t src = reverse ? b + (len - 1 - jj) : b + jj;
      c = ccol[src];
      if (inv) c = inv[c];
      v = cval[src];
      if (row_scale) v = sc * v;
    }
    const int64_t idx = k == 0 ? s * 64 + lane : head + h.ptr + (int64_t)(k - 1) * 64 + lane;
    col[idx] = c;
    val[idx] = (T)v;
  }
}

// Build (once per G) the sliced-ELL image of the operator.  The (renumbered) rows are cut
// into 8 contiguous id ranges, one per XCD; inside a range rows are handed to wavefront slices
// in order of decreasing length (longest first: LPT balance, little padding inside a slice);
// the ENTRY order inside a row is untouched.  G = 4 plans split rows longer than L1 entries
// over S = 4 slots and rows longer than L4 over S = 16 slots.  Every range
// is padded with empty slices to the same number of 4-slice blocks, so block b serves range
// b % 8 -- the XCD the dispatcher is observed to place it on.
int glx_graph_plan(glx_graph* g, int G, SellPlan** out, bool relaxed) {
  if (G != 4) relaxed = false;      // (segments exist in G = 4 plans only)
  for (auto& p : g->plans)
    if (p.G == G && p.relaxed == relaxed) {
      *out = &p;
      return GLX_OK;
    }
  int rc = glx_graph_ensure_order(g);
  if (rc) return rc;
  GLX_HIP(hipSetDevice(g->device));
  const auto t_plan0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (getenv("GLX_TIMING")) fprintf(stderr, "[glx] plan G=%d, %s: %.1f ms since the plan started\n", G, what,
                                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count());
  };
  const int R = 64 / G;
  const int64_t n = g->n_rows;
  const bool renum = !g->h_perm.empty();
  int L1 = 24, L4 = 96;   // measured on the 70k k=10 graph (fp64): 24/96 13.3 us, 32/128 15.2 us, 16/64 15.4 us
  // Round 3: once the vertex records no longer fit the L2s (8 x 4 MB) the balance tips the other way -- every gather is a trip to
  // the Infinity Cache / HBM whatever the slice shape, and what counts is fewer, longer slices (a row of 25 entries should not take
  // four quarter-filled slots) and less padding: n = 1e6 (d = 64): 277.1 us with 24/96, 258.5 with 40/160, 248.6 with 64/256
  // (-10 %), 250.9 with 96/384; n = 1e7: 4020 -> 3887 us; at 70 000 vertices 64/256 costs 18.1 us against 13.1
  // (profiles/r03_slot_thresholds.txt).
  // Relaxed plans (tolerance-mode CG: a row's entries may be added in any fixed order, its segments sum independently and are
  // combined once behind the chunk loop): what counts is the number of wavefronts, every one of which pays its header loads, its
  // epilogue and its share of the column dots -- fewer, longer slices win.  Config 3 (60 000 vertices, rows of 21..660 entries),
  // SpMM + dots per launch: 8/64 37.6 us, 24/96 32.2, 32/128 27.3, 64/256 26.5, 96/384 32.6 (profiles/r04_cg_slot_thresholds.txt)
  if (relaxed) { L1 = 64; L4 = 256; }
#ifndef GLX_SELL_WINDOW_MIN_MB
#define GLX_SELL_WINDOW_MIN_MB 32      // records beyond the eight L2s (32 MB)
#endif
  const bool big_state = (double)g->n_cols * G * 4.0 * (g->dtype == GLX_F64 ? 8.0 : 4.0) >= (double)GLX_SELL_WINDOW_MIN_MB * 1024 * 1024;
  if (big_state) { L1 = 64; L4 = 256; }
  const bool window_state = big_state;
  const int64_t sigma_big = 32768;      // measured at n = 10^6 (fp64, us per sweep): whole range 255, 8192: 255, 16384: 252, 32768: 229, 65536: 239, 131072: 253
  bool windowed = false;
  if (G != 4) { L1 = 1 << 30; L4 = 1 << 30; }
  auto old_of = [&](int64_t nid) -> int64_t { return renum ? g->h_perm[nid] : nid; };
  // (row lengths in the new numbering, looked up ~6 times per row below: one pass through the permutation instead of six)
  std::vector<int32_t> len_new((size_t)n);
  for (int64_t nid = 0; nid < n; ++nid) { const int64_t o = old_of(nid); len_new[nid] = (int32_t)(g->h_rowptr[o + 1] - g->h_rowptr[o]); }
  auto rowlen = [&](int64_t nid) -> int { return len_new[nid]; };
  auto klass = [&](int len) { return len > L4 ? 16 : (len > L1 ? 4 : 1); };

  const int NX = 8;
  std::vector<std::vector<SliceHdr>> ghdr(NX);
  std::vector<std::vector<int32_t>> grow(NX), glen(NX);
  // the eight id ranges carry equal WORK, not equal row counts (equal rows, the round-2 rule: 13.69 -> 13.10 us per sweep at config 2, profiles/r03_xcd_balance.txt): the launch ends
  // when the slowest XCD does, and under a locality order the rows' lengths drift along the order (a cluster's dense core
  // first, its fringe last).  Work of a row = its entries + XCD_ROW_COST (header, epilogue, store).
  std::vector<int64_t> xb(NX + 1, 0);
  {
    const bool by_work = n >= NX;
    const int64_t row_cost = 3;
    int64_t total = 0;
    if (by_work) for (int64_t i = 0; i < n; ++i) total += rowlen(i) + row_cost;
    int64_t acc = 0, i = 0;
    for (int x = 1; x < NX; ++x) {
      if (!by_work) { xb[x] = n * x / NX; continue; }
      const int64_t want = total * x / NX;
      while (i < n && acc < want) { acc += rowlen(i) + row_cost; ++i; }
      xb[x] = i;
    }
    xb[NX] = n;
  }
  for (int x = 0; x < NX; ++x) {
    const int64_t b0 = xb[x], b1 = xb[x + 1];
    const int64_t m = b1 - b0;
    std::vector<int32_t> order(m);
    grow[x].reserve((size_t)m + m / 4 + 4 * R);
    glen[x].reserve((size_t)m + m / 4 + 4 * R);
    ghdr[x].reserve((size_t)m / (R > 0 ? R : 1) + 16);
    // sort by decreasing length inside windows of `sigma` consecutive ids (SELL-C-sigma): small
    // windows keep rows that share neighbours in the same wavefronts/CUs (L1 reuse), large ones
    // minimise padding.  The whole XCD range is one window.
    // Round 5: once the records no longer fit the L2s the window must be SMALL.  The wavefronts resident on an XCD at one time cover
    // ~10^4 consecutive rows of the slice order; sorted by length over the whole range those rows come from all over the XCD's share
    // of the vertex order (16 MB of records at 10^6 vertices against a 4 MB L2) and their gathers miss -- counters at n = 10^6:
    // 1.82 GB per sweep against 0.57 GB for the distinct records of the eight ranges (profiles/r05_scale_1e6_pmc.txt).  Sorted inside
    // windows of `sigma` consecutive rows the resident wavefronts work on one stretch of the locality order at a time: -10 % at 32768
    // rows per window.  (Smaller windows gain nothing: inside a cluster of the blob data the kNN graph is an expander -- a window's
    // neighbours are spread over its whole cluster, 13 MB of records, whatever the window's size.)
    int64_t sigma = window_state ? sigma_big : m;
    if (sigma < m) windowed = true;
    // With windows the LONG rows (those split over 4 or 16 slots: chains of dependent round trips) still come first, longest first,
    // whatever their window: they are few and they are what a launch's tail consists of (without this the windows cost 20 % at
    // 3 x 10^5 vertices, where an XCD's range is hardly more than one window).
    int64_t nlong = 0;
    std::vector<int32_t> rest;
    if (sigma < m) {
      std::vector<int32_t> lng;
      rest.reserve(m);
      for (int64_t i = 0; i < m; ++i) (klass(rowlen(b0 + i)) > 1 ? lng : rest).push_back((int32_t)(b0 + i));
      std::stable_sort(lng.begin(), lng.end(), [&](int32_t a, int32_t b) { return rowlen(a) > rowlen(b); });
      nlong = (int64_t)lng.size();
      std::copy(lng.begin(), lng.end(), order.begin());
    }
    for (int64_t w0 = 0; w0 < m - nlong; w0 += sigma) {
      const int64_t w1 = std::min(m - nlong, w0 + sigma);
      auto rowat = [&](int64_t i) -> int64_t { return nlong || sigma < m ? (int64_t)rest[i] : b0 + i; };
      std::vector<int64_t> cnt(g->max_row + 2, 0);
      for (int64_t i = w0; i < w1; ++i) cnt[g->max_row - rowlen(rowat(i)) + 1]++;
      for (size_t b = 1; b < cnt.size(); ++b) cnt[b] += cnt[b - 1];
      for (int64_t i = w0; i < w1; ++i) order[nlong + w0 + cnt[g->max_row - rowlen(rowat(i))]++] = (int32_t)rowat(i);
    }
    int64_t pos = 0;
    while (pos < m) {
      const int S = klass(rowlen(order[pos]));
      SliceHdr h;
      h.ptr = 0;
      h.S = S;
      int width = 0, narrow = 1 << 30;
      for (int r = 0; r < R / S; ++r) {
        int32_t row = -1;
        int len = 0;
        if (pos < m && klass(rowlen(order[pos])) == S) {
          row = order[pos];
          len = rowlen(row);
          ++pos;
        }
        for (int sgm = 0; sgm < S; ++sgm) { grow[x].push_back(row); glen[x].push_back(len); }
        width = std::max(width, len);
        narrow = std::min(narrow, len);      // (an empty slot has len 0)
      }
      h.nchunks = G == 4 ? (width + 4 * S - 1) / (4 * S) : (width + G - 1) / G;
      // G = 4: chunks k < full hold four real entries in EVERY slot of the slice -- the kernel gathers them without predicates
      // (rows are sorted by length, so a slice's rows are nearly equally long and most chunks are full); bits 8.. of S
      if (G == 4) h.S = S | ((narrow / (4 * S)) << 8);
      ghdr[x].push_back(h);
    }
    // longest-running slices first (the launch ends when the last wavefront does): a slice's time is its gather rounds times the
    // cost of a round in its class -- measured per chunk at config 3: 1.25 / 2.5 / 3.6 us for S = 1 / 4 / 16 when the running sum
    // hops between segments, one price for all when it does not (relaxed).  Not with windows: the order of the windows IS the point
    // (and with thousands of slices per XCD the tail of a launch does not matter).
    if (!windowed) {
      const size_t ns = ghdr[x].size();
      std::vector<int32_t> idx(ns);
      for (size_t q = 0; q < ns; ++q) idx[q] = (int32_t)q;
      auto cost = [&](const SliceHdr& h) {
        const int S = h.S & 0xff;
        return (int64_t)h.nchunks * (relaxed ? 4 : (S == 1 ? 4 : (S == 4 ? 8 : 12)));
      };
      std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return cost(ghdr[x][a]) > cost(ghdr[x][b]); });
      std::vector<SliceHdr> h2(ns);
      std::vector<int32_t> r2(ns * R), l2(ns * R);
      for (size_t q = 0; q < ns; ++q) {
        h2[q] = ghdr[x][idx[q]];
        for (int r = 0; r < R; ++r) {
          r2[q * R + r] = grow[x][(size_t)idx[q] * R + r];
          l2[q * R + r] = glen[x][(size_t)idx[q] * R + r];
        }
      }
      ghdr[x].swap(h2);
      grow[x].swap(r2);
      glen[x].swap(l2);
    }
  }
  lap("slices formed");
  int64_t bpx = 0;   // blocks (GLX_WPB slices) per XCD range
  for (int x = 0; x < NX; ++x) bpx = std::max<int64_t>(bpx, ((int64_t)ghdr[x].size() + GLX_WPB - 1) / GLX_WPB);
  const int64_t spx = bpx * GLX_WPB;
  const int64_t nslices = spx * NX;
  std::vector<SliceHdr> hdr(nslices);
  std::vector<int32_t> slot_row(nslices * R, -1), slot_len(nslices * R, 0);
  int64_t stored = 0;
  for (int x = 0; x < NX; ++x)
    for (int64_t s = 0; s < spx; ++s) {
      SliceHdr h;
      h.ptr = stored;
      h.nchunks = 0;
      h.S = 1;
      if (s < (int64_t)ghdr[x].size()) {
        h = ghdr[x][s];
        h.ptr = stored;
        for (int r = 0; r < R; ++r) {
          slot_row[(x * spx + s) * R + r] = grow[x][s * R + r];
          slot_len[(x * spx + s) * R + r] = glen[x][s * R + r];
        }
      }
      stored += (int64_t)(h.nchunks > 0 ? h.nchunks - 1 : 0) * 64;   // chunk 0 lives in the dense head arrays
      hdr[x * spx + s] = h;
    }
  // image = [head: chunk 0 of every slice, nslices*64 entries][tail: chunks 1.. at hdr.ptr], filled by sell_fill_kernel
  const int64_t head = nslices * 64;
  SellPlan p;
  p.G = G;
  p.R = R;
  p.relaxed = relaxed;
  p.nslices = nslices;
  p.stored = head + stored;
  p.head = head;
  const size_t es = g->dtype == GLX_F64 ? 8 : 4;
  GLX_POOL(glx_pool_alloc((void**)&p.d_slot_row, std::max<size_t>(4, slot_row.size() * 4)));
  GLX_POOL(glx_pool_alloc((void**)&p.d_slot_len, std::max<size_t>(4, slot_len.size() * 4)));
  GLX_POOL(glx_pool_alloc((void**)&p.d_slice_hdr, std::max<size_t>(16, hdr.size() * sizeof(SliceHdr))));
  GLX_POOL(glx_pool_alloc((void**)&p.d_col, std::max<size_t>(4, (head + stored) * 4)));
  GLX_POOL(glx_pool_alloc((void**)&p.d_val, std::max<size_t>(8, (head + stored) * es)));
  glx_work* pw = nullptr;              // uploads and the fill run in a work set's stream (no blocking copies on the NULL stream)
  {
    int rcw = glx_work_acquire(g->device, &pw);
    if (rcw) return rcw;
  }
  struct WorkGuard { glx_work* w; ~WorkGuard() { hipStreamSynchronize(w->stream); glx_work_release(w); } } pwguard{pw};
  hipStream_t pst = pw->stream;
  GLX_UP(glx_upload(p.d_slot_row, slot_row.data(), slot_row.size() * 4, pst, __func__));
  GLX_UP(glx_upload(p.d_slot_len, slot_len.data(), slot_len.size() * 4, pst, __func__));
  GLX_UP(glx_upload(p.d_slice_hdr, hdr.data(), hdr.size() * sizeof(SliceHdr), pst, __func__));
  if (nslices > 0) {
    // the CSR arrays as they are: resident on the device already (glx_graph_create_resident), or uploaded into work buffers from
    // the pool that are released after the fill
    int32_t *d_rp = g->d_src_rowptr, *d_cc = g->d_src_col;
    double* d_cv = g->d_src_val;
    struct Tmp { void *a = nullptr, *b = nullptr, *c = nullptr; ~Tmp() { glx_pool_free(a); glx_pool_free(b); glx_pool_free(c); } } tmp;
    if (!d_cc) {
      int rc2 = glx_pool_alloc(&tmp.a, (size_t)(n + 1) * 4);
      if (!rc2) rc2 = glx_pool_alloc(&tmp.b, std::max<size_t>((size_t)g->nnz * 4, 4));
      if (!rc2) rc2 = glx_pool_alloc(&tmp.c, std::max<size_t>((size_t)g->nnz * 8, 8));
      if (rc2) return rc2;
      d_rp = (int32_t*)tmp.a; d_cc = (int32_t*)tmp.b; d_cv = (double*)tmp.c;
      GLX_UP(glx_upload(d_rp, g->h_rowptr.data(), (size_t)(n + 1) * 4, pst, __func__));
      if (g->nnz > 0) {
        GLX_UP(glx_upload(d_cc, g->h_col.data(), (size_t)g->nnz * 4, pst, __func__));
        GLX_UP(glx_upload(d_cv, g->h_val.data(), (size_t)g->nnz * 8, pst, __func__));
      }
    }
    const unsigned grid = (unsigned)((nslices + 3) / 4);
    if (g->dtype == GLX_F64)
      hipLaunchKernelGGL(sell_fill_kernel<double>, dim3(grid), dim3(256), 0, pst, (const int32_t*)d_rp, (const int32_t*)d_cc, (const double*)d_cv,
                         (const int32_t*)(renum ? g->d_perm : nullptr), (const int32_t*)(renum ? g->d_inv : nullptr), (const int32_t*)p.d_slot_row,
                         (const int32_t*)p.d_slot_len, (const SliceHdr*)p.d_slice_hdr, nslices, head, G, p.d_col, (double*)p.d_val,
                         (const double*)g->d_row_scale, g->reverse_rows ? 1 : 0);
    else
      hipLaunchKernelGGL(sell_fill_kernel<float>, dim3(grid), dim3(256), 0, pst, (const int32_t*)d_rp, (const int32_t*)d_cc, (const double*)d_cv,
                         (const int32_t*)(renum ? g->d_perm : nullptr), (const int32_t*)(renum ? g->d_inv : nullptr), (const int32_t*)p.d_slot_row,
                         (const int32_t*)p.d_slot_len, (const SliceHdr*)p.d_slice_hdr, nslices, head, G, p.d_col, (float*)p.d_val,
                         (const double*)g->d_row_scale, g->reverse_rows ? 1 : 0);
    GLX_HIP(hipGetLastError());
    GLX_HIP(hipStreamSynchronize(pst));       // the pooled CSR copies go back before anybody else may draw them
  }
  GLX_HIP(hipStreamSynchronize(pst));         // the host vectors of the plan may go
  lap("uploaded");
  g->plans.push_back(p);
  *out = &g->plans.back();
  return GLX_OK;
}
