// Round 6: the parity failure of the randomised soak is a host-to-device copy that delivers wrong bytes -- hipMemcpyAsync from pageable host
// memory on a non-blocking stream, a dozen processes sharing the GPU: a run of ~1 KB ending at a 4 KB boundary of the buffer differs from the
// source (csrc/knn.hip: glx_debug_set(1) caught it twice in 12 500 uploads).  This probe isolates the copy: every process loops
//   fill a host buffer with words that encode (iteration, word index) -> copy to the device -> read back through page-locked memory -> compare
// and reports what the wrong words contain (zeros / words of an EARLIER iteration = a stale staging buffer / other).
//   hipcc --offload-arch=gfx950 -O2 -w -o /tmp/h2d_probe scripts/probes/h2d_pageable_probe.hip
//   for i in $(seq 12); do /tmp/h2d_probe MODE SECONDS & done; wait
// MODE: async  = hipMemcpyAsync(pageable) on a non-blocking stream + hipStreamSynchronize            (what libglx did)
//       sync   = hipMemcpy(pageable)
//       staged = memcpy into a page-locked block of the process, hipMemcpyAsync from there             (the fix)
//       reg    = hipHostRegister the source, hipMemcpyAsync, hipHostUnregister
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <chrono>
#include <vector>

static inline unsigned long long word_of(unsigned long long it, unsigned long long i, unsigned pid) { return (it << 32) ^ (i * 0x9e3779b97f4a7c15ull) ^ ((unsigned long long)pid << 48); }

__global__ void touch_kernel(unsigned long long* p, long n) {     // a little device work between the copies, like a search's kernels
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] ^= 0ull;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "async";
  const double seconds = argc > 2 ? atof(argv[2]) : 30.0;
  const unsigned pid = (unsigned)getpid();
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const size_t cap = (size_t)6 << 20;
  void* dev[3];
  for (int q = 0; q < 3; ++q) hipMalloc(&dev[q], cap);
  unsigned long long* back = nullptr;
  hipHostMalloc((void**)&back, cap, hipHostMallocDefault);
  unsigned long long* stage = nullptr;
  hipHostMalloc((void**)&stage, cap, hipHostMallocDefault);
  srand(pid);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long it = 0, bad_copies = 0, bad_words = 0, zeros = 0, stale = 0, other = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    ++it;
    const size_t words = (size_t)(8192 + rand() % (cap / 8 - 8192));
    const size_t bytes = words * 8;
    unsigned long long* src = (unsigned long long*)malloc(bytes + 64);      // a fresh pageable array per copy, like a new numpy array
    unsigned long long* s = (unsigned long long*)((char*)src + (rand() % 2) * 16);
    for (size_t i = 0; i < words; ++i) s[i] = word_of(it, i, pid);
    void* d = dev[it % 3];
    hipError_t e = hipSuccess;
    if (!strcmp(mode, "async")) {
      e = hipMemcpyAsync(d, s, bytes, hipMemcpyDefault, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    } else if (!strcmp(mode, "sync")) {
      e = hipMemcpy(d, s, bytes, hipMemcpyHostToDevice);
    } else if (!strcmp(mode, "staged")) {
      memcpy(stage, s, bytes);
      e = hipMemcpyAsync(d, stage, bytes, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
    } else {
      e = hipHostRegister(s, bytes, hipHostRegisterDefault);
      if (e == hipSuccess) e = hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      hipHostUnregister(s);
    }
    if (e != hipSuccess) { printf("pid %u: copy failed: %s\n", pid, hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(touch_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (unsigned long long*)d, (long)words);
    hipStreamSynchronize(st);
    hipMemcpy(back, d, bytes, hipMemcpyDeviceToHost);
    size_t nb = 0, first = 0, last = 0;
    for (size_t i = 0; i < words; ++i) {
      if (back[i] != s[i]) {
        if (!nb) first = i;
        last = i;
        ++nb;
        if (back[i] == 0) ++zeros;
        else {
          bool st_found = false;
          for (unsigned long long back_it = 1; back_it <= 8 && back_it < it; ++back_it)
            if (back[i] == word_of(it - back_it, i, pid)) { st_found = true; break; }
          if (st_found) ++stale; else ++other;
        }
      }
    }
    if (nb) {
      ++bad_copies;
      bad_words += nb;
      if (bad_copies <= 6)
        printf("pid %u mode %s: copy %llu of %zu bytes (source %p, device %p): %zu words differ, bytes %zu .. %zu (source page offsets %zx .. %zx); e.g. got %016llx expected %016llx\n",
               pid, mode, it, bytes, (void*)s, d, nb, first * 8, last * 8 + 7, ((size_t)s + first * 8) & 0xfff, ((size_t)s + last * 8 + 7) & 0xfff, back[first], s[first]);
    }
    free(src);
  }
  printf("pid %u mode %s: %llu copies, %llu with wrong bytes (%llu words: %llu zero, %llu = the same word of one of the 8 copies before, %llu other)\n", pid, mode, it, bad_copies,
         bad_words, zeros, stale, other);
  return 0;
}
