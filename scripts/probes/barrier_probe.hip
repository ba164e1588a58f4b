// Developer probe: cost of a device-wide barrier + cross-XCD visibility inside one persistent kernel.
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/barrier_probe scripts/probes/barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Bar { unsigned* xcd; unsigned* glob; int* abort_flag; };   // xcd: 8 counters, 64 B apart

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, int* abort_flag) {
  long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > 100000000ll) { *abort_flag = 1; return false; }   // 1 s at 100 MHz
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  }
  return true;
}

// mode 0: one counter; mode 1: per-XCD counter, last arriver of each XCD bumps the global one
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned epoch, int mode, unsigned nblocks) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (mode == 0) {
      __hip_atomic_fetch_add(b.glob, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      ok = spin_until(b.glob, epoch * nblocks, b.abort_flag);
    } else {
      const unsigned x = blockIdx.x & 7;
      const unsigned nx = (nblocks - x + 7) / 8;   // blocks with blockIdx % 8 == x
      const unsigned old = __hip_atomic_fetch_add(b.xcd + x * 16, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * nx) __hip_atomic_fetch_add(b.glob, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      ok = spin_until(b.glob, epoch * 8u, b.abort_flag);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default for HIP device code? use builtin below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

__global__ void persist(double* buf, size_t n, int iters, Bar b, int mode, int payload, unsigned long long* errs) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nth = (size_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (int it = 1; it <= iters; ++it) {
    double* cur = buf + (size_t)(it & 1) * n;
    const double* prev = buf + (size_t)((it - 1) & 1) * n;
    if (payload) {
      for (size_t i = tid; i < n; i += nth) {
        // read something another block (other XCD) wrote in the previous sweep
        const size_t j = (i + (size_t)blockDim.x * 3 + 17) % n;
        if (it > 1 && prev[j] != (double)(it - 1)) ++bad;
        cur[i] = (double)it;
      }
    }
    if (!grid_barrier(b, (unsigned)it, mode, gridDim.x)) break;
  }
  if (bad) atomicAdd(errs, bad);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s CUs=%d\n", prop.name, prop.multiProcessorCount);
  const size_t n = 70000 * 16;   // one fp64 record array of the 70k sweep
  double* buf; CK(hipMalloc(&buf, 2 * n * 8)); CK(hipMemset(buf, 0, 2 * n * 8));
  unsigned* ctr; CK(hipMalloc(&ctr, 4096));
  int* ab; CK(hipMalloc(&ab, 4));
  unsigned long long* errs; CK(hipMalloc(&errs, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int payload = 0; payload < 2; ++payload)
  for (int mode = 0; mode < 2; ++mode)
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int nb = cfg == 0 ? 256 : cfg == 1 ? 512 : cfg == 2 ? 1024 : 2048;
    const int bs = cfg == 0 ? 1024 : cfg == 1 ? 512 : cfg == 2 ? 256 : 128;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persist, bs, 0));
    if ((long)occ * prop.multiProcessorCount < nb) { printf("skip nb=%d bs=%d (occupancy %d)\n", nb, bs, occ); continue; }
    CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(ab, 0, 4)); CK(hipMemset(errs, 0, 8));
    Bar b{ctr + 64, ctr, ab};
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(persist, dim3(nb), dim3(bs), 0, 0, buf, n, iters, b, mode, payload, errs);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int habort; unsigned long long herr;
    CK(hipMemcpy(&habort, ab, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, errs, 8, hipMemcpyDeviceToHost));
    printf("payload=%d mode=%d blocks=%4d x %4d thr: %.2f us per sweep+barrier  abort=%d stale_reads=%llu\n", payload, mode, nb, bs,
           ms * 1e3 / iters, habort, herr);
  }
  return 0;
}
