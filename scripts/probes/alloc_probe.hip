// What allocations cost on this box (first-use latency): hipMalloc / hipFree, hipHostMalloc / hipHostFree, hipHostRegister of
// malloc'ed memory, and host-to-device / device-to-host copies from pageable against page-locked memory, by size.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/alloc_probe.hip -o scripts/probes/alloc_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  void* warm; hipMalloc(&warm, 1 << 20); hipFree(warm);
  for (size_t mb : {1, 4, 16, 64}) {
    const size_t bytes = mb << 20;
    double t0 = now(); void* d; hipMalloc(&d, bytes); double t1 = now();
    void* h; hipHostMalloc(&h, bytes, hipHostMallocDefault); double t2 = now();
    memset(h, 1, bytes); double t3 = now();
    void* p = malloc(bytes); memset(p, 1, bytes); double t4 = now();
    hipMemcpy(d, p, bytes, hipMemcpyHostToDevice); double t5 = now();
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); double t6 = now();
    hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t7 = now();
    hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double t8 = now();
    hipHostRegister(p, bytes, hipHostRegisterDefault); double t9 = now();
    hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost); double t10 = now();
    hipHostUnregister(p); double t11 = now();
    hipHostFree(h); double t12 = now();
    hipFree(d); double t13 = now();
    void* d2; hipMalloc(&d2, bytes); double t14 = now(); hipFree(d2);
    printf("%3zu MB: hipMalloc %.3f ms (again %.3f), hipHostMalloc %.3f, first touch pinned %.3f, malloc+touch pageable %.3f | H2D pageable %.3f pinned %.3f | "
           "D2H pageable %.3f pinned %.3f | hipHostRegister %.3f, D2H registered %.3f, unregister %.3f | hipHostFree %.3f hipFree %.3f\n",
           mb, t1 - t0, t14 - t13, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8, t10 - t9, t11 - t10, t12 - t11, t13 - t12);
    free(p);
  }
  // many small allocations (the search's work buffers)
  double t0 = now();
  void* ptrs[40];
  for (int i = 0; i < 40; ++i) hipMalloc(&ptrs[i], (size_t)(64 << 10) * (1 + i % 8));
  double t1 = now();
  for (int i = 0; i < 40; ++i) hipFree(ptrs[i]);
  double t2 = now();
  printf("40 small hipMalloc (64-512 KB): %.3f ms, 40 hipFree %.3f ms\n", t1 - t0, t2 - t1);
  return 0;
}
