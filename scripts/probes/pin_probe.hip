// How fast can 19 MB of fresh host memory be made GPU-writable?  hipHostMalloc by flags, and hipHostRegister on fresh anonymous memory
// with and without transparent huge pages.  Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/pin_probe.hip -o scripts/probes/pin_probe.bin
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  void* w; hipHostMalloc(&w, 1 << 20, 0); hipHostFree(w);
  const size_t bytes = (size_t)19 << 20;
  for (int rep = 0; rep < 2; ++rep) {
    const unsigned flags[] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocCoherent, hipHostMallocPortable | hipHostMallocMapped, hipHostMallocNumaUser};
    const char* names[] = {"default", "noncoherent", "coherent", "portable|mapped", "numa_user"};
    for (int f = 0; f < 5; ++f) {
      double t0 = now(); void* h = nullptr; hipError_t e = hipHostMalloc(&h, bytes, flags[f]); double t1 = now();
      void* dv = nullptr; hipError_t e2 = e == hipSuccess ? hipHostGetDevicePointer(&dv, h, 0) : e;
      printf("hipHostMalloc(%s): %.2f ms (%s, device view %s)\n", names[f], t1 - t0, hipGetErrorString(e), e2 == hipSuccess ? "ok" : "no");
      double t2 = now(); if (h) hipHostFree(h); printf("   hipHostFree %.2f ms\n", now() - t2);
    }
    for (int huge = 0; huge < 2; ++huge) {
      double t0 = now();
      void* p = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      char* a = (char*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
      if (huge) madvise(a, bytes, MADV_HUGEPAGE);
      double t1 = now();
      memset(a, 0, bytes);
      double t2 = now();
      hipError_t e = hipHostRegister(a, bytes, hipHostRegisterDefault);
      double t3 = now();
      void* dv = nullptr; hipError_t e2 = hipHostGetDevicePointer(&dv, a, 0);
      printf("mmap%s %.2f ms, first touch %.2f ms, hipHostRegister %.2f ms (%s, device view %s)\n", huge ? " + MADV_HUGEPAGE" : "", t1 - t0, t2 - t1, t3 - t2,
             hipGetErrorString(e), e2 == hipSuccess ? "ok" : "no");
      double t4 = now(); hipHostUnregister(a); munmap(p, bytes + (2 << 20)); printf("   unregister + munmap %.2f ms\n", now() - t4);
    }
  }
  return 0;
}
