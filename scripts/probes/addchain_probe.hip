// Developer probe: latency of a dependent v_add_f64 chain (the CG reference-order reducer's floor).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void chain(const double* __restrict__ a, double* out, long long* cycles, int iters, int lanes) {
#pragma clang fp contract(off)
  if ((int)threadIdx.x >= lanes) return;
  double v[32];
  for (int q = 0; q < 32; ++q) v[q] = a[threadIdx.x * 32 + q];
  double tot = 0.0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      if (MODE == 0) tot = tot + v[q];
      else if (MODE == 1) tot = __builtin_fma(v[q], 1.0, tot);
      else { float f = (float)tot; f = f + (float)v[q]; tot = f; }
    }
    asm volatile("" : "+v"(tot));
  }
  const long long t1 = clock64();
  out[threadIdx.x] = tot;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  double *a, *out; long long* cyc;
  CK(hipMalloc(&a, 64 * 32 * 8)); CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  double h[64 * 32]; for (int i = 0; i < 64 * 32; ++i) h[i] = 1.0 / (i + 1);
  CK(hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode)
    for (int lanes : {64, 16, 1}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, a, out, cyc, iters, lanes);
        else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, a, out, cyc, iters, lanes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("mode %s lanes %2d: %.3f ns per dependent op (%.2f clock64 ticks)\n", mode == 0 ? "add" : "fma", lanes,
             ms * 1e6 / ((double)iters * 32), (double)c / ((double)iters * 32));
    }
  return 0;
}
