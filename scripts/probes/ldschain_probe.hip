// Developer probe: cost per row of the reducer's add chain fed from LDS, for a few read patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// VAR 0: 16-row groups, b128 reads, double buffered (the kernel's pattern)   VAR 1: same with 32-row groups
// VAR 2: 16-row groups, all 16 reads issued then adds (no double buffer)     VAR 3: adds only (registers), the floor
template <int VAR>
__global__ __launch_bounds__(256) void k(double* out, int TR, int ntiles, int ncols) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) double s_tile[];
  const int LDT = TR + 2;
  for (int i = threadIdx.x; i < ncols * LDT; i += 256) s_tile[i] = 1.0 / (i + 1);
  __syncthreads();
  const int tid = threadIdx.x;
  double tot = 0.0;
  constexpr int GR = VAR == 1 ? 32 : 16;
  for (int t = 0; t < ntiles; ++t) {
    if (tid < ncols) {
      const double* col = s_tile + tid * LDT;
      const int ngr = TR / GR;
      double va[GR], vb[GR];
      auto fetch = [&](double (&v)[GR], int g) {
        const double2* src = (const double2*)(col + g * GR);
#pragma unroll
        for (int q = 0; q < GR / 2; ++q) { const double2 x = src[q]; v[2 * q] = x.x; v[2 * q + 1] = x.y; }
      };
      if (VAR == 3) {
        fetch(va, 0);
        for (int g = 0; g < ngr; ++g) {
#pragma unroll
          for (int q = 0; q < GR; ++q) tot = tot + va[q];
          asm volatile("" : "+v"(tot));
        }
      } else if (VAR == 2) {
        for (int g = 0; g < ngr; ++g) {
          fetch(va, g);
#pragma unroll
          for (int q = 0; q < GR; ++q) tot = tot + va[q];
        }
      } else {
        fetch(va, 0);
        int g = 0;
        while (g < ngr) {
          if (g + 1 < ngr) fetch(vb, g + 1);
#pragma unroll
          for (int q = 0; q < GR; ++q) tot = tot + va[q];
          if (++g >= ngr) break;
          if (g + 1 < ngr) fetch(va, g + 1);
#pragma unroll
          for (int q = 0; q < GR; ++q) tot = tot + vb[q];
          ++g;
        }
      }
    }
    __syncthreads();
  }
  if (tid < ncols) out[tid] = tot;
}

int main() {
  double* out; CK(hipMalloc(&out, 64 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int TR = 384, ntiles = 2000;
  for (int ncols : {12, 16, 64})
    for (int var = 0; var < 4; ++var) {
      const size_t shm = (size_t)ncols * (TR + 2) * 8;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (var == 0) { CK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), shm, 0, out, TR, ntiles, ncols); }
        if (var == 1) { CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), shm, 0, out, TR, ntiles, ncols); }
        if (var == 2) { CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), shm, 0, out, TR, ntiles, ncols); }
        if (var == 3) { CK(hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), shm, 0, out, TR, ntiles, ncols); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("ncols %2d var %d: %.3f ns per row\n", ncols, var, ms * 1e6 / ((double)TR * ntiles));
    }
  return 0;
}
