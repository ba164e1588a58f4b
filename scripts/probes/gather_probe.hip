// Developer probe: what does a random record gather cost on the MI355X, by record size and by
// working-set size (one XCD's L2 / all L2s / Infinity Cache / HBM)?  The sweep kernel of sweep.hip is
// one such gather per stored edge, so these rates are its roofline at sizes where nothing is reused.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe scripts/probes/gather_probe.hip && /tmp/gather_probe
// Every wavefront walks a coalesced stream of random 32-bit record ids; LPR lanes fetch one record of
// LPR*BPL bytes (BPL = 8/16/32 bytes per lane), 4 independent ids per lane group in flight per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BPL> struct Vec;
template <> struct Vec<8> { typedef double type; };
template <> struct Vec<16> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<32> { typedef double type __attribute__((ext_vector_type(4))); };

template <int BPL> __device__ __forceinline__ double vsum(typename Vec<BPL>::type v);
template <> __device__ __forceinline__ double vsum<8>(double v) { return v; }
template <> __device__ __forceinline__ double vsum<16>(Vec<16>::type v) { return v[0] + v[1]; }
template <> __device__ __forceinline__ double vsum<32>(Vec<32>::type v) { return v[0] + v[1] + v[2] + v[3]; }

// LPR lanes per record (power of two), ACT of them active (ACT <= LPR)
template <int BPL, int LPR, int ACT>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ table, int rec_bytes, const int32_t* __restrict__ ids,
                                                     int64_t ids_per_wave, double* __restrict__ out) {
  typedef typename Vec<BPL>::type V;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int32_t* my = ids + wave * ids_per_wave;
  const int sub = lane % LPR;
  const int grp = lane / LPR;           // 64/LPR groups per wave
  constexpr int NG = 64 / LPR;
  double acc = 0;
  // ids_per_wave is a multiple of NG*4
  for (int64_t base = 0; base < ids_per_wave; base += NG * 4) {
    int32_t id[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) id[t] = my[base + t * NG + grp];
    V x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      x[t] = V{};
      if (sub < ACT) x[t] = *(const V*)(table + (size_t)id[t] * rec_bytes + sub * BPL);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc += vsum<BPL>(x[t]);
  }
  if (acc == 123.456) out[wave] = acc;   // never true: keeps the loads alive
}

__global__ void stream_kernel(const double4* __restrict__ a, int64_t n, double* out) {
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456) out[0] = acc;
}

template <int BPL, int LPR, int ACT>
static void run(const char* name, const char* table, int rec_bytes, int64_t nrec, const int32_t* d_ids, int64_t nids, double* d_out) {
  const int64_t nwaves = 256 * 4 * 16;   // 16 blocks of 4 waves per CU
  constexpr int NG = 64 / LPR;
  int64_t ipw = nids / nwaves / (NG * 4) * (NG * 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((gather_kernel<BPL, LPR, ACT>), dim3((unsigned)(nwaves / 4)), dim3(256), 0, 0, table, rec_bytes, d_ids, ipw, d_out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double edges = (double)ipw * nwaves;
  printf("%-28s rec=%3dB useful=%3dB table=%7.1f MB: %7.2f Gedge/s, useful %6.2f TB/s, line(128B-padded) %6.2f TB/s\n", name, rec_bytes,
         BPL * ACT, (double)nrec * rec_bytes / 1e6, edges / ms / 1e6, edges * BPL * ACT / ms / 1e9,
         edges * ((BPL * ACT + 127) / 128 * 128) / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int64_t nids = 96ll << 20;
  std::vector<int32_t> h(nids);
  int32_t* d_ids;
  double* d_out;
  CK(hipMalloc(&d_ids, nids * 4));
  CK(hipMalloc(&d_out, 1 << 20));
  const size_t table_max = 4ull << 30;
  char* table;
  CK(hipMalloc(&table, table_max));
  CK(hipMemset(table, 0, table_max));
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, (const double4*)table, (int64_t)(table_max / 32), d_out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("streaming read of %.1f GB: %.2f TB/s\n", table_max / 1e9, table_max / ms / 1e9);
  }
  const double sizes_mb[] = {2, 9, 16, 96, 1280};
  std::mt19937_64 rng(1);
  for (double mb : sizes_mb) {
    for (int rec : {128, 96, 64, 32}) {
      const int64_t nrec = (int64_t)(mb * 1e6 / rec);
      for (int64_t i = 0; i < nids; ++i) h[i] = (int32_t)(rng() % (uint64_t)nrec);
      CK(hipMemcpy(d_ids, h.data(), nids * 4, hipMemcpyHostToDevice));
      if (rec == 128) {
        run<16, 8, 8>("128B = 8 lanes x 16B", table, rec, nrec, d_ids, nids, d_out);
        run<16, 8, 6>("128B rec, 6 of 8 lanes x 16B", table, rec, nrec, d_ids, nids, d_out);
        run<32, 4, 4>("128B = 4 lanes x 32B", table, rec, nrec, d_ids, nids, d_out);
        run<32, 4, 3>("128B rec, 3 lanes x 32B", table, rec, nrec, d_ids, nids, d_out);
      } else if (rec == 96) {
        run<32, 4, 3>("96B = 3 lanes x 32B", table, rec, nrec, d_ids, nids, d_out);
      } else if (rec == 64) {
        run<16, 4, 4>("64B = 4 lanes x 16B", table, rec, nrec, d_ids, nids, d_out);
        run<32, 2, 2>("64B = 2 lanes x 32B", table, rec, nrec, d_ids, nids, d_out);
      } else {
        run<8, 4, 4>("32B = 4 lanes x 8B", table, rec, nrec, d_ids, nids, d_out);
        run<32, 1, 1>("32B = 1 lane x 32B", table, rec, nrec, d_ids, nids, d_out);
        run<16, 2, 2>("32B = 2 lanes x 16B", table, rec, nrec, d_ids, nids, d_out);
      }
    }
  }
  return 0;
}
