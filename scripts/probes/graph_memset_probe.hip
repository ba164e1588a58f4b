// Probe (round 6): does a memset node of a captured launch sequence keep ITS fill value when the sequence is replayed after an eager
// hipMemset with another value?  Run once against the system runtime and once against the one bundled with PyTorch:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/gmp scripts/probes/graph_memset_probe.hip && /tmp/gmp
//   LD_LIBRARY_PATH=$(python -c 'import torch,os;print(os.path.dirname(torch.__file__)+"/lib")') /tmp/gmp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(unsigned long long* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1000] += 1; }
int main() {
  int ver = 0;
  CK(hipRuntimeGetVersion(&ver));
  printf("runtime version %d\n", ver);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long *a = nullptr, *other = nullptr;
  CK(hipMalloc((void**)&a, 1 << 16));
  CK(hipMalloc((void**)&other, 1 << 20));
  CK(hipMemset(a, 0x11, 1 << 16));
  unsigned long long *src_dev = nullptr, *src_pin = nullptr, *other2 = nullptr, *pin2 = nullptr;
  CK(hipMalloc((void**)&src_dev, 4096));
  CK(hipMalloc((void**)&other2, 4096));
  CK(hipHostMalloc((void**)&src_pin, 4096, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin2, 4096, hipHostMallocDefault));
  CK(hipMemset(src_dev, 0x22, 4096));
  CK(hipMemset(other2, 0x55, 4096));
  for (int i = 0; i < 512; ++i) { src_pin[i] = 0x3333333333333333ull; pin2[i] = 0x6666666666666666ull; }
  CK(hipDeviceSynchronize());
  hipGraph_t g;
  hipGraphExec_t ex;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(a + 64, 0, 512, st));               // what the library captured until round 6: one row of stop values back to zero
  CK(hipMemcpyAsync(a + 256, src_dev, 512, hipMemcpyDeviceToDevice, st));      // a device-to-device copy node (dist.hip: the reset of a rank's rows)
  CK(hipMemcpyAsync(a + 512, src_pin, 64, hipMemcpyHostToDevice, st));         // a copy node from page-locked host memory (solver.hip: err0 of min_iter = 0)
  hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, a);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  std::vector<unsigned long long> h(1 << 13);
  for (int round = 0; round < 4; ++round) {
    if (round == 1) { CK(hipMemset(other, 0x7f, 1 << 20)); CK(hipDeviceSynchronize()); }                     // an eager memset, other value, other buffer
    if (round == 2) { CK(hipMemsetD32Async((hipDeviceptr_t)other, 0x7f800000, 1 << 10, st)); CK(hipStreamSynchronize(st)); }
    if (round == 3) { CK(hipMemsetAsync(other, 0xff, 1 << 12, st)); CK(hipStreamSynchronize(st)); }
    std::vector<unsigned long long> pre(1 << 13, 0x1111111111111111ull);
    CK(hipMemcpy(a, pre.data(), 1 << 16, hipMemcpyHostToDevice));      // (not a memset: the last eager memset stays the one above)
    if (round >= 1) {      // eager copies of other buffers between the replays
      CK(hipMemcpyAsync(other, other2, 4096, hipMemcpyDeviceToDevice, st));
      CK(hipMemcpyAsync(other + 1024, pin2, 4096, hipMemcpyHostToDevice, st));
      CK(hipMemcpy(other + 2048, other2, 4096, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(other + 4096, pin2, 4096, hipMemcpyHostToDevice));
      CK(hipStreamSynchronize(st));
      for (int i = 0; i < 8; ++i) src_pin[i] = 0x3333333333333300ull + round;      // the node reads the source as it is at launch time
    }
    CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), a, 1 << 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 64; i < 128; ++i) bad += h[i] != 0ull;
    int bad_dd = 0, bad_hd = 0;
    for (int i = 256; i < 320; ++i) bad_dd += h[i] != 0x2222222222222222ull;
    for (int i = 512; i < 520; ++i) bad_hd += h[i] != (round >= 1 ? 0x3333333333333300ull + round : 0x3333333333333333ull);
    printf("         copy nodes: device-to-device %d of 64 words wrong (first %016llx), from page-locked memory %d of 8 wrong (first %016llx)\n", bad_dd, h[256], bad_hd, h[512]);
    printf("round %d: the 64 words of the node hold %016llx .. %016llx  (%d not zero)   neighbours %016llx %016llx\n", round, h[64], h[127], bad, h[63], h[128]);
  }
  return 0;
}
