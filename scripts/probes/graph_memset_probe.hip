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
  CK(hipDeviceSynchronize());
  hipGraph_t g;
  hipGraphExec_t ex;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(a + 64, 0, 512, st));               // what the library captures: one row of stop values back to zero
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
    CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), a, 1 << 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 64; i < 128; ++i) bad += h[i] != 0ull;
    printf("round %d: the 64 words of the node hold %016llx .. %016llx  (%d not zero)   neighbours %016llx %016llx\n", round, h[64], h[127], bad, h[63], h[128]);
  }
  return 0;
}
