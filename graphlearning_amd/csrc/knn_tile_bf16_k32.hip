// split-bf16 candidate filter, lists of 32 entries (knn_tile_bf16.h)
#include "knn_tile_bf16.h"

int knn_launch_tile_bf16_k32(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed) {
  return launch_tile_bf16_kp<32>(NKB, b, n, q0, q1, nsplit, st, cat, seed);
}

int knn_launch_tile_bf16_k8(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed);
int knn_launch_tile_bf16_k16(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed);

// the dispatch over the list lengths
int knn_launch_tile_bf16(int KP, int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed) {
  if (KP == 8) return knn_launch_tile_bf16_k8(NKB, b, n, q0, q1, nsplit, st, cat, seed);
  if (KP == 16) return knn_launch_tile_bf16_k16(NKB, b, n, q0, q1, nsplit, st, cat, seed);
  return knn_launch_tile_bf16_k32(NKB, b, n, q0, q1, nsplit, st, cat, seed);
}
