// split-bf16 candidate filter, lists of 16 entries (knn_tile_bf16.h)
#include "knn_tile_bf16.h"

int knn_launch_tile_bf16_k16(int NKB, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st, int cat, bool seed) {
  return launch_tile_bf16_kp<16>(NKB, b, n, q0, q1, nsplit, st, cat, seed);
}
