// fp32-input candidate filter, lists of 8 and 16 entries (the short lists of k <= 12 / k <= 28)
#include "knn_tile_f32.h"

int knn_launch_tile_f32_short(int KP, int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st) {
  return KP == 8 ? launch_tile_f32_kp<8>(DH, nkb, b, n, q0, q1, nsplit, st) : launch_tile_f32_kp<16>(DH, nkb, b, n, q0, q1, nsplit, st);
}
