// fp32-input candidate filter, lists of 32 and 64 entries; the dispatch over the list lengths
#include "knn_tile_f32.h"

int knn_launch_tile_f32_short(int KP, int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st);

int knn_launch_tile_f32(int KP, int DH, int nkb, const KnnBufs& b, int64_t n, int64_t q0, int64_t q1, int nsplit, hipStream_t st) {
  if (KP <= 16) return knn_launch_tile_f32_short(KP, DH, nkb, b, n, q0, q1, nsplit, st);
  return KP == 32 ? launch_tile_f32_kp<32>(DH, nkb, b, n, q0, q1, nsplit, st) : launch_tile_f32_kp<64>(DH, nkb, b, n, q0, q1, nsplit, st);
}
