// placeholder until the MFMA tile kernel lands (next commit)
#include "glx_internal.h"
static double g_knn_stats[8];
extern "C" int glx_knn_bruteforce(const double* X, int64_t n, int d, int k, int similarity, int64_t* ind_out, double* dist_out, int device) {
  glx_set_error("glx_knn_bruteforce: not built yet");
  return GLX_EUNSUPPORTED;
}
extern "C" int glx_knn_stats(double stats[8]) {
  for (int i = 0; i < 8; ++i) stats[i] = g_knn_stats[i];
  return GLX_OK;
}
