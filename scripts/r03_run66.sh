#!/bin/bash
# group-record variants of the bf16 tile kernel (config 2 and n = 1e6 / d = 64)
for v in "-DKNN_GSLOTS=4" "-DKNN_REGL2=1 -DKNN_GSLOTS=6" "-DKNN_REGL2=1 -DKNN_GSLOTS=3" "-DKNN_REGL2=1 -DKNN_GREC=0" "-DKNN_GREC=0"; do
  echo "== $v"
  GLX_CXXFLAGS="$v" python -m graphlearning_amd._build > /dev/null 2>&1
  export GLX_CXXFLAGS="$v"
  for i in 1 2; do python scripts/knn_host_breakdown.py 2>&1 | sed -n 2p; done
  python scripts/knn_big_breakdown.py 2>&1 | tail -1
  unset GLX_CXXFLAGS
done
