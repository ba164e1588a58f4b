#!/bin/bash
# thresholds shared between the ref ranges of a query through global memory (KNN_GTAU=1) against none (=0), one box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for v in "-DKNN_GTAU=0" "-DKNN_GTAU=1" "-DKNN_GTAU=1 -DKNN_COUNT=1"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 150 python scripts/knn_variant_probe.py big 2>&1 | awk '/^knn counters/ && (++k%4==1); !/^knn/' | tee -a gpurun_out/r02x/knn_gtau.txt
done
