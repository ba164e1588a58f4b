#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for rep in 1 2; do
for v in "-DKNN_GTAU=0" "-DKNN_GTAU=1"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 150 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_gtau2.txt
done
done
