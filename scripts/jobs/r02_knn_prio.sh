#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for v in "-DKNN_PRIO=0" "-DKNN_PRIO=1" "-DKNN_PRIO=2"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 120 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_prio.txt
done
