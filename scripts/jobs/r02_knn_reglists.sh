#!/bin/bash
# 8-entry candidate lists in registers (KNN_REGLISTS=1: a fourth workgroup per CU) against lists in LDS (=0), one box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for v in "-DKNN_REGLISTS=0" "-DKNN_REGLISTS=1"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 120 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_reglists.txt
done
