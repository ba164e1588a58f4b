#!/bin/bash
# (record of a job of round 2: KNN_XCD_MAP was an experiment and no longer exists in knn.hip; results in profiles/r02_knn_ablation.txt)
# ref range as the fast index of the workgroup id (KNN_XCD_MAP=1: one range per XCD) against (query block, range) = (x, y), one box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for rep in 1 2; do
for v in "-DKNN_XCD_MAP=0" "-DKNN_XCD_MAP=1"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 150 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_xcd.txt
done
done
