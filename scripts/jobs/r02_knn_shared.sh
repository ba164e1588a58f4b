#!/bin/bash
# (record of a job of round 2: some of the compile-time switches it sets were experiments and no longer exist in knn.hip -- the
#  results are in profiles/r02_knn_ablation.txt; KNN_ABLATE, KNN_COUNT and KNN_BF16_NSUB are the ones that remain)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
B="-mllvm -amdgpu-mfma-vgpr-form -DKNN_PADDED_STAGE=1"
export GLX_CXXFLAGS="$B -DKNN_SHARED=0"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
timeout 120 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_shared.txt
export GLX_CXXFLAGS="$B -DKNN_SHARED=1"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
for ns in "" 1 2 4; do
  echo "== shared lists, GLX_KNN_NSPLIT=$ns" | tee -a gpurun_out/r02x/knn_shared.txt
  if [ -n "$ns" ]; then export GLX_KNN_NSPLIT=$ns; else unset GLX_KNN_NSPLIT; fi
  timeout 120 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_shared.txt
done
