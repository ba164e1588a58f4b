#!/bin/bash
# (record of a job of round 2: some of the compile-time switches it sets were experiments and no longer exist in knn.hip -- the
#  results are in profiles/r02_knn_ablation.txt; KNN_ABLATE, KNN_COUNT and KNN_BF16_NSUB are the ones that remain)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
B="-mllvm -amdgpu-mfma-vgpr-form -DKNN_PADDED_STAGE=1"
for v in "$B" "$B -DKNN_ABLATE=1" "$B -DKNN_ABLATE=3" "$B -DKNN_ABLATE=7" "$B -DKNN_ABLATE=4" "$B -DKNN_ABLATE=2"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 300 python scripts/knn_variant_probe.py 2>&1 | tee -a gpurun_out/r02x/knn_ablate.txt
done
