#!/bin/bash
# compile-time variants of the bf16 kNN tile kernel, one box, one job
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
V=(
""
"-mllvm -amdgpu-mfma-vgpr-form"
"-mllvm -amdgpu-mfma-vgpr-form -DKNN_CINIT=1"
"-mllvm -amdgpu-mfma-vgpr-form -DKNN_PADDED_STAGE=1"
"-mllvm -amdgpu-mfma-vgpr-form -DKNN_CINIT=1 -DKNN_PADDED_STAGE=1"
)
for v in "${V[@]}"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 300 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_variants.txt
done
