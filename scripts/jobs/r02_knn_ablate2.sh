#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
B="-mllvm -amdgpu-mfma-vgpr-form -DKNN_PADDED_STAGE=1"
for v in "$B -DKNN_ABLATE=1" "$B -DKNN_ABLATE=3"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 200 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_ablate2.txt
done
