#!/bin/bash
# the ablation done right (the first version's kernel had lost its MFMAs to dead-code elimination): no list maintenance, contraction kept alive
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for v in "" "-DKNN_ABLATE=1" "-DKNN_ABLATE=3"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 150 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_ablate3.txt
done
