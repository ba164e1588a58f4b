#!/bin/bash
# (record of a job of round 2: some of the compile-time switches it sets were experiments and no longer exist in knn.hip -- the
#  results are in profiles/r02_knn_ablation.txt; KNN_ABLATE, KNN_COUNT and KNN_BF16_NSUB are the ones that remain)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
for v in "-DKNN_BF_APPEND=0 -DKNN_ANY_LOOP=0" "-DKNN_BF_APPEND=0 -DKNN_ANY_LOOP=1" "-DKNN_BF_APPEND=1 -DKNN_ANY_LOOP=0" "-DKNN_BF_APPEND=1 -DKNN_ANY_LOOP=1"; do
  export GLX_CXXFLAGS="$v"
  timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
  timeout 120 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_append.txt
done
