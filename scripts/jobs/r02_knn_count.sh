#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02x
export GLX_CXXFLAGS="-DKNN_COUNT=2"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" || echo build failed
timeout 200 python scripts/knn_variant_probe.py big 2>&1 | tee -a gpurun_out/r02x/knn_count.txt
