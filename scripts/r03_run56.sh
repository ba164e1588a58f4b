#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03az
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in "" "-DKNN_DIRECT=1" "" "-DKNN_DIRECT=1"; do
export GLX_CXXFLAGS="$f"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; }
echo "flags '$f'"; timeout 300 python scripts/knn_probe.py 2>&1 | tail -1
done
export GLX_CXXFLAGS="-DKNN_DIRECT=1"
timeout 600 python -m pytest tests/test_gpu_knn.py -x -q 2>&1 | tail -2
