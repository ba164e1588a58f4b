#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ar
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for f in "" "-DKNN_REGL2=1" "-DKNN_REGL2=1 -DKNN_BF16_NSUB=1"; do
export GLX_CXXFLAGS="$f"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; }
echo "flags '$f'"; timeout 300 python scripts/knn_probe.py 2>&1 | tail -1
timeout 300 python scripts/knn_filter_probe.py 2>&1 | grep "bf16" | head -2
done
