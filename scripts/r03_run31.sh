#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03af
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_knn.py -x -q > $O/pytest_knn.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest_knn.log | tail -8
for f in "" "-DKNN_BF16_NSTG=1" "-DKNN_BF16_NSTG=4" ""; do
export GLX_CXXFLAGS="$f"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" > $O/build.log 2>&1 || { echo build failed; tail -5 $O/build.log; }
echo "flags '$f'"; timeout 300 python scripts/knn_probe.py 2>&1 | tail -1
timeout 300 python scripts/knn_filter_probe.py 2>&1 | grep "bf16" | head -6
done
