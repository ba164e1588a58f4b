#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03y
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export GLX_CXXFLAGS="-DKNN_COUNT=1"
timeout 600 python -c "from graphlearning_amd import _build; _build.build_lib()" > $O/build.log 2>&1 || echo build failed
for s in 0 8; do
GLX_KNN_SEED=$s timeout 300 python - > $O/count_$s.txt 2>&1 <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '/root/repo')
import bench
from graphlearning_amd import _hip
X = bench.make_features(bench.load_labels(70000))
ind, d = _hip.knn_bruteforce(X, 11)
print(_hip.knn_stats())
PY
echo "seed $s"; grep "knn counters\|seed_sample" $O/count_$s.txt
done
