#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03av
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_knn.py -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
for c in 1 0; do
echo "GLX_KNN_ORDER=$c"
GLX_KNN_ORDER=$c GLX_TIMING=1 timeout 600 python - <<'PY' 2>&1 | grep -v "^RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | grep "locality order\|graph_build\|first fit" | tail -6
import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
t0 = time.perf_counter(); s = bench.scale_shard_line(); dt = time.perf_counter() - t0
print({k: s[k] for k in ('graph_build_s', 'knn_search')}, 'f64 us', s['f64']['avg_launch_us'], 'f32 us', s['f32']['avg_launch_us'], 'whole line %.1f s' % dt)
PY
done
