#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ao
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_knn.py -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
for c in 1 0; do
echo "GLX_KNN_CLUSTERED=$c (0: all pairs, library's own order)"
if [ $c = 0 ]; then export GLX_KNN_CLUSTERED=0; fi
timeout 600 python - <<'PY' 2>&1 | grep -v "^RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -3
import sys, time, json
sys.path.insert(0, '/root/repo')
import bench
t0 = time.perf_counter(); s = bench.scale_shard_line(); dt = time.perf_counter() - t0
print({k: s[k] for k in ('graph_build_s', 'knn_search')}, 'f64 us', s['f64']['avg_launch_us'], 'f32 us', s['f32']['avg_launch_us'], 'whole line %.1f s' % dt)
PY
done
