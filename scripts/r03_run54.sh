#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ax
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -5
timeout 300 python scripts/configs_report.py 2>&1 | grep "config 2" | tail -5
for c in 1 0; do echo "GLX_KNN_ORDER=$c"; GLX_KNN_ORDER=$c timeout 600 python bench.py --no-scale > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json
d=json.load(open('$O/bench_$c.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], 'fp32', d['fp32']['value'], d['graph_build']['knn_plus_weights_s'], d['graph_build']['all_calls_s'])"; done
