#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03be
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -5
timeout 300 python scripts/knn_filter_probe.py 2>&1 | grep "bf16" | head -5
timeout 600 python bench.py --no-scale > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['graph_build'])"
