#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03aw
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed\|Error\|error" $O/pytest.log | tail -5
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['graph_build']['knn_plus_weights_s']); s=d['scale_shard_1e6']; print({k: s[k] for k in s if k not in ('f64','f32')}); print(s['f64']); print(s['f32'])"
