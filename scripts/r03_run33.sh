#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03ah
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
timeout 300 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print(d['graph_build'])
s = d['scale_shard_1e6']; print({k: s[k] for k in s if k not in ('f64', 'f32')}); print(s['f64'])
"
