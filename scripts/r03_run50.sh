#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03au
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for lo in block rcm; do
for n in 1e6; do
GLX_CONFIG4_LOCAL_ORDER=$lo timeout 900 python bench.py --config 4 --n $n --steps 3 --warmup 1 > $O/c4_${lo}_$n.json 2> $O/c4_${lo}_$n.err
echo "local order $lo n=$n"; grep "config 4" $O/c4_${lo}_$n.err | tail -4; python -c "
import json,sys
d=json.load(open('$O/c4_${lo}_$n.json')); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'build', {k: round(v,2) if isinstance(v,float) else v for k,v in d['build'].items() if k.endswith('_s')})"
done; done
GLX_CONFIG4_LOCAL_ORDER=rcm timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/c4_rcm_1e7.json 2> $O/c4_rcm_1e7.err; grep "config 4" $O/c4_rcm_1e7.err | tail -9; head -c 400 $O/c4_rcm_1e7.json; echo
timeout 1200 python -m pytest tests/test_gpu_dist.py -x -q > $O/pytest_dist.log 2>&1; grep -n "passed\|failed" $O/pytest_dist.log | tail -2
