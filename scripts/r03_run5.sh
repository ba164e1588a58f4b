#!/bin/bash
cd /root/repo
O=gpurun_out/r03e
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/rcm_probe.py 2>&1 | grep "rcm" | tee $O/rcm_probe.log
for b in 0 1; do GLX_XCD_BALANCE=$b timeout 300 python scripts/persist_probe.py --big 1000000 --cache /tmp/knn_1e6.npz --reps 40 2>&1 | grep GLX_PERSIST | sed "s/^/balance=$b /"; done | tee $O/balance_probe.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -12
timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/config4_1e7.json 2> $O/config4_1e7.err; grep "config 4" $O/config4_1e7.err | tail -20; head -c 1500 $O/config4_1e7.json; echo
