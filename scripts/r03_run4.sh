#!/bin/bash
cd /root/repo
O=gpurun_out/r03d
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -25
for m in gc_comm_first exc; do
  timeout 90 python scripts/exit_hang_probe.py $m torch > $O/exit_${m}.log 2>&1; echo "exit probe $m: rc $?"
done 2>&1 | tee $O/exit_probe.log
timeout 300 python scripts/rcm_probe.py 2>&1 | grep "rcm" | tee $O/rcm_probe.log
timeout 120 python scripts/knn_host_breakdown.py 2>&1 | tail -8 | tee $O/knn_host.log
timeout 600 python bench.py > $O/bench_single.json 2> $O/bench_single.err; head -c 2500 $O/bench_single.json; echo
GLX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 > $O/bench_dist1.json 2> $O/bench_dist1.err; head -c 600 $O/bench_dist1.json; echo
timeout 900 python scripts/scale_model.py --n4 2e6 --out $O/scale_model.json > $O/scale_model.log 2>&1; grep scale_model $O/scale_model.log | cut -c1-250 | tail -16
timeout 1500 python bench.py --config 4 --n 1e7 --steps 2 --warmup 1 > $O/config4_1e7.json 2> $O/config4_1e7.err; grep "config 4" $O/config4_1e7.err | tail -20; head -c 3000 $O/config4_1e7.json; echo
