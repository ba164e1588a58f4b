#!/bin/bash
cd /root/repo
O=gpurun_out/r03i
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -8
for l in "" "20 80" "24 72" "28 112" "24 128"; do
  if [ -z "$l" ]; then unset GLX_SELL_L1 GLX_SELL_L4; else set -- $l; export GLX_SELL_L1=$1 GLX_SELL_L4=$2; fi
  timeout 300 python scripts/persist_probe.py --reps 40 2>&1 | grep "float64\|float32"
done | tee $O/l1_fine.log
unset GLX_SELL_L1 GLX_SELL_L4
timeout 600 python bench.py > $O/bench_single.json 2> $O/bench_single.err; head -c 3000 $O/bench_single.json; echo
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_i -o run -- python /root/repo/bench.py --no-traffic --no-scale > $O/prof_bench.log 2>&1)
f=$(find /tmp/prof_i -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-200
timeout 300 python scripts/dist_probe.py > $O/dist_probe.log 2>&1; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/dist_probe.log | grep "graph=1" | cut -c1-200
timeout 900 python scripts/scale_model.py --n4 2e6 --out $O/scale_model.json > $O/scale_model.log 2>&1; grep scale_model $O/scale_model.log | cut -c1-230 | tail -16
timeout 300 python scripts/configs_report.py > $O/configs_report.log 2>&1; tail -25 $O/configs_report.log
