#!/bin/bash
O=/root/repo/gpurun_out/r03r
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o run -- python /root/repo/bench.py --no-traffic --no-scale --steps 25 --warmup 2 --min-timed-s 0 > $O/prof_bench.log 2>&1; echo "rc $?"
f=$(find /tmp/prof_r -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-230
grep "^{" $O/prof_bench.log | head -c 1200; echo
cd /root/repo
timeout 900 python scripts/scale_model.py --n4 2e6 --out $O/scale_model.json > $O/scale_model.log 2>&1; grep scale_model $O/scale_model.log | cut -c1-200 | tail -15
timeout 300 python scripts/dist_probe.py > $O/dist_probe.log 2>&1; grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/dist_probe.log | grep "graph=1" | cut -c1-200
timeout 300 python scripts/configs_report.py 2>&1 | tail -9 | tee $O/configs_report.log
timeout 900 python bench.py > $O/bench_single.json 2> $O/bench_single.err; head -c 700 $O/bench_single.json; echo
