#!/bin/bash
O=/root/repo/gpurun_out/r03bj
mkdir -p $O
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o run -- python /root/repo/bench.py --no-traffic --no-scale --steps 25 --warmup 2 --min-timed-s 0 > $O/prof_bench.log 2>&1; echo "rc $?"
f=$(find /tmp/prof_r -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; head -8 $O/kernel_stats.csv | cut -c1-200
cd /root/repo
timeout 300 python scripts/configs_report.py 2>&1 | tail -9 | tee $O/configs_report.log
timeout 900 python bench.py > $O/bench_single.json 2> $O/bench_single.err; head -c 600 $O/bench_single.json; echo
