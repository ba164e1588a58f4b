#!/bin/bash
O=/root/repo/gpurun_out/r03ba
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python /root/repo/scripts/mbo_probe.py > $O/mbo.log 2>&1
grep "poisson_mbo" $O/mbo.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-160
