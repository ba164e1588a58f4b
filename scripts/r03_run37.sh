#!/bin/bash
O=/root/repo/gpurun_out/r03al
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python /root/repo/scripts/cg_probe.py > $O/cg.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/cg_kernel_stats.csv
head -22 "$f" | cut -c1-200
