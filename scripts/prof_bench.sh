#!/bin/bash
# usage: scripts/prof_bench.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/ (kernel stats)
# (short run: rocprofv3 of ROCm 7.2 segfaults after 16 384 dispatches launched from device graphs, profiles/README.md)
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-scale --steps 25 --warmup 2 --min-timed-s 0 "$@" > $out/bench.log 2>&1
find $out -name "*kernel_stats*" | head -3
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -12 "$f"
tail -2 $out/bench.log | cut -c1-600
