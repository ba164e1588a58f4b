#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe scripts/probes/gather_probe.hip && timeout 300 /tmp/gather_probe > $O/gather_probe2.txt 2>&1; grep -E "128B|stream" $O/gather_probe2.txt
timeout 600 python scripts/configs_report.py > $O/configs.log 2>&1; cat $O/configs.log
timeout 600 python scripts/trials_probe.py > $O/trials.log 2>&1; grep -E "^==|cumulative|tottime" -A12 $O/trials.log | head -60 | cut -c1-160
