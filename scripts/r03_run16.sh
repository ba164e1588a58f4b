#!/bin/bash
O=/root/repo/gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for args in "many 1000" "many,devsync,launches 1000" "many 50"; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o run -- python -X faulthandler /root/repo/scripts/prof_repro.py $args > $O/repro.log 2>&1; rc=$?
  echo "== $args: rc $rc; $(grep -c '    @' $O/repro.log) trace lines; last: $(grep -v '^W2026\|^E2026\|    @' $O/repro.log | tail -4 | tr '\n' '|' | cut -c1-300)"
done
