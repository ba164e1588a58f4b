#!/bin/bash
O=/root/repo/gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for args in "plain 50" "plain 1000" "plain 51" "nograph 1000" "warm4096,build4 1000"; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o run -- python /root/repo/scripts/prof_repro.py $args > $O/repro.log 2>&1; rc=$?
  echo "== $args: rc $rc; $(grep -c '@' $O/repro.log) trace lines; last: $(grep -v '^W2026\|^E2026\|@' $O/repro.log | tail -2 | tr '\n' '|' | cut -c1-200)"
done
