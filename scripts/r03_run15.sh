#!/bin/bash
O=/root/repo/gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for args in "devsync 1000" "req 1000" "req,devsync,warm4096,build4 1000"; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o run -- python /root/repo/scripts/prof_repro.py $args > $O/repro.log 2>&1; rc=$?
  echo "== $args: rc $rc; $(grep -c '@' $O/repro.log) trace lines; last: $(grep -v '^W2026\|^E2026\|@' $O/repro.log | tail -2 | tr '\n' '|' | cut -c1-200)"
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o run -- python -X faulthandler /root/repo/bench.py --no-traffic --no-scale --steps 5 --warmup 1 > $O/fh.log 2>&1; echo "bench rc $?"
grep -v "^W2026\|^E2026\|    @" $O/fh.log | head -40 | cut -c1-200
