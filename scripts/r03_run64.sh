#!/bin/bash
O=/root/repo/gpurun_out/r03bh
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python /root/repo/bench.py --config 4 --n 1e6 --steps 2 --warmup 1 > $O/log.txt 2> $O/err.txt
grep "config 4" $O/err.txt | tail -8
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
print('%-70s %7s %10s %10s' % ('kernel', 'calls', 'avg us', 'total ms'))
for r in rows[:30]:
    print('%-70s %7s %10.1f %10.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
