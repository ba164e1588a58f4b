#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
echo "== persist_probe under kernel-trace --stats"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o run -- python /root/repo/scripts/persist_probe.py --reps 5 > $O/p1.log 2>&1; echo "rc $?"; grep -c "@" $O/p1.log; tail -3 $O/p1.log | cut -c1-200
echo "== bench steps 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o run -- python /root/repo/bench.py --no-traffic --no-scale --steps 5 --warmup 1 > $O/p2.log 2>&1; echo "rc $?"; grep -c "@" $O/p2.log; tail -2 $O/p2.log | cut -c1-300
echo "== bench default steps, kernel-trace only (no --stats)"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o run -- python /root/repo/bench.py --no-traffic --no-scale > $O/p3.log 2>&1; echo "rc $?"; grep -c "@" $O/p3.log; tail -2 $O/p3.log | cut -c1-300
ls /tmp/p3/* | head; f=$(find /tmp/p3 -name "*kernel_trace.csv" | head -1); python3 - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f: print('no trace file'); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
for r in csv.DictReader(open(f)):
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg[r['Kernel_Name']]; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (k, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
PY
