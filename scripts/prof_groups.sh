#!/bin/bash
# usage: scripts/prof_groups.sh <tag> "<grp1 counters>" "<grp2 counters>" ...   (one rocprofv3 --pmc pass per group)
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/grp_$tag
mkdir -p $out
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > $out/g$i.log 2>&1
  f=$(find $out/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f: print('no counter file'); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:44], r['Counter_Name']); agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
for (kn, cn), (cnt, tot) in sorted(agg.items()):
    if 'spmm_sell_kernel<double' in kn: print('%-46s %-36s n=%d mean=%.1f' % (kn, cn, cnt, tot / cnt))
PY
done
