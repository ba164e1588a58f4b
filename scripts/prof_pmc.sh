#!/bin/bash
# usage: scripts/prof_pmc.sh <tag>  -> gpurun_out/pmc_<tag>/  (per-kernel PMC counters, separate passes)
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$name -o run -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-scale --steps 4 --warmup 1 --min-timed-s 0 "$@" > $out/$name.log 2>&1
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  echo "== $grp  ($f)"
  python3 - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f:
    print('no counter file'); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for r in csv.DictReader(fh):
        k = (r['Kernel_Name'][:60], r['Counter_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
for (kn, cn), (cnt, tot) in sorted(agg.items()):
    if 'spmm' in kn or 'knn_tile' in kn:
        print('%-62s %-24s dispatches=%d mean=%.1f' % (kn, cn, cnt, tot / cnt))
PY
done
