#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03l
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -8
timeout 600 python bench.py > $O/bench_single.json 2> $O/bench_single.err; head -c 1500 $O/bench_single.json; echo
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o run -- python /root/repo/bench.py --no-traffic --no-scale > $O/prof_bench.log 2>&1)
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; head -8 $O/kernel_stats.csv | cut -c1-220
GLX_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 > $O/bench_dist1.json 2> $O/bench_dist1.err; head -c 400 $O/bench_dist1.json; echo
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -6 | tee $O/knn_host.log
(cd /tmp && for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU"; do d=/tmp/pmc_knn; rm -rf $d; timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o run -- python /root/repo/scripts/knn_probe.py > /dev/null 2>&1; ff=$(find $d -name "*counter_collection.csv" | head -1); python3 - "$ff" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f: print('no counter file'); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if 'knn_tile' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:60], r['Counter_Name'])][0] += 1; agg[(r['Kernel_Name'][:60], r['Counter_Name'])][1] += float(r['Counter_Value'])
for (kn, cn), (cnt, tot) in sorted(agg.items()):
    print('pmc %-62s %-28s mean of %d: %.1f' % (kn, cn, cnt, tot / cnt))
PY
done) 2>&1 | tee $O/knn_pmc.log
