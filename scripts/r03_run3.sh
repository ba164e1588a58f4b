#!/bin/bash
cd /root/repo
O=gpurun_out/r03c
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in leak exc gc_comm_first; do for t in notorch torch; do
  timeout 90 python scripts/exit_hang_probe.py $m $t > $O/exit_${m}_${t}.log 2>&1; echo "exit probe $m $t: rc $?"
  GLX_NO_ATEXIT_ABANDON=1 timeout 90 python scripts/exit_hang_probe.py $m $t > $O/exit_noabandon_${m}_${t}.log 2>&1; echo "exit probe (no abandon) $m $t: rc $?"
done; done 2>&1 | tee $O/exit_probe.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -15
# ---- vertex-order experiment with counters: n = 1e6, library RCM order vs k-means cells of feature space
export TMPDIR=/tmp
timeout 600 python scripts/order_probe.py 1000000 --cache /tmp/knn_1e6.npz --orders rcm,blob,kmeans 2>&1 | grep "^order" | tee $O/order_1e6.log
for ord in rcm kmeans; do
  for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    d=/tmp/pmc_$ord; rm -rf $d
    (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o run -- python /root/repo/scripts/order_probe.py 1000000 --cache /tmp/knn_1e6.npz --orders $ord --reps 1 --T 10 > /dev/null 2>&1)
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$ord" <<'PY'
import sys, csv, collections
f, ord_ = sys.argv[1], sys.argv[2]
if not f: print('order %s: no counter file' % ord_); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if 'spmm_sell_kernel<double' in r['Kernel_Name']:
        agg[r['Counter_Name']][0] += 1; agg[r['Counter_Name']][1] += float(r['Counter_Value'])
for cn, (cnt, tot) in sorted(agg.items()):
    print('pmc order %-7s %-28s per launch (mean of %d): %.1f' % (ord_, cn, cnt, tot / cnt))
PY
  done
done 2>&1 | tee $O/order_1e6_pmc.log
timeout 1200 python scripts/order_probe.py 10000000 --orders rcm,blob --reps 2 --T 10 2>&1 | grep "^order" | tee $O/order_1e7.log
timeout 900 python scripts/scale_model.py --n4 2e6 --out $O/scale_model.json > $O/scale_model.log 2>&1; grep scale_model $O/scale_model.log | cut -c1-330 | tail -30
