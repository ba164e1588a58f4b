#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r03dz
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
timeout 300 python scripts/knn_host_breakdown.py 2>&1 | tail -6 | tee $O/knn_host.log
(cd /tmp && for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do d=/tmp/pmc_knn; rm -rf $d; timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o run -- python /root/repo/scripts/knn_probe.py > /dev/null 2>&1; ff=$(find $d -name "*counter_collection.csv" | head -1); python3 - "$ff" <<'PY'
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
