#!/bin/bash
# usage: scripts/prof_cg_pmc.sh <tag>  -> gpurun_out/pmc_cg_<tag>/ : instruction and wait counters of the exact-mode CG's reduction kernels
# (block form: ss_sum / ss_quant / ss_walk; chain form: cg_seqsum_dpp), separate rocprofv3 --pmc passes over scripts/cg_forms.py
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_cg_$tag
mkdir -p $out
cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$name -o run -- python $GRAFT_REPO_ROOT/scripts/cg_forms.py 1 auto > $out/$name.log 2>&1
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  echo "== $grp"
  python3 - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
if not f:
    print('no counter file'); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for r in csv.DictReader(fh):
        kn = r['Kernel_Name']
        if not ('ss_' in kn or 'cg_seqsum' in kn):
            continue
        k = (kn.split('(')[0][-40:], r['Counter_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
for (kn, cn), (cnt, tot) in sorted(agg.items()):
    print('%-42s %-20s dispatches=%5d mean=%14.1f' % (kn, cn, cnt, tot / cnt))
PY
done
