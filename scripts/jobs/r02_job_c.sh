#!/bin/bash
# round-2 job C: full GPU suite on the new fit path, bench line with measured traffic, nontemporal-stream experiment
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/pytest_all.log; cat $O/pytest_all.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -c 6000 $O/bench.log
for nt in 0 1 2 3; do
  echo "== GLX_NT=$nt"; GLX_NT=$nt timeout 600 python scripts/scale_probe.py 1000000 --cache /tmp/knn6.npz --reps 3 --dtype both 2>&1 | grep sweep
done
echo "== 70k GLX_NT"; for nt in 0 1 2; do GLX_NT=$nt timeout 300 python bench.py --no-traffic --no-scale --steps 50 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_us'], d['ms_per_step'])"; done
timeout 600 python scripts/configs_report.py > $O/configs.log 2>&1; cat $O/configs.log
