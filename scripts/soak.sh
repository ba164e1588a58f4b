#!/bin/bash
# A soak of the randomised parity suite with breadcrumbs: scripts/soak.sh TAG SCALE WORKERS [INSTANCES] [extra pytest args]
#   INSTANCES independent pytest-xdist sessions of WORKERS workers each run side by side on the one GPU (more processes sharing the
#   device than one session has: the condition the round-4 flake appeared under, and N soaks for the wall time of one).
#   Session i walks the seeds from GLX_SOAK_BASE + (i-1)*10^5 on (different cases in every session).
#   GLX_SOAK_X= (empty) runs every case instead of stopping at the first failure (-x is the default).
# Output: gpurun_out/TAG/inst<i>/{pytest.log,crumbs/*.log}; the last lines of every session are echoed; exit status = number of
# sessions that did not pass.
tag=$1; scale=$2; workers=$3; inst=${4:-1}; shift 4 2>/dev/null || shift $#
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
pids=()
for i in $(seq 1 $inst); do
  out=$root/gpurun_out/$tag/inst$i
  mkdir -p "$out/crumbs"
  GLX_CRUMBS=$out/crumbs GLX_FUZZ_SCALE=$scale GLX_FUZZ_BASE=$(( ${GLX_SOAK_BASE:-0} + (i-1)*100000 )) timeout ${GLX_SOAK_TIMEOUT:-3000} \
    python -m pytest tests/test_gpu_fuzz.py -q ${GLX_SOAK_X--x} -n $workers --tb=long -rf -p no:cacheprovider "$@" > "$out/pytest.log" 2>&1 &
  pids+=($!)
done
bad=0
for i in $(seq 1 $inst); do
  wait ${pids[$((i-1))]}; rc=$?
  echo "== [$tag/inst$i] exit $rc"
  tail -n 4 "$root/gpurun_out/$tag/inst$i/pytest.log" | cut -c1-300
  [ $rc -ne 0 ] && bad=$((bad+1))
done
# what every process was doing last (a crash or a hang leaves a START without an outcome)
for f in "$root"/gpurun_out/$tag/inst*/crumbs/*.log; do
  last=$(grep -a -E ' (START|PASSED|FAILED|SKIPPED) ' "$f" | tail -n 1)
  case "$last" in *" START "*|*" FAILED "*) echo "   $(basename $(dirname $(dirname $f)))/$(basename $f): $last" | cut -c1-300;; esac
done
# the checked uploads of every process (tests/conftest.py writes them at the end of its session)
grep -a -h ' UPLOADS ' "$root"/gpurun_out/$tag/inst*/crumbs/*.log | python3 -c "
import sys, re
t = [0, 0, 0, 0]
for ln in sys.stdin:
    v = re.findall(r': (\\d+)', ln.split('UPLOADS', 1)[1])
    for q in range(min(4, len(v))): t[q] += int(v[q])
print('uploads checked %d, wrong sums %d, repaired by a repeat %d, given up %d' % tuple(t))
"
exit $bad
