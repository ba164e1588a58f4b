#!/bin/bash
# The whole -m gpu suite under pytest-xdist with breadcrumbs: scripts/gpu_suite.sh TAG [WORKERS] [extra pytest args]
# gpurun_out/TAG/{pytest.log,crumbs/}; every failure's traceback is in the log AND in the failing worker's crumb file.
tag=$1; workers=${2:-6}; shift 2 2>/dev/null || shift $#
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
out=$root/gpurun_out/$tag
mkdir -p "$out/crumbs"
GLX_CRUMBS=$out/crumbs timeout ${GLX_SUITE_TIMEOUT:-2400} python -m pytest tests/ -m gpu -q -n $workers --tb=long -rf -p no:cacheprovider "$@" > "$out/pytest.log" 2>&1
rc=$?
echo "== [$tag] pytest -m gpu -n $workers: exit $rc"
grep -a -E "^(FAILED|ERROR)|passed|failed" "$out/pytest.log" | tail -n 30 | cut -c1-300
for f in "$out"/crumbs/*.log; do
  last=$(grep -a -E ' (START|PASSED|FAILED|SKIPPED) ' "$f" | tail -n 1)
  case "$last" in *" START "*) echo "   unfinished in $(basename $f): $last" | cut -c1-300;; esac
done
[ -f "$root/gpurun_out/default_mode_deviations.txt" ] && { echo "== default-mode deviations recorded:"; sort "$root/gpurun_out/default_mode_deviations.txt" | uniq | tail -n 40 | cut -c1-260; }
exit $rc
