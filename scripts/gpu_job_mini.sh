# Developer job (round 6): the four test files that held the replayed-memset-node failure and the stale-event failure, over and over, six workers
# usage: bash scripts/gpu_job_mini.sh RUNS "ablation list" ["ablation list" ...]      -> gpurun_out/r06_mini/fail_*.log for the runs that failed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
runs=${1:-20}; shift
rm -rf gpurun_out/r06_mini; mkdir -p gpurun_out/r06_mini
for ab in "$@"; do
  fails=0
  for i in $(seq 1 $runs); do
    GLX_TEST_ABLATE=$ab timeout 300 python -m pytest tests/test_gpu_groups.py tests/test_gpu_knn.py tests/test_gpu_trials.py tests/test_gpu_switches.py -m gpu -q -n 6 -p no:cacheprovider --tb=short > /tmp/mini.log 2>&1
    if grep -q "failed\|error" /tmp/mini.log; then fails=$((fails+1)); cp /tmp/mini.log "gpurun_out/r06_mini/fail_${ab:-default}_$i.log"; fi
  done
  echo "== ablations [$ab]: $fails of $runs runs had a failure ($(tail -1 /tmp/mini.log))"
done
