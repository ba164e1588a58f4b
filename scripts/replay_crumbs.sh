#!/bin/bash
# Replay of a soak worker's test sequence (from its crumbs file): scripts/replay_crumbs.sh SEQUENCE_FILE FIRST_N COPIES REPEATS [GLX_FUZZ_BASE]
#   (GLX_TEST_ABLATE / HIP_LAUNCH_BLOCKING are passed through; failures and the library's DEBUG lines are echoed)
#   COPIES processes at once (1 = the history alone, 12 = the history under the contention of the soak), each REPEATS times.
seq=$1; first=$2; copies=${3:-1}; reps=${4:-1}; base=${5:-100000}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp GLX_FUZZ_SCALE=400 GLX_FUZZ_BASE=$base
ids=$(head -n $first "$seq" | tr '\n' ' ')
for r in $(seq $reps); do
  pids=()
  for c in $(seq $copies); do
    python -m pytest $ids -q -s -p no:cacheprovider -x > /tmp/replay_${r}_${c}.log 2>&1 &
    pids+=($!)
  done
  for c in $(seq $copies); do
    wait ${pids[$((c-1))]}; rc=$?
    [ $rc -ne 0 ] && echo "repeat $r copy $c: exit $rc -- $(tail -n 1 /tmp/replay_${r}_${c}.log | cut -c1-200)"
    grep -a "knn DEBUG\|upload check" /tmp/replay_${r}_${c}.log | head -12 | cut -c1-700
  done
done
echo "replay done: $copies copies x $reps repeats of $first cases (GLX_TEST_ABLATE=$GLX_TEST_ABLATE HIP_LAUNCH_BLOCKING=$HIP_LAUNCH_BLOCKING)"
