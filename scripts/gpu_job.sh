#!/bin/bash
# The one GPU job script: gpurun -- 'bash scripts/gpu_job.sh TAG "<command>" ["<command>" ...]'
# Runs every command from the repository root with a per-command timeout (GLX_JOB_TIMEOUT, default 900 s), stdout+stderr of
# command i to gpurun_out/TAG/cmd<i>.log (the tail is echoed).  The numbers quoted in profiles/ name the TAG they came from.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
out=$root/gpurun_out/$tag
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "== [$tag/$i] $cmd"
  timeout ${GLX_JOB_TIMEOUT:-900} bash -c "$cmd" > "$out/cmd$i.log" 2>&1
  echo "   exit $?"
  tail -n ${GLX_JOB_TAIL:-25} "$out/cmd$i.log" | cut -c1-400
done
