#!/bin/bash
# Phase replays of the headline sweep kernel: the shipped kernel and four cut-down builds of it on the same plan,
# each timed by bench.py's own HIP events (avg_launch_us) -> gpurun_out/TAG/phases.txt.   gpurun -- 'bash scripts/sweep_phases.sh TAG'
tag=${1:-sweep_phases}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cp graphlearning_amd/csrc/sweep.hip /tmp/sweep.hip.shipped
patch -p1 -s < scripts/probes/sweep_replay_experiment.patch || exit 1
: > "$out/phases.txt"
for v in 0 1 2 3 4 0; do
  GLX_CXXFLAGS=-DGLX_SWEEP_REPLAY=$v python -m graphlearning_amd._build > "$out/build$v.log" 2>&1 || { echo "build $v failed"; tail -5 "$out/build$v.log"; }
  GLX_CXXFLAGS=-DGLX_SWEEP_REPLAY=$v python bench.py --steps 20 --warmup 5 --no-configs --no-scale --no-traffic > "$out/bench$v.json" 2> "$out/bench$v.err"
  python - "$out/bench$v.json" $v >> "$out/phases.txt" <<'PY'
import json, sys
names = {0: 'full kernel (shipped)', 1: 'gathers only (no bias load / store / stop reduction)', 2: 'stores only (header + epilogue, no chunk loop)',
         3: 'header only', 4: 'empty kernel'}
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('GLX_SWEEP_REPLAY=%s  %-58s avg launch %.2f us (fp64), %.2f us per sweep in the captured graph; fp32 %.0f sweeps/s' % (
        sys.argv[2], names[int(sys.argv[2])], j['roofline']['avg_launch_us'], 1e3 * j['ms_per_step'] / j['config']['sweeps_per_step'], j['fp32']['value']))
except Exception as e:
    print('GLX_SWEEP_REPLAY=%s  failed: %r' % (sys.argv[2], e))
PY
done
cp /tmp/sweep.hip.shipped graphlearning_amd/csrc/sweep.hip
python -m graphlearning_amd._build > "$out/build_final.log" 2>&1
cat "$out/phases.txt"
