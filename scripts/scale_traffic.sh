#!/bin/bash
# Memory traffic of the fp64 sweep at a size where HBM is real: scripts/scale_traffic.sh TAG N [T]
#   1. the run itself with the host-side bounds (algorithmic bytes, distinct-line bound), kNN lists cached under /tmp
#   2. rocprofv3 --pmc passes over the same command (FETCH_SIZE | WRITE_SIZE | TCC hit/miss | TCP->TCC requests), per-kernel means
# Output under gpurun_out/TAG/; the summary goes to profiles/ by hand.
tag=$1; n=$2; T=${3:-20}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python scripts/scale_probe.py $n --cache /tmp/glx_knn_$n.npz --T $T --reps 3 --bound 2>&1 | tee gpurun_out/$tag/run.log | tail -12
python scripts/prof_run.py $tag --no-stats --match "spmm_sell_kernel<double, 4, true" --pmc FETCH_SIZE --pmc WRITE_SIZE --pmc "TCC_HIT_sum TCC_MISS_sum" \
  --pmc "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" -- python scripts/scale_probe.py $n --cache /tmp/glx_knn_$n.npz --T $T --reps 2 2>&1 | tail -14
