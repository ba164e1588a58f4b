#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "1 1" "1 0" "0 1"; do set -- $v; echo "== torch=$1 graph=$2"; REPRO_TORCH=$1 REPRO_GRAPH=$2 timeout 120 python scripts/dist_rccl_repro.py 2>&1 | tail -6 | cut -c1-300; done
echo "== rocgdb"; REPRO_TORCH=1 REPRO_GRAPH=1 timeout 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex run -ex bt --args python scripts/dist_rccl_repro.py 2>&1 | tail -40 | cut -c1-250
