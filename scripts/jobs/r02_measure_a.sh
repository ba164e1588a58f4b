#!/bin/bash
# round-2 measurement job A: gather micro-benchmark + config-4-shaped sweeps at 1e6 / 1e7 with rocprof evidence
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe scripts/probes/gather_probe.hip && timeout 300 /tmp/gather_probe > $O/gather_probe.txt 2>&1
tail -80 $O/gather_probe.txt
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0_RDREQ[A-Za-z0-9_]*|TCC_EA0_WRREQ[A-Za-z0-9_]*|TCP_[A-Z_0-9a-z]*HIT[A-Za-z_0-9]*|TCC_BUBBLE[a-z_A-Z0-9]*|TCC_REQ[a-zA-Z_0-9]*|TCP_TCC_[A-Za-z_0-9]*" | sort -u | tr '\n' ' ' > $O/counters.txt
cat $O/counters.txt; echo
timeout 600 python scripts/scale_probe.py 1000000 --cache /tmp/knn6.npz --diag --dtype both > $O/scale_1e6.log 2>&1; cat $O/scale_1e6.log
PMC1="FETCH_SIZE"; PMC2="WRITE_SIZE"; PMC3="TCC_HIT_sum TCC_MISS_sum"; PMC4="TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; PMC5="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
timeout 900 python scripts/prof_run.py r02a/scale_1e6 --match spmm_sell --pmc "$PMC1" --pmc "$PMC2" --pmc "$PMC3" --pmc "$PMC4" --pmc "$PMC5" -- python scripts/scale_probe.py 1000000 --cache /tmp/knn6.npz --reps 2
timeout 900 python scripts/scale_probe.py 10000000 --cache /tmp/knn7.npz --diag > $O/scale_1e7.log 2>&1; cat $O/scale_1e7.log
timeout 1500 python scripts/prof_run.py r02a/scale_1e7 --match spmm_sell --pmc "$PMC1" --pmc "$PMC2" --pmc "$PMC3" --pmc "$PMC4" --pmc "$PMC5" -- python scripts/scale_probe.py 10000000 --cache /tmp/knn7.npz --reps 1 --T 20
