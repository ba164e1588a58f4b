# Developer job (round 6): the host-array paths -- transfer modes, then an API / kernel / copy trace of knn + first fit on a fresh graph
mkdir -p gpurun_out/r06_fresh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python scripts/fresh_path_modes.py > gpurun_out/r06_fresh/modes.txt 2>&1
tail -6 gpurun_out/r06_fresh/modes.txt
rm -rf gpurun_out/r06_fresh/prof
timeout 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/r06_fresh/prof -o fresh -- python scripts/fresh_fit_loop.py > gpurun_out/r06_fresh/prof.log 2>&1
find gpurun_out/r06_fresh/prof -name "*kernel_trace.csv" -delete
head -16 gpurun_out/r06_fresh/prof/fresh_hip_api_stats.csv | cut -c1-110
