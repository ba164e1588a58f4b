cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_mbo -o run -- python $R/scripts/mbo_probe.py > $R/gpurun_out/prof_r04_mbo.log 2>&1
head -14 $(find $R/gpurun_out/prof_r04_mbo -name "*kernel_stats.csv" | head -1) | cut -c1-180
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_knn -o run -- python $R/scripts/knn_stage_times.py 6 > $R/gpurun_out/prof_r04_knn.log 2>&1
head -30 $(find $R/gpurun_out/prof_r04_knn -name "*kernel_stats.csv" | head -1) | cut -c1-180
tail -4 $R/gpurun_out/prof_r04_knn.log
