# rocprofv3 kernel stats of the C5 (70B shard) graft inference bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c5 -o c5 --output-format csv -- python3 $R/bench.py --workload c5 --steps 3 --warmup 2 --new-tokens 16 > $R/gpurun_out/prof_c5.log 2>&1 || exit 1
cp $R/gpurun_out/prof_c5/c5_kernel_stats.csv $R/gpurun_out/prof_c5_stats.csv
rm -f $R/gpurun_out/prof_c5/c5_kernel_trace.csv
tail -1 $R/gpurun_out/prof_c5.log | cut -c1-400
