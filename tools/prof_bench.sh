# rocprofv3 kernel stats of the default bench workload; summary copied to gpurun_out/prof_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_bench.log 2>&1 || exit 1
cp $R/gpurun_out/prof/bench_kernel_stats.csv $R/gpurun_out/prof_stats.csv
rm -f $R/gpurun_out/prof/bench_kernel_trace.csv
tail -1 $R/gpurun_out/prof_bench.log
