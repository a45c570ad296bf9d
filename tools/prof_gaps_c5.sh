# kernel trace of the c5 (graft decode) bench -> idle-gap analysis in gpurun_out/gaps_c5.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_gaps
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gaps -o bench --output-format csv -- python3 $R/bench.py --workload c5 --steps 3 --warmup 2 --new-tokens 16 > $R/gpurun_out/prof_gaps_c5.log 2>&1 || exit 1
python3 $R/tools/gap_analysis.py $R/gpurun_out/prof_gaps/bench_kernel_trace.csv 0.8 > $R/gpurun_out/gaps_c5.txt
rm -rf $R/gpurun_out/prof_gaps
cat $R/gpurun_out/gaps_c5.txt
