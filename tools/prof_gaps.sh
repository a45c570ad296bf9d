# kernel trace of a short bench run -> idle-gap analysis (tools/gap_analysis.py) in gpurun_out/gaps.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_gaps
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gaps -o bench --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_gaps.log 2>&1 || exit 1
python3 $R/tools/gap_analysis.py $R/gpurun_out/prof_gaps/bench_kernel_trace.csv 0.55 > $R/gpurun_out/gaps.txt
rm -rf $R/gpurun_out/prof_gaps
cat $R/gpurun_out/gaps.txt
