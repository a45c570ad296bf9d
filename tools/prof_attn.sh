cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_attn
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_attn -o attn --output-format csv -- python3 $R/tools/bench_attn.py 10 > $R/gpurun_out/prof_attn.log 2>&1 || exit 1
rm -f $R/gpurun_out/prof_attn/attn_kernel_trace.csv
