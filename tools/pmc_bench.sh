# HBM traffic of every kernel of the default bench workload: two rocprofv3 --pmc passes (the TCC block
# holds FETCH_SIZE or WRITE_SIZE, not both), each a run of its own; summary -> gpurun_out/pmc_summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_fetch.log 2>&1 || { tail -5 $R/gpurun_out/pmc_fetch.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_write.log 2>&1 || { tail -5 $R/gpurun_out/pmc_write.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_summary.json
# the raw per-dispatch tables are large: keep only the summary
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
