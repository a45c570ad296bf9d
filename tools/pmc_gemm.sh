cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 0 1 2; do
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2_t$t -o pmc --output-format csv -- python3 $R/tools/pmc_gemm.py 6144 16384 3072 $t > $R/gpurun_out/pmc2_t$t.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc3_t$t -o pmc --output-format csv -- python3 $R/tools/pmc_gemm.py 6144 16384 3072 $t > $R/gpurun_out/pmc3_t$t.log 2>&1 || exit 1
done
