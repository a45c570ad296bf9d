# MFMA utilisation of every kernel of the default bench workload from hardware counters: one
# rocprofv3 --pmc pass (7 SQ counters + GRBM_GUI_ACTIVE) over a short bench run, and the same pass over
# tools/mfma_peak.hip (back-to-back MFMAs on every SIMD = the 100 % point that calibrates the ratio
# SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE).  Summary -> gpurun_out/pmc_mfma_summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_peak
hipcc --offload-arch=gfx950 -O3 $R/tools/mfma_peak.hip -o /tmp/mfma_peak > $R/gpurun_out/pmc_mfma.log 2>&1 || { tail -5 $R/gpurun_out/pmc_mfma.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_peak -o pmc --output-format csv -- /tmp/mfma_peak >> $R/gpurun_out/pmc_mfma.log 2>&1 || { tail -5 $R/gpurun_out/pmc_mfma.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc $C -d $R/gpurun_out/pmc_mfma -o pmc --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing >> $R/gpurun_out/pmc_mfma.log 2>&1 || { tail -5 $R/gpurun_out/pmc_mfma.log; exit 1; }
python3 $R/tools/pmc_mfma_summary.py $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_peak $R/gpurun_out/pmc_mfma_summary.json
rm -rf $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_peak
