# all bench workloads + the one-GPU rehearsal of the RCCL path; logs under gpurun_out/
R=${GRAFT_REPO_ROOT:-.}
cd $R
for w in c3 c4 c2; do
  timeout -k 10 240 python bench.py --workload $w --no-cpu-baseline --gemm-shapes > gpurun_out/b_$w.log 2> gpurun_out/b_$w.err || { echo "$w failed"; tail -5 gpurun_out/b_$w.err; exit 1; }
  python - <<P
import json
d = json.loads(open("gpurun_out/b_$w.log").read().strip().splitlines()[-1])
r = d["roofline"]
print("$w", d["value"], "img/s", d["ms_per_step"], "ms", "util", d["mfma_util_step"], "gemm", r["achieved"], "TF/s", r["gemm_ms_per_step"], "ms")
P
done
timeout -k 10 240 python bench.py --force-dist --no-cpu-baseline --no-kernel-timing > gpurun_out/b_fd.log 2> gpurun_out/b_fd.err || { echo "force-dist failed"; tail -5 gpurun_out/b_fd.err; exit 1; }
tail -c 600 gpurun_out/b_fd.log
