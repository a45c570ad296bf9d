# A/B of two library builds on ONE device: alternating runs of tools/bench_attn.py (base = libzero_hip_base.so)
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "--- base";  ZERO_AMD_LIB=$R/zero_amd/libzero_hip_base.so timeout -k 10 100 python $R/tools/bench_attn.py 50 2>&1 | grep -v amdgpu.ids || exit 1
  echo "--- new";   timeout -k 10 100 python $R/tools/bench_attn.py 50 2>&1 | grep -v amdgpu.ids || exit 1
done
