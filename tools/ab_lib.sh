# A/B of two builds of the library on one box: zero_amd/libzero_hip_base.so (copy of the previous build, not
# tracked) against zero_amd/libzero_hip.so, alternating.   bash tools/ab_lib.sh '<python command>'
R=${GRAFT_REPO_ROOT:-.}
cd $R
for rep in 1 2; do
  echo "== base (run $rep)"; ZERO_AMD_LIB=$R/zero_amd/libzero_hip_base.so timeout -k 10 280 $1 || exit 1
  echo "== new  (run $rep)"; timeout -k 10 280 $1 || exit 1
done
