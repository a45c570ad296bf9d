# A/B of one environment switch on one box, alternating:   bash tools/ab_env.sh NAME=off_value NAME=on_value '<python command>'
R=${GRAFT_REPO_ROOT:-.}
cd $R
for rep in 1 2; do
  echo "== $1 (run $rep)"; env $1 timeout -k 10 280 $3 || exit 1
  echo "== $2 (run $rep)"; env $2 timeout -k 10 280 $3 || exit 1
done
