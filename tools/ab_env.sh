# Interleaved A/B of environment settings on ONE build and box: bash tools/ab_env.sh "<workloads>" "<env A>|<env B>|..." [reps] ["bench args"]
cd $GRAFT_REPO_ROOT
WL=$1; IFS='|' read -ra ENVS <<< "$2"; REPS=${3:-3}; ARGS=${4:-"--steps 200 --warmup 8"}
for w in $WL; do for r in $(seq $REPS); do for e in "${ENVS[@]}"; do
  env $e timeout -k 10 200 python bench.py --workload $w $ARGS --lean 2>/dev/null | python3 tools/_line.py "$w [$e]" ms_per_step roofline.kernel_ms_avg || echo "$w [$e] failed"
done; done; done
