# Pipeline depth x steps sweep of the headline loop on ONE box: bash tools/sweep_pipeline.sh "<workloads>" "<depths>" "<steps list>" [reps]
cd $GRAFT_REPO_ROOT
WL=${1:-ns}; DEPTHS=${2:-"1 2 3 4"}; STEPS=${3:-"20 200"}; REPS=${4:-2}
for w in $WL; do for r in $(seq $REPS); do for p in $DEPTHS; do for st in $STEPS; do
  timeout -k 10 200 python bench.py --workload $w --steps $st --warmup 5 --lean --pipeline $p 2>/dev/null | python3 tools/_line.py "$w p=$p steps=$st" ms_per_step value roofline.kernel_ms_avg roofline.binning_kernel_ms_avg roofline.kernel_ms_avg_while_overlapped || echo "$w p=$p failed"
done; done; done; done
