# Interleaved A/B of two ENVIRONMENT settings on one build and one box:
#   bash tools/ab_env.sh "<workloads>" "<env A>" "<env B>" [reps] [extra bench args]
cd $GRAFT_REPO_ROOT
WL=$1; EA=$2; EB=$3; REPS=${4:-3}; EXTRA=${5:---steps 96 --warmup 16}
for w in $WL; do
  for r in $(seq $REPS); do
    for v in "$EA" "$EB"; do
      env $v timeout -k 10 120 python bench.py --workload $w $EXTRA --lean 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'ms/frame', d['ms_per_step'])" || echo "$w $v failed"
    done
  done
done
