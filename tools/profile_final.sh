# Runs on the GPU box (gpurun): kernel-trace stats pass + PMC passes of one bench.py workload's ROOFLINE LEG (`--profile-leg`: solo
# launches only, one camera each, a synchronize after each - the launches `roofline.kernel_ms_avg` of the bench line averages).
# Usage: HEAD_SHA=<sha> bash tools/profile_final.sh <tag> [bench.py args, e.g. --workload c4]    -> gpurun_out/prof_<tag>/...
# Each rocprofv3 call has the program itself after `--` (python3 bench.py ...), --pmc passes carry --kernel-trace only.
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
shift || true
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
STEPS_STATS=${STEPS_STATS:-64}
STEPS_PMC=${STEPS_PMC:-12}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --profile-leg --steps $STEPS_STATS $EXTRA > $OUT/stats_bench.log 2>&1
echo "stats pass done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --profile-leg --steps $STEPS_PMC $EXTRA > $OUT/pmc${i}_bench.log 2>&1
  echo "pmc pass $i ($grp) done"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
python3 -c "
import json,sys; d=json.load(open('$OUT/summary.json')); k=d['kernel_trace']; print(k['name'][:60], 'calls', k['calls'], 'avg_us', round(k['avg_ns']/1e3,1)); print({x: d.get(x) for x in ('workload_key','frames_per_launch','per_frame','valu_issue_frac','head')}); print(d.get('hbm_traffic'))"
