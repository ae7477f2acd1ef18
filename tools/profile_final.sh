# Runs on the GPU box (gpurun): kernel-trace stats pass + PMC passes of bench.py's default workload.
# Usage: bash tools/profile_final.sh <tag> [extra bench.py args]    -> gpurun_out/prof_<tag>/...
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
shift || true
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 200 --warmup 24 --no-cpu-baseline $EXTRA > $OUT/stats_bench.log 2>&1
echo "stats pass done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline $EXTRA > $OUT/pmc${i}_bench.log 2>&1
  echo "pmc pass $i ($grp) done"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
