# Extra PMC passes of one workload's roofline leg: instruction cache, branches, scalar pipeline, f64 instruction classes.
# Usage (GPU box): bash tools/pmc_extra.sh <tag> [bench.py args]   -> gpurun_out/pmcx_<tag>/summary.txt
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-x}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcx_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py --profile-leg --steps 10 "$@" > $OUT/p${i}.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
disp = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            disp[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k in sorted(disp):
    v = disp[k]
    print(f"{k:32s} {sum(v.values()) / len(v):16.0f}   (mean over {len(v)} launches of k_trace, summed over XCDs / SEs)")
PY
