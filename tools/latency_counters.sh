# Runs on the GPU box (gpurun): where do the waves of one bench.py workload WAIT? Counter passes (one rocprofv3 call each,
# --pmc with --kernel-trace only, program directly after `--`) for the scalar cache, the in-flight levels of scalar / vector /
# LDS instructions (level sum / instruction count = average latency in cycles) and the wait/busy cycles per instruction class.
# Usage: bash tools/latency_counters.sh <tag> [bench.py args]    -> gpurun_out/lat_<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
shift || true
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/lat_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
i=0
for grp in "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 16 --warmup 8 --lean --no-dropin --no-secondary $EXTRA > $OUT/pmc${i}_bench.log 2>&1 \
    && echo "pass $i ($grp) done" || echo "pass $i ($grp) FAILED"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].startswith("void k_trace"):
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} per launch {tot[k] / max(calls[k], 1):16.1f}   ({calls[k]} launches)")
def ratio(a, b, what):
    if tot.get(a) and tot.get(b):
        print(f"{what}: {tot[a] / calls[a] / (tot[b] / calls[b]):.2f}")
ratio("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM", "average scalar-load latency (cycles)")
ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "average vector-load latency (cycles, level / reads)")
ratio("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "average LDS latency (cycles)")
ratio("SQC_DCACHE_HITS", "SQC_DCACHE_REQ", "scalar cache hit rate")
ratio("SQC_ICACHE_HITS", "SQC_ICACHE_REQ", "instruction cache hit rate")
ratio("SQ_IFETCH_LEVEL", "SQ_IFETCH", "average instruction-fetch latency (cycles)")
PY
