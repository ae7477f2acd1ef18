cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
python bench.py > gpurun_out/r3m/bench_default.json 2> gpurun_out/r3m/bench_default.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin --no-secondary 2>/dev/null > gpurun_out/r3m/bench_driverlike.json
for w in c2 c3 c4 c5; do python bench.py --workload $w --no-cpu-baseline --no-dropin --no-secondary 2>/dev/null > gpurun_out/r3m/bench_$w.json; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3m/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("kernel_ms_per_frame"), d.get("single_view"))
PY
