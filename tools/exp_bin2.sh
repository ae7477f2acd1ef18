cd $GRAFT_REPO_ROOT
run() { label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout -k 10 150 python bench.py --steps 96 --warmup 16 --lean "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$envs', ' '.join('$*'.split()), '| ms/frame', d['ms_per_step'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'])" || echo "$label failed"; }
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python tests/stress_parity.py 6000 1200000 2>&1 | tail -2
for r in 1 2; do
for w in ns c2 c4; do
  run binned -- --workload $w
  run walk RTC_BINNING=0 -- --workload $w
done
done
run binned-single RTC_BIN_SMALL_VIEWS=1 -- --workload ns --views-per-launch 1
run walk-single RTC_BINNING=0 -- --workload ns --views-per-launch 1
run binned -- --workload c3
run binned -- --workload c5
run binned -- --reflective
run walk RTC_BINNING=0 -- --reflective
