# The driver's own invocation (python bench.py --gpus 1 --steps 20 --warmup 5) under different binning settings, interleaved.
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for env in "RTC_BINNING=1" "RTC_BINNING=0" "RTC_BIN_SMALL_PIXELS=999999999999"; do
    env $env python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-secondary 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env', 'ms/frame', d['ms_per_step'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'value', d['value'])"
  done
done
