cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
run() { label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout -k 10 150 python bench.py --steps 64 --warmup 8 --lean "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$envs', ' '.join('$*'.split()), '| ms/frame', d['ms_per_step'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'])" || echo "$label failed"; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python tests/stress_parity.py 6000 1100000 2>&1 | tail -2
for w in c3 c5; do
  run binned -- --workload $w
  run walk RTC_BINNING=0 -- --workload $w
  run binned -- --workload $w --views-per-launch 1
  run walk RTC_BINNING=0 -- --workload $w --views-per-launch 1
done
export TMPDIR=/tmp
for w in c3 c5; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2r/prof_$w -- python3 bench.py --workload $w --steps 32 --warmup 8 --lean > gpurun_out/r2r/prof_$w.log 2>&1
cat gpurun_out/r2r/prof_$w/*/*kernel_stats.csv | cut -c1-60,200-400 | head -6
done
RTC_CXXFLAGS=-DRTC_STAMPS python raytracer-challenge_amd/build.py --force > /dev/null 2>&1 && python tools/phase_shares.py 10000 flat 2>&1 | grep -v amdgpu
python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
