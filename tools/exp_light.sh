cd $GRAFT_REPO_ROOT
run() { label=$1; shift; envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout -k 10 150 python bench.py --steps 96 --warmup 16 --lean "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$envs', ' '.join('$*'.split()), '| ms/frame', d['ms_per_step'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'])" || echo "$label failed"; }
timeout -k 10 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -k 10 200 python tests/stress_parity.py 6000 1300000 2>&1 | tail -2
for r in 1 2; do
for w in c3 c5; do
  run lists -- --workload $w
  run walk RTC_LIGHT_LISTS=0 -- --workload $w
done
done
run lists -- --spheres 1000 --reflective
run walk RTC_LIGHT_LISTS=0 -- --spheres 1000 --reflective
run lists -- --workload c3 --views-per-launch 1
run walk RTC_LIGHT_LISTS=0 -- --workload c3 --views-per-launch 1
RTC_CXXFLAGS=-DRTC_STAMPS python raytracer-challenge_amd/build.py --force > /dev/null 2>&1 && python tools/phase_shares.py 10000 flat 2>&1 | grep -v amdgpu
python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
