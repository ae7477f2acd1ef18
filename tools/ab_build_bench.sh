# A/B builds on the GPU box: bash tools/ab_build_bench.sh "<workloads>" "<flags A>" "<flags B>" ...   ("" = default build)
# Prints ms/frame (wall) and kernel ms/frame for every (flags, workload); restores the default build at the end.
cd $GRAFT_REPO_ROOT
WL=$1; shift
for f in "$@"; do
  RTC_CXXFLAGS="$f" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  for w in $WL; do
    timeout -k 10 120 python bench.py --workload $w --steps 64 --warmup 8 --lean 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', '$w', 'ms/frame', d['ms_per_step'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'Mrays/s', d['value'])" || echo "[$f] $w failed"
  done
done
python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
