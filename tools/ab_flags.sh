# A/B arbitrary RTC_CXXFLAGS sets: bash tools/ab_flags.sh "<bench args>" "<flags A>" "<flags B>" ...
set -e
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for f in "$@"; do
  RTC_CXXFLAGS="$f" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  timeout -k 10 100 python bench.py --steps 100 --warmup 10 --lean $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', '$ARGS', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'])"
done
