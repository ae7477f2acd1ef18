set -e
cd $GRAFT_REPO_ROOT
for w in 2 3 4 5 6 8; do
  RTC_CXXFLAGS="-DRTC_WAVES_PER_SIMD=$w" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  RTC_SRC=3 timeout -k 10 100 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves',$w,'cull', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
  RTC_SRC=0 timeout -k 10 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves',$w,'brute', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done
