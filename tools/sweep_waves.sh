set -e
cd $GRAFT_REPO_ROOT
for w in ${WAVES:-2 3 4}; do
  RTC_CXXFLAGS="-DRTC_WAVES_PER_SIMD=$w $EXTRA" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  timeout -k 10 100 python bench.py --steps 200 --warmup 20 --lean 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves',$w,'$EXTRA', d['value'], 'Mrays/s kernel_ms', d['roofline']['kernel_ms_avg'])"
done
