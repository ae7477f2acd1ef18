set -e
cd $GRAFT_REPO_ROOT
for w in ${WAVES:-2 3 4}; do
  RTC_CXXFLAGS="-DRTC_WAVES_PER_SIMD_STACK=$w" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  timeout -k 10 100 python bench.py --steps 30 --warmup 3 --lean --reflective --width 2048 --height 2048 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stack waves',$w, d['value'], 'Mrays/s ms/step', d['ms_per_step'])"
done
