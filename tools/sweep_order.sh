set -e
cd $GRAFT_REPO_ROOT
for o in 0 1 2; do
  RTC_CXXFLAGS="-DRTC_TILE_ORDER=$o" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  for extra in "" "--reflective" "--spheres 10000 --no-plane"; do
  timeout -k 10 100 python bench.py --steps 200 --warmup 20 --lean $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order',$o,'$extra', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'])"
  done
done
