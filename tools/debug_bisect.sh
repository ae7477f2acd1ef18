cd $GRAFT_REPO_ROOT
python tools/debug_seed.py $1 2>&1 | grep "^src"
for f in "-DRTC_NO_LANE_FILTER" "-DRTC_NO_SECONDARY_CULL" "-DRTC_NO_SHADOW_CULL" "-DRTC_NO_PRIMARY_CULL"; do
  RTC_CXXFLAGS="$f" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
  echo "== $f"; python tools/debug_seed.py $1 2>&1 | grep "^src None"
done
