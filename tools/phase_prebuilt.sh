# Phase shares with PREBUILT -DRTC_STAMPS libraries (_ab/rtc_<name>.so, built in the container): bash tools/phase_prebuilt.sh "<names>" "<n> <flat|reflective>" ...
cd $GRAFT_REPO_ROOT
LIB=raytracer-challenge_amd/librtc.so
NAMES=$1; shift
cp $LIB /tmp/rtc_orig.so
for v in $NAMES; do
  cp _ab/rtc_$v.so $LIB
  for a in "$@"; do
    echo "== $v: $a"
    timeout -k 10 120 python tools/phase_shares.py $a 2>&1 | grep -v amdgpu.ids || echo "phase_shares $a failed"
  done
done
cp /tmp/rtc_orig.so $LIB
