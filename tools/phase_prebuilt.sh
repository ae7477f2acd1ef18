# Phase shares with a PREBUILT -DRTC_STAMPS library (_ab/rtc_stamps.so, built in the container): bash tools/phase_prebuilt.sh "<n> <flat|reflective>" ...
cd $GRAFT_REPO_ROOT
LIB=raytracer-challenge_amd/librtc.so
cp $LIB /tmp/rtc_orig.so
cp _ab/rtc_stamps.so $LIB
for a in "$@"; do
  timeout -k 10 120 python tools/phase_shares.py $a || echo "phase_shares $a failed"
done
cp /tmp/rtc_orig.so $LIB
