# Interleaved comparison of several PREBUILT libraries (_ab/rtc_<name>.so) on ONE box: bash tools/ab_multi.sh "<names>" "<bench args>" [reps]
cd $GRAFT_REPO_ROOT
NAMES=$1; ARGS=$2; REPS=${3:-2}
LIB=raytracer-challenge_amd/librtc.so
cp $LIB /tmp/rtc_orig.so
for r in $(seq $REPS); do
  for v in $NAMES; do
    cp _ab/rtc_$v.so $LIB
    timeout -k 10 200 python bench.py $ARGS --lean 2>/dev/null | python3 tools/_line.py "$v" ms_per_step roofline.kernel_ms_avg || echo "$v failed"
  done
done
cp /tmp/rtc_orig.so $LIB
