# like ab_multi.sh, with the keys to print: bash tools/ab_multi_keys.sh "<names>" "<bench args>" "<keys>" [reps]
cd $GRAFT_REPO_ROOT
NAMES=$1; ARGS=$2; KEYS=$3; REPS=${4:-2}
LIB=raytracer-challenge_amd/librtc.so
cp $LIB /tmp/rtc_orig.so
for r in $(seq $REPS); do for v in $NAMES; do
  cp _ab/rtc_$v.so $LIB
  timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 tools/_line.py "$v" $KEYS || echo "$v failed"
done; done
cp /tmp/rtc_orig.so $LIB
