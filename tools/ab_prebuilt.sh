# Interleaved A/B of two PREBUILT libraries (_ab/rtc_A.so, _ab/rtc_B.so: e.g. HEAD and the working tree, built in the
# container) on ONE box: bash tools/ab_prebuilt.sh "<workloads>" [reps]
cd $GRAFT_REPO_ROOT
WL=$1; REPS=${2:-3}
LIB=raytracer-challenge_amd/librtc.so
cp $LIB /tmp/rtc_orig.so
for w in $WL; do
  for r in $(seq $REPS); do
    for v in A B; do
      cp _ab/rtc_$v.so $LIB
      timeout -k 10 120 python bench.py --workload $w --steps 96 --warmup 16 --lean 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'ms/frame', d['ms_per_step'])" || echo "$w $v failed"
    done
  done
done
cp /tmp/rtc_orig.so $LIB
