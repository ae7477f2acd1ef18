# Interleaved A/B of two builds on ONE box (box-to-box differences are +-8 %): bash tools/ab_interleaved.sh "<workloads>" "<flags A>" "<flags B>" [reps]
cd $GRAFT_REPO_ROOT
WL=$1; FA=$2; FB=$3; REPS=${4:-3}
LIB=raytracer-challenge_amd/librtc.so
RTC_CXXFLAGS="$FA" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1 && cp $LIB /tmp/rtc_A.so || echo "build A failed"
RTC_CXXFLAGS="$FB" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1 && cp $LIB /tmp/rtc_B.so || echo "build B failed"
for w in $WL; do
  for r in $(seq $REPS); do
    for v in A B; do
      cp /tmp/rtc_$v.so $LIB
      timeout -k 10 120 python bench.py --workload $w --steps 96 --warmup 16 --lean 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '$v', 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'ms/frame', d['ms_per_step'])" || echo "$w $v failed"
    done
  done
done
python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
