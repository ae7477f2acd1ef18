# Diagnosis on the GPU box: what does a frame cost when the World is empty / a plane / small, and what do the canvas stores and
# the ray-counter atomics contribute? Needs prebuilt variants _ab/rtc_V*.so (RTC_CXXFLAGS = "", -DRTC_DIAG_NO_COUNTERS,
# -DRTC_DIAG_NO_STORE, both).   bash tools/floor_cost.sh
cd $GRAFT_REPO_ROOT
LIB=raytracer-challenge_amd/librtc.so
cp $LIB /tmp/rtc_orig.so
for v in V VNO_COUNTERS VNO_STORE VNO_COUNTERSNO_STORE; do
  cp _ab/rtc_$v.so $LIB
  for args in "--spheres 0 --no-plane" "--spheres 0" "--workload ns"; do
    timeout -k 10 120 python bench.py $args --steps 96 --warmup 16 --lean 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$args', 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'ms/frame', d['ms_per_step'])" || echo "$v $args failed"
  done
done
cp /tmp/rtc_orig.so $LIB
