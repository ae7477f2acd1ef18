# The BASELINE.json configurations on one GPU (same build as the headline run): bash tools/other_configs.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in c2 c3 c4 c5 ns; do
  timeout -k 10 300 python bench.py --workload $w --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/cfg_$w.log 2>&1
  grep -a '^{' gpurun_out/cfg_$w.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$w', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s', c['rays_per_frame_primary_shadow'], c['rays_per_frame_other'], 'kernel/frame', d['roofline']['kernel_ms_per_frame'], 'single_view', d['single_view']['ms_per_step'])"
done
