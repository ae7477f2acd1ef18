# The other single-GPU configurations of DESIGN.md's table (same build as the headline run).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/cfg_$tag.log 2>&1; grep -a '^{' gpurun_out/cfg_$tag.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$tag', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s', c['rays_per_frame_primary_shadow'], c['rays_per_frame_other'], 'kernel', d['roofline']['kernel_ms_avg'])"; }
run c2 --spheres 3 --steps 200 --warmup 20 &&
run c3 --spheres 10000 --no-plane --steps 100 --warmup 10 &&
run c4 --spheres 100 --width 4096 --height 4096 --reflective --steps 30 --warmup 5 &&
run c5 --spheres 1000 --width 8192 --height 8192 --steps 20 --warmup 3 &&
run ns --steps 400 --warmup 40
