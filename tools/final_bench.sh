# The round's bench lines on ONE box: bash tools/final_bench.sh <tag>  -> gpurun_out/<tag>/bench_*.json
cd $GRAFT_REPO_ROOT
TAG=${1:-final}
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driverlike.json 2> gpurun_out/$TAG/bench_driverlike.err
for w in c2 c3 c4 c5; do timeout -k 10 600 python bench.py --workload $w --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null > gpurun_out/$TAG/bench_$w.json; done
for f in gpurun_out/$TAG/bench_*.json; do
  python3 tools/_line.py "$(basename $f .json)" value ms_per_step roofline.frac roofline.kernel_ms_avg roofline.binning_kernel_ms_avg serial_single_view.ms_per_step batched_views.ms_per_frame brute_force_lds.kernel_ms_per_frame brute_force_lds.f64_valu.frac dropin.rtc_render_rgb8_ms cpu_baseline.value cpu_baseline.cores < $f
done
