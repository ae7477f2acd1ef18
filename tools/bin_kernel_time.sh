# Duration of the binning kernel (rocprofv3 kernel trace) beside the render kernel: bash tools/bin_kernel_time.sh "<workloads>"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in $1; do
  rm -rf /tmp/bk_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk_$w -- python3 bench.py --workload $w --steps 96 --warmup 16 --lean > /tmp/bk_$w.log 2>&1
  f=$(ls /tmp/bk_$w/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$w" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_bin", "void k_trace")):
        print(sys.argv[2], r["Name"][:24], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
