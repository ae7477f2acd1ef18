# reflective workloads (stack kernel): 1080p north-star scene with kr > 0, and C4 (4096^2)
cd $GRAFT_REPO_ROOT
for cfg in "--width 1920 --height 1080" "--width 2048 --height 2048" "--width 4096 --height 4096"; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --lean --reflective $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], 'ms/frame kernel', d['roofline']['kernel_ms_avg'], d['value'], 'Mrays/s')"
done
