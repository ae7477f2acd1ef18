cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for te in 1 4 1000; do
python bench.py --workload ns --steps 192 --warmup 16 --lean --time-every $te 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('time-every $te', 'kernel/frame', d['roofline'].get('kernel_ms_per_frame'), 'ms/frame', d['ms_per_step'])"
done; done
