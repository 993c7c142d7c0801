# same-box A/B of the single-GPU lock-step: actor-stream priority x first-dense-layer K splits (bench.py's own flags), three interleaved repetitions
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in "--actor-stream low" "--actor-stream normal" "--actor-stream high" "--actor-stream low --fc1-neighbour 8" "--actor-stream low --fc1-neighbour 2" "--actor-stream default"; do
    python bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 150 --warmup 30 $v 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$v'.ljust(44), round(d['value']), round(d['ms_per_lock_step'], 4), 'conv', round(r['avg_launch_ms'], 4), 'fc1', round(r['fc1']['avg_launch_ms'], 4) if r.get('fc1') else None)"
  done
done
