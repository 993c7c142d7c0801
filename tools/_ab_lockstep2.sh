# same-box A/B: K splits of the actors' first dense layer (= how many CUs its 64 tiles x splits workgroups occupy) x the actors' stream priority
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for v in "--actor-stream low --fc1-neighbour 2" "--actor-stream low --fc1-neighbour 1" "--actor-stream low --fc1-neighbour 3" "--actor-stream normal --fc1-neighbour 2" "--actor-stream high --fc1-neighbour 2" "--actor-stream high --fc1-neighbour 1" "--actor-stream low --fc1-neighbour 4"; do
    python bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 150 --warmup 30 $v 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$v'.ljust(44), round(d['value']), round(d['ms_per_lock_step'], 4), 'conv', round(r['avg_launch_ms'], 4), 'fc1', round(r['fc1']['avg_launch_ms'], 4) if r.get('fc1') else None)"
  done
done
