# same-box A/B after k_fc1_planes_h's loop was cleaned (fragments double-buffered per k-step): K splits of the actors' first dense layer, three interleaved repetitions
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in "--fc1-neighbour 2" "--fc1-neighbour 4" "--fc1-neighbour 3" "--fc1-neighbour 8"; do
  python bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 150 --warmup 30 $v 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('[$v]', round(d['value']), round(d['ms_per_lock_step'], 4), 'conv', round(r['avg_launch_ms'], 4), 'fc1', round(r['fc1']['avg_launch_ms'], 4), r['fc1'].get('kernel_span_ms'))"
done; done
