for v in "" "--no-multi-trunk" "--updates 0" "--updates 0 --no-multi-trunk"; do
python bench.py --algo agent57_light --steps 6 --warmup 1 --inner 16 --no-cpu-baseline $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '|', round(d['value']), d['ms_per_lock_step'], d['roofline']['avg_launch_ms'])
"
done
