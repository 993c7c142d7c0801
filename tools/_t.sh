cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline $EXTRA 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['learner_updates_per_s'], d['roofline']['avg_launch_group_ms'], d['final'])"; }
echo "== default"; run; run
EXTRA="--updates 0"; echo "== actor only"; run
EXTRA="--envs 16 --capacity 200000"; echo "== learner alone"; run
