cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline $EXTRA 2>/dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['learner_updates_per_s'], d['roofline']['avg_launch_group_ms'])"; }
echo "== fused adam"; run
echo "== unfused adam"; SRLX_FUSED_ADAM=0 run
echo "== fused adam, side update, no graph"; EXTRA="--no-graph" SRLX_SIDE_UPDATE=1 run
EXTRA="--envs 16 --capacity 200000"
echo "== learner alone fused"; run
echo "== learner alone unfused"; SRLX_FUSED_ADAM=0 run
echo "== learner alone fused nograph"; EXTRA="--envs 16 --capacity 200000 --no-graph" run
echo "== learner alone unfused nograph"; EXTRA="--envs 16 --capacity 200000 --no-graph" SRLX_FUSED_ADAM=0 run
