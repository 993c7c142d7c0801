#!/bin/bash
# operating points of the shipped lock-step on one box: environments per lock-step (one update each); the actors have ~60 us of slack at E = 1024
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for e in 1024 1152 1280 1536 2048; do
timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 --envs $e 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('subfigures',{}); print('E = %5d  %8d env-steps/s  %7.1f updates/s  %.4f ms per lock-step  actors-only %.3f  update-only %.3f' % ($e, d['value'], d['learner_updates_per_s'], d['ms_per_lock_step'], s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log
done
} 2>&1 | tee gpurun_out/r4_operating_points.txt
