#!/bin/bash
# same-box A/B on top of --actor-stream low: learner stream priority, write-back placement, K splits of the actors' first dense layer, hardware queues per pool
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-44s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one X=low
one SRLX_LEARNER_PRIO=0
one SRLX_UPDATE_SIDE=0
one SRLX_FC1_NEIGHBOUR=8
one SRLX_FC1_NEIGHBOUR=2
one GPU_MAX_HW_QUEUES=4
one GPU_MAX_HW_QUEUES=1
done
} 2>&1 | tee gpurun_out/r4_probe23.log
