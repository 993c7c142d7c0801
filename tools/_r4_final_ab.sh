#!/bin/bash
# the round's total on ONE box: bench.py as shipped against the round-3 lock-step (SRLX_FAST=0, actors on torch's current stream), arms interleaved three times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
one() { env $1 timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 $2 $3 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-46s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2 3; do
one X=shipped
one SRLX_FAST=0 --actor-stream default
one SRLX_FAST=1 --actor-stream default
done
} 2>&1 | tee gpurun_out/r4_final_ab.log
