#!/bin/bash
# same-box A/B: priority write-back on the weight-gradient branch (SRLX_UPDATE_SIDE) x where the Adam-fused FC1 weight gradient goes (SRLX_FC1_ORDER 0 last / 1 first / 2 own branch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
timeout 1200 python -m pytest tests/test_fast_lockstep_gpu.py -x -q 2>&1 | tail -2
SRLX_FC1_ORDER=2 timeout 1200 python -m pytest tests/test_fast_lockstep_gpu.py -x -q 2>&1 | tail -2
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-44s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one SRLX_UPDATE_SIDE=0 SRLX_FC1_ORDER=0
one SRLX_UPDATE_SIDE=1 SRLX_FC1_ORDER=0
one SRLX_UPDATE_SIDE=1 SRLX_FC1_ORDER=1
one SRLX_UPDATE_SIDE=1 SRLX_FC1_ORDER=2
done
for o in 0 2; do SRLX_UPDATE_SIDE=1 SRLX_FC1_ORDER=$o python tools/lockstep_phases.py 2>&1 | tail -6; done
} 2>&1 | tee gpurun_out/r4_probe12.log
