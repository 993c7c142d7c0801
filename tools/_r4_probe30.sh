#!/bin/bash
# same-box A/B: the actors' convolution pass as n launches of consecutive samples (SRLX_CONV_CHUNKS): the dispatcher feeds a running kernel's pending workgroups
# before another queue's, so the update's convolution pass waited for the actors' last round
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
SRLX_CONV_CHUNKS=4 timeout 1200 python -m pytest tests/test_qnet_gpu.py tests/test_qnet_pinned.py tests/test_fast_lockstep_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -1
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-24s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one SRLX_CONV_CHUNKS=1
one SRLX_CONV_CHUNKS=2
one SRLX_CONV_CHUNKS=4
one SRLX_CONV_CHUNKS=5
one SRLX_CONV_CHUNKS=8
done
SRLX_CONV_CHUNKS=4 SRLX_ACTOR_STREAM=low SRLX_BACKWARD_STAMPS=1 python tools/freerun_phases.py 2>&1 | grep "free-running"
} 2>&1 | tee gpurun_out/r4_probe30.log
