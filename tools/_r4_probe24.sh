#!/bin/bash
# same-box A/B: pixel chunks per (sample, frame) of conv1's weight gradient (SRLX_C1_CHUNKS 2 / 4), on the write-through partials build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for c in 2 4; do SRLX_C1_CHUNKS=$c timeout 1200 python -m pytest tests/test_qnet_gpu.py tests/test_engine_gpu.py tests/test_fast_lockstep_gpu.py tests/test_agent57_engine_gpu.py -x -q 2>&1 | tail -1; done
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-24s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2 3; do
one SRLX_C1_CHUNKS=2
one SRLX_C1_CHUNKS=4
done
for c in 2 4; do SRLX_C1_CHUNKS=$c bash tools/_trace_learner_fast.sh 2>&1 | grep "conv1_wgrad\|span"; done
} 2>&1 | tee gpurun_out/r4_probe24.log
