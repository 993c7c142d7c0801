#!/bin/bash
# same-box A/B: 128-column tiles for the learner's first-dense-layer launches (SRLX_FC1_BN128=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
SRLX_FC1_BN128=1 timeout 1200 python -m pytest tests/test_qnet_gpu.py tests/test_fast_lockstep_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -2
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-34s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2 3; do
one SRLX_FC1_BN128=0
one SRLX_FC1_BN128=1
done
SRLX_FC1_BN128=1 bash tools/_trace_learner_fast.sh 2>&1 | grep "gemm_s16\|span"
} 2>&1 | tee gpurun_out/r4_probe20.log
