#!/bin/bash
# same-box A/B: conv1 weight gradient's partial sums added up in the launch (tickets) or by a k_reduce_parts launch behind it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for i in 1 2 3 4 5 6; do python -m pytest tests/test_agent57_engine_gpu.py -x -q -k "trainable_trunk_gradients" 2>&1 | tail -1 | cut -c1-10; done | sort | uniq -c
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-30s %8d env-steps/s  %.4f ms per lock-step  update-only %.4f' % ('$*', d['value'], d['ms_per_lock_step'], s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2 3; do
one SRLX_C1_REDUCE=launch
one SRLX_C1_REDUCE=in_launch
done
} 2>&1 | tee gpurun_out/r4_probe32.log
