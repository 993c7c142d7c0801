#!/bin/bash
# same-box A/B: HIP graph runtime switches (streams a graph launch may use, packet capture) against the update graph's branch count
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-74s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one SRLX_FC1_ORDER=0
one SRLX_FC1_ORDER=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
one SRLX_FC1_ORDER=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
one SRLX_FC1_ORDER=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
one SRLX_FC1_ORDER=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
one SRLX_FC1_ORDER=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
one SRLX_FC1_ORDER=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=6
one SRLX_FC1_ORDER=2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
one SRLX_FC1_ORDER=2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
one SRLX_FC1_ORDER=2 GPU_MAX_HW_QUEUES=3
done
} 2>&1 | tee gpurun_out/r4_probe13.log
