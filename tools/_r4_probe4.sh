#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-50s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0)))" || tail -20 gpurun_out/bench_err.log; }
{
one SRLX_FC1_NEIGHBOUR=4
one SRLX_FC1_NEIGHBOUR=4 GPU_MAX_HW_QUEUES=3
one SRLX_FC1_NEIGHBOUR=4 GPU_MAX_HW_QUEUES=4
one SRLX_FC1_NEIGHBOUR=4 GPU_MAX_HW_QUEUES=8
one SRLX_FC1_NEIGHBOUR=4 SRLX_LEARNER_PRIO=0
echo "--- timeline of one lock-step (fast, neighbour 4)"
SRLX_FC1_NEIGHBOUR=4 bash tools/_trace_loop.sh
echo "--- phases"
SRLX_FC1_NEIGHBOUR=4 timeout 300 python tools/lockstep_phases.py 2>&1 | tail -5
} 2>&1 | tee gpurun_out/r4_probe4.log
