#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-40s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -20 gpurun_out/bench_err.log; }
{
one SRLX_FAST=1
one SRLX_FAST=1 SRLX_FC1_SPLITS=8
one SRLX_FAST=1 SRLX_FC1_SPLITS=16
one SRLX_FAST=1 SRLX_NO_PLANES_GEMM=1
echo "--- actor CU mask (fast engine)"
timeout 600 python tools/actor_mask_probe.py 2>&1 | grep "CUs kept"
} 2>&1 | tee gpurun_out/r4_probe2.log
