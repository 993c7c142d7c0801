#!/bin/bash
# same-box A/B: K splits of the learner's first-dense-layer launches (SRLX_FC1_TARGET_WGS workgroups: 512 = 31 splits of 8 slabs, 256 = 16 of 16, 128 = 8 of 31)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
one() { env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-per-micro --steps 12 2>gpurun_out/bench_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('subfigures',{}); print('%-30s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms  actors-only %.3f  update-only %.3f' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0), s.get('actors_only',{}).get('ms_per_lock_step',0), s.get('learner_only',{}).get('ms_per_update',0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one SRLX_FC1_TARGET_WGS=512
one SRLX_FC1_TARGET_WGS=256
one SRLX_FC1_TARGET_WGS=384
one SRLX_FC1_TARGET_WGS=128
done
} 2>&1 | tee gpurun_out/r4_probe29.log
