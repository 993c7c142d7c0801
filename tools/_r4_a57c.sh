#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
timeout 900 python -m pytest tests/test_agent57_engine_gpu.py -x -q 2>&1 | tail -4
one() { timeout 600 python bench.py --algo agent57_light --steps 6 --inner 16 --warmup 1 "$@" 2>gpurun_out/a57_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-20s %9.0f env-steps/s %.3f ms per lock-step' % ('$*', d['value'], d['ms_per_lock_step']))" || tail -5 gpurun_out/a57_err.log; }
one
one --no-overlap
one
one --no-overlap
} 2>&1 | tee gpurun_out/r4_a57c.log
