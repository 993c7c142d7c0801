#!/bin/bash
# Agent57_light: actors on a low-priority stream (own hardware-queue pool) against torch's current stream, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
one() { timeout 400 python bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 --no-cpu-baseline "$@" 2>gpurun_out/bench_err.log | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-28s %8d env-steps/s %.3f ms per lock-step  roofline frac %.3f' % ('$*', d['value'], d['ms_per_lock_step'], (d.get('roofline') or {}).get('frac', 0)))" || tail -3 gpurun_out/bench_err.log; }
for rep in 1 2; do
one --actor-stream default
one --actor-stream low
done
timeout 400 python bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 --cpu-seconds 8 2>/dev/null | tail -1 | cut -c1-3000
} 2>&1 | tee gpurun_out/r4_a57d.log
