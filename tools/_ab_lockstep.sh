#!/bin/bash
# same-box A/B of library builds / switches on the Rainbow lock-step (bench.py, 12 steps, no side measurements)
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { env "$@" python $R/bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-70s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms' % ('$*', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0)))"; }
for rep in 1 2; do
  one SRLX_FC1_PLANES=0
  one SRLX_FC1_PLANES=1
  [ -f $R/tools/_ab/libsrlx_r3start.so ] && one SRLX_LIB=$R/tools/_ab/libsrlx_r3start.so SRLX_NO_FC1_PLANES=1
done
