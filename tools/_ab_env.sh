#!/bin/bash
# same-box A/B of environment switches on the Rainbow lock-step: bash tools/_ab_env.sh "A=1" "B=2 C=3" ...   (each argument = one arm; run twice, interleaved)
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { env $1 python $R/bench.py --no-cpu-baseline --no-per-micro --no-subfigures --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-50s %8d env-steps/s  %.4f ms per lock-step  conv %.3f ms  fc1 %.3f ms' % ('$1', d['value'], d['ms_per_lock_step'], r['avg_launch_ms'], (r.get('fc1') or {}).get('avg_launch_ms', 0)))"; }
for rep in 1 2; do for arm in "$@"; do one "$arm"; done; done
