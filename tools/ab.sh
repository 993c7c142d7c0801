#!/bin/bash
# SAME-box A/B of the lock-step (boxes of the pool differ by +-5 %, more than most kernel changes gain): every arm runs in one gpurun call, interleaved REPS times.
#   usage:  gpurun -- 'bash tools/ab.sh "name1:ENV=VAL ENV2=VAL" "name2:SRLX_LIB=$PWD/tools/_abl/libsrlx_x.so" ...'
# An arm is "label:environment assignments"; SRLX_LIB=<path> selects another BUILD of libsrlx (simple_distributed_rl_amd/_native.py).  CMD (default: the bench
# line without its side figures) and REPS (default 2) come from the environment.  Prints label, env-steps/s, ms per lock-step, updates/s per run.
cd ${GRAFT_REPO_ROOT:-.}
CMD=${CMD:-"python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-per-micro --no-subfigures"}
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for arm in "$@"; do
    label=${arm%%:*}; envs=${arm#*:}; [ "$envs" = "$arm" ] && envs=""
    out=$(env $envs $CMD 2>/dev/null | tail -1)
    echo "$label $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('%.4f M env-steps/s  %.4f ms per lock-step  %.0f updates/s' % (d['value']/1e6, d['ms_per_lock_step'], d['learner_updates_per_s']))" "$out" 2>/dev/null || echo FAILED)"
  done
done
