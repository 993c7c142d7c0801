#!/bin/bash
# when do the actors' kernels START after the update graph's launch?  (a stamp kernel on the actors' stream; no stamps inside the graph)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for arm in "SRLX_ACTOR_STREAM=" "SRLX_ACTOR_STREAM=normal" "SRLX_ACTOR_STREAM=high" "SRLX_ACTOR_STREAM=low" "SRLX_ACTOR_STREAM=high SRLX_LEARNER_PRIO=0" "SRLX_ACTOR_STREAM=normal SRLX_LEARNER_PRIO=0" "SRLX_ACTOR_STREAM= SRLX_LEARNER_PRIO=0"; do
echo "== $arm"
env $arm SRLX_PY_MARKS=0 SRLX_BACKWARD_STAMPS=0 SRLX_LEARNER_PHASES=1 timeout 300 python tools/lockstep_phases.py 2>&1 | grep "ACTORS\|lock-step end\|Error\|error" | head -8
done
} 2>&1 | tee gpurun_out/r4_probe16.log
