#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
echo "--- phases, update in line"
timeout 300 python tools/lockstep_phases.py 2>&1 | tail -6
echo "--- phases, update on a branch"
# (the SRLX_UPDATE_BRANCH arm of this probe was removed with the code path: superseded by srlx_qnet_set_priority_sink)
echo "--- phases, legacy"
SRLX_FAST=0 timeout 300 python tools/lockstep_phases.py 2>&1 | tail -6
} 2>&1 | tee gpurun_out/r4_probe8.log
