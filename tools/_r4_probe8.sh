#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
echo "--- phases, update in line"
timeout 300 python tools/lockstep_phases.py 2>&1 | tail -6
echo "--- phases, update on a branch"
SRLX_UPDATE_BRANCH=1 timeout 300 python tools/lockstep_phases.py 2>&1 | tail -6
echo "--- phases, legacy"
SRLX_FAST=0 timeout 300 python tools/lockstep_phases.py 2>&1 | tail -6
} 2>&1 | tee gpurun_out/r4_probe8.log
