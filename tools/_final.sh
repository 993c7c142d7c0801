#!/bin/bash
# full GPU suite (per-test timeouts, log in gpurun_out/) followed by the round's measurement pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -q -m gpu --timeout 200 -p no:cacheprovider > gpurun_out/gputest.log 2>&1; tail -5 gpurun_out/gputest.log
bash tools/r2_measure.sh
