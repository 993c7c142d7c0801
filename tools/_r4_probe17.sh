#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for arm in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=4" "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=4" "SRLX_UPDATE_SIDE=0"; do
echo "== $arm"; env $arm timeout 300 python tools/graph_launch_host.py 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/r4_probe17.log
