#!/bin/bash
cd $GRAFT_REPO_ROOT
for fr in 0 5 6; do for cfg in 512x4 1024x2; do echo "cfg $cfg free $fr"; SRLX_PER_FREE=$fr SRLX_PER_CFG=$cfg timeout 300 python tools/per_probe.py quick 2>&1 | grep draws | cut -c1-100; done; done
