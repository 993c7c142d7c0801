#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/per_probe.py quick 4000 30000 60000 120000 250000 500000 1000000 4000000 2>&1 | grep draws | cut -c1-110
